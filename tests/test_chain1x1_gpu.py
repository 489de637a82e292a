"""The register-chained 1x1 pair of the frozen bottleneck stack (csrc/chain1x1.hip, F.chain_expand_reduce):
    out = relu(bn3(conv3(y1)) + identity),  y0' = relu(bn1'(conv1'(out)))        (models/resnet_encoder.py:115-126)
against float64, against the two layers run separately on the conv engine, launch-to-launch repeatability (the kernel counts
its LDS-DMA / load / store completions by hand: a missed wait shows as a difference between launches), and through the
ResNet trunk (chained vs layer-by-layer features)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def _case(rows, n2, seed, kind="normal", k=256):
    g = torch.Generator().manual_seed(seed)
    if kind == "normal":
        y1 = torch.relu(torch.randn(rows, k, generator=g))
        res = torch.relu(torch.randn(rows, n2, generator=g))
        w2 = torch.randn(n2, k, generator=g) * (2.0 / k) ** 0.5
        w3 = torch.randn(k, n2, generator=g) * (2.0 / n2) ** 0.5
    else:                       # wide dynamic range
        y1 = torch.randn(rows, k, generator=g) * torch.exp2(torch.randint(-10, 11, (rows, k), generator=g).float())
        res = torch.randn(rows, n2, generator=g) * torch.exp2(torch.randint(-6, 7, (rows, n2), generator=g).float())
        w2 = torch.randn(n2, k, generator=g) * torch.exp2(torch.randint(-6, 7, (n2, k), generator=g).float()) / 16
        w3 = torch.randn(k, n2, generator=g) * torch.exp2(torch.randint(-6, 7, (k, n2), generator=g).float()) / 32
    s2, t2 = 0.5 + torch.rand(n2, generator=g), 0.2 * torch.randn(n2, generator=g)
    s3, t3 = 0.5 + torch.rand(k, generator=g), 0.2 * torch.randn(k, generator=g)
    return y1, res, w2, w3, s2, t2, s3, t3


def _ref(y1, res, w2, w3, s2, t2, s3, t3, dt):
    y1, res, w2, w3, s2, t2, s3, t3 = (v.to(dt) for v in (y1, res, w2, w3, s2, t2, s3, t3))
    out = torch.relu(y1 @ w2.t() * s2 + t2 + res)
    return out, torch.relu(out @ w3.t() * s3 + t3)


@pytest.mark.parametrize("kind", ["normal", "wide"])
@pytest.mark.parametrize("rows,n2,k", [(128, 64, 256), (256, 1024, 256), (16384, 1024, 256), (4096, 512, 256), (8192, 1024, 256), (1024, 256, 256),
                                       (128, 64, 64), (32768, 256, 64), (1024, 128, 64), (128, 64, 128), (16384, 512, 128), (4096, 1024, 128)])
def test_chain_matches_float64_and_the_unfused_layers(rows, n2, k, kind):
    """k = the mid width: 256 (layer3) on both arithmetics; 128 / 64 (layer2 / layer1) exist on the fp16-pair arithmetic only."""
    from codeps_amd import functional as F
    if k != 256 and F._arith() != 16:
        pytest.skip("the narrow forms exist on the fp16-pair arithmetic only")
    y1, res, w2, w3, s2, t2, s3, t3 = _case(rows, n2, 11 + rows % 97 + n2 + k, kind, k)
    o64, z64 = _ref(y1, res, w2, w3, s2, t2, s3, t3, torch.float64)
    o32, z32 = _ref(y1, res, w2, w3, s2, t2, s3, t3, torch.float32)
    d = lambda v: v.to(DEV)      # noqa: E731
    h = 8 if rows % 8 == 0 else 1
    y1d, resd = d(y1).view(1, h, rows // h, k), d(res).view(1, h, rows // h, n2)
    w2d, w3d = d(w2).view(n2, k, 1, 1), d(w3).view(k, n2, 1, 1)
    with torch.no_grad():
        out, z = F.chain_expand_reduce(y1d, resd, w2d, d(s2), d(t2), w3d, d(s3), d(t3))
        # the same two layers, one launch each, on the conv engine
        uo = F.conv2d(y1d, w2d, scale=d(s2), shift=d(t2), residual=resd, act="relu")
        uz = F.conv2d(uo, w3d, scale=d(s3), shift=d(t3), act="relu")
    torch.cuda.synchronize()
    out, z, uo, uz = out.view(rows, n2).cpu(), z.view(rows, k).cpu(), uo.view(rows, n2).cpu(), uz[..., :k].reshape(rows, k).cpu()
    e_o, e_z = _l2(out, o64), _l2(z, z64)
    t_o, t_z = _l2(o32, o64), _l2(z32, z64)
    u_o, u_z = _l2(uo, o64), _l2(uz, z64)
    # within 3x torch's own fp32 error, or 1.5x the unfused engine's (same products, another summation order)
    assert e_o <= max(3 * t_o, 1.5 * u_o) + 1e-9, f"expand: {e_o:.3e} vs torch fp32 {t_o:.3e}, unfused {u_o:.3e}"
    assert e_z <= max(3 * t_z, 1.5 * u_z) + 1e-9, f"reduce: {e_z:.3e} vs torch fp32 {t_z:.3e}, unfused {u_z:.3e}"
    # element-wise: no pixel / channel went to the wrong place (a permutation slip would be O(1) wrong somewhere)
    scale = float(o64.abs().max())
    assert float((out.double() - o64).abs().max()) <= 2e-5 * scale
    assert float((z.double() - z64).abs().max()) <= 5e-5 * float(z64.abs().max())


@pytest.mark.parametrize("k,n2", [(256, 1024), (128, 512), (64, 256)])
def test_chain_is_repeatable_launch_to_launch(k, n2):
    from codeps_amd import functional as F
    if k != 256 and F._arith() != 16:
        pytest.skip("the narrow forms exist on the fp16-pair arithmetic only")
    y1, res, w2, w3, s2, t2, s3, t3 = _case(16384, n2, 5, "normal", k)
    d = lambda v: v.to(DEV)      # noqa: E731
    y1d, resd = d(y1).view(8, 32, 64, k), d(res).view(8, 32, 64, n2)
    w2d, w3d = d(w2).view(n2, k, 1, 1), d(w3).view(k, n2, 1, 1)
    args = (y1d, resd, w2d, d(s2), d(t2), w3d, d(s3), d(t3))
    with torch.no_grad():
        o0, z0 = F.chain_expand_reduce(*args)
        busy = torch.empty(64 << 20, device=DEV)
        for i in range(200):
            if i % 3 == 0:
                busy.normal_()                       # unrelated traffic between launches: changes DMA / load timing
            o, z = F.chain_expand_reduce(*args)
            assert torch.equal(o, o0) and torch.equal(z, z0), f"launch {i} differs from launch 0"


def test_chain_refuses_what_it_cannot_do():
    from codeps_amd import functional as F
    from codeps_amd._lib import CodepsHipError
    y1 = torch.zeros(1, 8, 16, 256, device=DEV)
    w2 = torch.zeros(1024, 256, 1, 1, device=DEV)
    w3 = torch.zeros(256, 1024, 1, 1, device=DEV)
    v = torch.zeros(1024, device=DEV)
    v3 = torch.zeros(256, device=DEV)
    with torch.no_grad():
        F.chain_expand_reduce(y1, torch.zeros(1, 8, 16, 1024, device=DEV), w2, v, v, w3, v3, v3)
        with pytest.raises(CodepsHipError):            # 120 pixels: not a multiple of 128
            F.chain_expand_reduce(y1[:, :, :15].contiguous(), torch.zeros(1, 8, 15, 1024, device=DEV), w2, v, v, w3, v3, v3)
    with pytest.raises(CodepsHipError):                # inference only
        F.chain_expand_reduce(y1.requires_grad_(True), torch.zeros(1, 8, 16, 1024, device=DEV), w2, v, v, w3, v3, v3)


@pytest.mark.parametrize("layers", [50, 101])
def test_trunk_features_with_and_without_the_chain(layers):
    """ResNet-50 / 101 in eval mode (the frozen encoder of the adaptation step): the features with layer3's 1x1 pairs chained
    equal the layer-by-layer features to fp32 rounding (another summation order inside the 32-channel groups)."""
    from codeps_amd import functional as F
    from codeps_amd.models import ResnetEncoder
    torch.manual_seed(3)
    enc = ResnetEncoder(layers, False).to(DEV).eval()
    for m in enc.modules():                          # non-trivial eval statistics
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
    x = torch.randn(2, 3, 128, 256, device=DEV)
    prev = F._CHAIN
    try:
        with torch.no_grad():
            F._CHAIN = True
            a = [f.clone() for f in enc(x)]
            F._CHAIN = False
            b = [f.clone() for f in enc(x)]
    finally:
        F._CHAIN = prev
    for i, (fa, fb) in enumerate(zip(a, b)):
        assert fa.shape == fb.shape
        assert _l2(fa, fb) < 5e-6, f"feature {i}: {_l2(fa, fb):.2e}"
    assert not torch.equal(a[3], b[3]), "the chained path did not run (layer3 features are bit-identical)"
    if F._arith() == 16:
        assert not torch.equal(a[1], b[1]) and not torch.equal(a[2], b[2]), "the narrow chained forms (layer1 / layer2) did not run"


@pytest.mark.parametrize("b,h,w", [(2, 64, 128), (1, 96, 160), (3, 32, 64)])
def test_frozen_three_channel_stem_against_float64_and_the_double_pixel_form(b, h, w):
    """F.stem3_conv2d (conv1 as a 7 x 1 convolution over overlapping 8-pixel windows of 4-channel pixels, K = 224) against
    torch's conv2d in float64 and against the double-pixel form (F.stem_conv2d, K = 448) on the same images."""
    import torch.nn.functional as TF
    from codeps_amd import functional as F
    g = torch.Generator().manual_seed(b * 100 + h)
    img = torch.randn(b, 3, h, w, generator=g)
    wt = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    ref = torch.relu(TF.conv2d(img.double(), wt.double(), None, 2, 3) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    r32 = torch.relu(TF.conv2d(img, wt, None, 2, 3) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    with torch.no_grad():
        x4 = F.images_to_nhwc([img.to(DEV)], 4, F.STEM3_XPAD)
        y3 = F.stem3_conv2d(x4, wt.to(DEV), scale=sc.to(DEV), shift=sh.to(DEV), act="relu")
        x8 = F.images_to_nhwc([img.to(DEV)], 8, F.STEM_XPAD)
        y8 = F.stem_conv2d(x8, wt.to(DEV), scale=sc.to(DEV), shift=sh.to(DEV), act="relu")
    got3, got8 = y3[..., :64].permute(0, 3, 1, 2).cpu(), y8[..., :64].permute(0, 3, 1, 2).cpu()
    assert got3.shape == ref.shape
    e3, e8, e32 = _l2(got3, ref), _l2(got8, ref), _l2(r32, ref)
    assert e3 <= max(3 * e32, 1.5 * e8) + 1e-9, f"stem3 {e3:.2e}, double-pixel {e8:.2e}, torch fp32 {e32:.2e}"
    assert float((got3.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
