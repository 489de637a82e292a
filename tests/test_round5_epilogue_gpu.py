"""Round-5 epilogue extras of the conv engine (codeps_conv_epilogue) against the separate passes they replace: activation backward +
bias gradient in the consumer's data-gradient epilogue (bit-identical), train-mode BatchNorm statistics from the producing
convolution's epilogue and BatchNorm's backward reduce from the consuming convolution's (same values up to summation order; checked
against float64), the misuse guard of `exclusive=True`."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _kfam(name):
    from helpers import kernel_family
    return kernel_family(name)

DEV = "cuda:0"


def _F():
    from codeps_amd import functional as F
    return F


def _nhwc(g, b, h, w, c, scale=1.0):
    return (torch.randn(b, h, w, c, generator=g) * scale).to(DEV)


def _conv_chain(F, x, ws, bs, acts, strides, exclusive, up=None, skip=None):
    """conv + act chain on NHWC; layer i > 0 reads layer i - 1 exclusively when asked to"""
    y = x
    for i, (w, b, a, st) in enumerate(zip(ws, bs, acts, strides)):
        k = w.shape[-1]
        kw = dict(stride=st, pad=k // 2, act=a, exclusive=exclusive and i > 0)
        if up is not None and i == up:
            y = F.conv2d(y, w, b, x1=skip, up0=True, **kw)
        else:
            y = F.conv2d(y, w, b, **kw)
    return y


@pytest.mark.parametrize("case", [
    # (b, h, w, channels per layer, kernel sizes, activations, strides)
    (2, 32, 64, [64, 64, 64, 32], [3, 3, 3], ["elu", "elu", None], [1, 1, 1]),            # halo kernels, 64 and 32 columns
    (2, 16, 32, [128, 128, 128], [3, 3], ["relu", "relu"], [1, 1]),                       # ring kernel (maybe K slices: the finish pass gates)
    (1, 16, 32, [512, 256, 256, 16], [1, 3, 1], ["relu", "relu", None], [1, 1, 1]),       # pose head: 1x1 -> 3x3 -> 1x1 to 16 columns
    (3, 32, 64, [32, 16, 16], [3, 3], ["elu", "elu"], [1, 1]),                            # 16-column kernels
    (2, 32, 64, [64, 64, 128], [3, 3], ["relu", None], [1, 2]),                           # stride-2 consumer: four parity-class launches
    (2, 8, 16, [256, 256, 256], [3, 3], ["elu", "elu"], [1, 1]),                          # small maps
])
def test_activation_backward_in_the_consumer_epilogue_is_bit_identical(case):
    F = _F()
    b, h, w, chans, ks, acts, strides = case
    g = torch.Generator().manual_seed(sum(chans) + h)
    x = _nhwc(g, b, h, w, chans[0])
    ws = [(torch.randn(chans[i + 1], chans[i], ks[i], ks[i], generator=g) / (chans[i] * ks[i] * ks[i]) ** 0.5).to(DEV).requires_grad_() for i in range(len(ks))]
    bs = [(torch.randn(chans[i + 1], generator=g) * 0.1).to(DEV).requires_grad_() for i in range(len(ks))]
    res = {}
    for excl in (False, True):
        xi = x.clone().requires_grad_()
        y = _conv_chain(F, xi, ws, bs, acts, strides, excl)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
        grads = torch.autograd.grad(y, [xi] + ws + bs, gy)
        res[excl] = [y] + list(grads)
    _same(res, len(bs))


def _same(res, n_bias):
    """everything bit for bit, except the bias gradients: their column sums come from per-tile partials of the consumer's launch
    instead of act_bwd_kernel's row blocks -- another (fixed) order of the same additions"""
    n = len(res[False])
    for i, (a, c) in enumerate(zip(res[False], res[True])):
        if i >= n - n_bias:
            assert float((a - c).abs().max()) <= 2e-6 * (float(a.abs().max()) + float(a.abs().sum()) / a.numel() * 100), (i, float((a - c).abs().max()))
        else:
            assert torch.equal(a, c), (i, float((a - c).abs().max()))


def test_gate_through_the_upsampling_decoder_stage_is_bit_identical():
    """conv(i, 0) + ELU -> [nearest x2 + concat skip] conv(i, 1) + ELU -> conv(i - 1, 0): the gradient of the half-resolution tensor
    comes out of a FORWARD launch (the 4 x 4 / stride-2 form of the up-convolution's data gradient), gated in its epilogue"""
    F = _F()
    g = torch.Generator().manual_seed(3)
    b, h, w = 2, 16, 32
    x = _nhwc(g, b, h, w, 128)
    skip = _nhwc(g, b, 2 * h, 2 * w, 64)
    mk = lambda co, ci: (torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5).to(DEV).requires_grad_()     # noqa: E731
    ws = [mk(64, 128), mk(64, 128), mk(32, 64)]
    bs = [(torch.randn(c, generator=g) * 0.1).to(DEV).requires_grad_() for c in (64, 64, 32)]
    res = {}
    for excl in (False, True):
        xi = x.clone().requires_grad_()
        y = _conv_chain(F, xi, ws, bs, ["elu", "elu", "elu"], [1, 1, 1], excl, up=1, skip=skip)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
        res[excl] = [y] + list(torch.autograd.grad(y, [xi] + ws + bs, gy))
    _same(res, len(bs))


def test_forked_consumer_gates_the_sum_of_both_gradients():
    """y = elu(conv(x)); z, y' = conv2d(y, fork=True, exclusive=True); loss reads z and y': (dgrad + gradient through y') * elu'(y)"""
    F = _F()
    g = torch.Generator().manual_seed(11)
    x = _nhwc(g, 2, 32, 64, 64)
    w0 = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(DEV).requires_grad_()
    b0 = torch.zeros(64, device=DEV).requires_grad_()
    w1 = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(DEV).requires_grad_()
    res = {}
    for excl in (False, True):
        xi = x.clone().requires_grad_()
        y = F.conv2d(xi, w0, b0, pad=1, act="elu")
        z, ya = F.conv2d(y, w1, None, pad=1, fork=True, exclusive=excl)
        loss = (z * z).sum() + (ya * 0.5).sum()
        res[excl] = torch.autograd.grad(loss, [xi, w0, w1, b0])
    _same(res, 1)


def test_exclusive_with_a_second_consumer_raises():
    F = _F()
    from codeps_amd._lib import CodepsHipError
    g = torch.Generator().manual_seed(12)
    x = _nhwc(g, 1, 32, 64, 64).requires_grad_()
    w0 = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(DEV).requires_grad_()
    w1 = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(DEV).requires_grad_()
    y = F.conv2d(x, w0, torch.zeros(64, device=DEV).requires_grad_(), pad=1, act="elu")
    z = F.conv2d(y, w1, None, pad=1, exclusive=True)               # a lie: y is also read below
    loss = z.sum() + (y * y).sum()
    with pytest.raises((CodepsHipError, RuntimeError)):
        loss.backward()
        torch.cuda.synchronize()


def _bn_modules(c_in, c, n_blocks, stride, seed):
    from codeps_amd.models.resnet_encoder import _BasicBlock
    from torch import nn
    torch.manual_seed(seed)
    blocks = []
    for i in range(n_blocks):
        ds = None
        if i == 0 and (stride != 1 or c_in != c):
            ds = nn.Sequential(nn.Conv2d(c_in, c, 1, stride, bias=False), nn.BatchNorm2d(c))
        blocks.append(_BasicBlock(c_in if i == 0 else c, c, stride if i == 0 else 1, ds))
    m = nn.Sequential(*blocks).to(DEV).train()
    for p in m.parameters():
        if p.dim() == 1:
            p.data.uniform_(0.5, 1.5)
    for mod in m.modules():
        if isinstance(mod, nn.BatchNorm2d):
            mod.bias.data.normal_(0, 0.3)
    return m


def _run_blocks(F, m, x, groups, own):
    from codeps_amd.models.resnet_encoder import _run_layer
    return _run_layer(m, x, None, groups, own_input=own)


def _ref_blocks_f64(m, x_nhwc, groups):
    """float64 torch reference of the BasicBlock stack with per-group batch statistics"""
    import torch.nn.functional as tf
    x = x_nhwc.permute(0, 3, 1, 2).double()

    def bn(z, mod):
        out = torch.empty_like(z)
        for b0, b1 in groups:
            zz = z[b0:b1]
            mu = zz.mean((0, 2, 3), keepdim=True)
            var = zz.var((0, 2, 3), unbiased=False, keepdim=True)
            out[b0:b1] = (zz - mu) / torch.sqrt(var + mod.eps) * mod.weight.double().view(1, -1, 1, 1) + mod.bias.double().view(1, -1, 1, 1)
        return out
    for blk in m:
        idn = x
        out = torch.relu(bn(tf.conv2d(x, blk.conv1.weight.double(), None, blk.conv1.stride, 1), blk.bn1))
        out = bn(tf.conv2d(out, blk.conv2.weight.double(), None, 1, 1), blk.bn2)
        if blk.downsample is not None:
            idn = bn(tf.conv2d(x, blk.downsample[0].weight.double(), None, blk.downsample[0].stride), blk.downsample[1])
        x = torch.relu(out + idn)
    return x.permute(0, 2, 3, 1)


@pytest.mark.parametrize("case", [(6, 64, 128, 64, 64, 2, 1, 3), (4, 64, 128, 64, 128, 2, 2, 2), (8, 32, 64, 256, 256, 2, 1, 1), (6, 32, 64, 256, 512, 2, 2, 2),
                                  (6, 32, 64, 64, 64, 2, 1, 3), (2, 16, 32, 256, 512, 2, 2, 2)])
def test_train_mode_batchnorm_through_the_conv_epilogues_against_float64(case):
    """ResNet-18 stages in train mode: statistics from the producing convolution's epilogue, ReLU mask + (sum g, sum g xhat) in the
    consuming convolution's data-gradient epilogue -- same results as the separate passes up to summation order: both compared with a
    float64 evaluation of the same stack (outputs, input gradient, every parameter gradient, running statistics)"""
    F = _F()
    b, h, w, c_in, c, n_blocks, stride, ngroups = case
    per = b // ngroups
    groups = [(i * per, (i + 1) * per) for i in range(ngroups)]
    g = torch.Generator().manual_seed(c + h)
    x = _nhwc(g, b, h, w, c_in)
    out = {}
    for mode in ("separate", "fused"):
        m = _bn_modules(c_in, c, n_blocks, stride, seed=7)
        F._EPI = mode == "fused"
        try:
            xi = x.clone().requires_grad_()
            y = _run_blocks(F, m, xi, groups if ngroups > 1 else None, True)
            gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
            params = [p for p in m.parameters()]
            grads = torch.autograd.grad(y, [xi] + params, gy)
            stats = [t.clone() for mod in m.modules() if isinstance(mod, torch.nn.BatchNorm2d) for t in (mod.running_mean, mod.running_var)]
            out[mode] = (y.detach(), grads, stats, m)
        finally:
            F._EPI = True
    # float64 yardstick
    m64 = out["fused"][3]
    x64 = x.clone().double().requires_grad_()
    p64 = [p for p in m64.parameters()]
    y64 = _ref_blocks_f64(m64, x64, groups)
    g64 = torch.autograd.grad(y64, [x64] + p64, gy.double(), allow_unused=True)

    # relative L2: a ReLU whose pre-activation is ~1e-7 may fall on the other side of zero under another summation order of the batch
    # statistics, which moves single gradient elements by their full size (max-norm errors of 1e-3 .. 1e-2 against float64 in EITHER
    # path, as in tests/test_modules_gpu.py) but not the norm
    def err(a, ref):
        return float((a.double() - ref).norm() / (ref.norm() + 1e-30))
    e_sep, e_fus = err(out["separate"][0], y64), err(out["fused"][0], y64)
    print(f"BN case {case}: output rel-L2 error vs float64: separate {e_sep:.2e}, fused {e_fus:.2e}")
    assert e_fus <= max(2e-6, 3 * e_sep), (e_fus, e_sep)
    for i, r in enumerate(g64):
        if r is None:
            continue
        es, ef = err(out["separate"][1][i], r.double()), err(out["fused"][1][i], r.double())
        assert ef <= max(5e-3, 3 * es), (i, ef, es)            # (a flipped ReLU moves the norm by ~ 1 / sqrt(elements))
    for s_, f_ in zip(out["separate"][2], out["fused"][2]):
        assert float((s_ - f_).abs().max()) <= 1e-5 * (1 + float(s_.abs().max()))


def test_epilogue_kernels_really_ran():
    """the fused path is the one that runs: after a fused BasicBlock step no bn_stats / bn_bwd_reduce launch is needed -- checked through
    the hand-off records (a consumer that gated leaves one; the producer clears it)"""
    F = _F()
    g = torch.Generator().manual_seed(2)
    x = _nhwc(g, 4, 64, 128, 64)
    w0 = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(DEV).requires_grad_()
    y = F.conv2d(x, w0, None, pad=1, bn_stats=True)
    assert getattr(y, "_codeps_stats", None) is not None and y._codeps_stats["rows"] in (128, 256)
    gamma, beta = torch.ones(64, device=DEV).requires_grad_(), torch.zeros(64, device=DEV).requires_grad_()
    rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
    z = F.batch_norm_train(y, gamma, beta, rm, rv, act="relu")
    gate = getattr(z, "_codeps_gate", None)
    assert gate is not None and gate.kind == "bn"
    w1 = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(DEV).requires_grad_()
    o = F.conv2d(z, w1, None, pad=1, exclusive=True)
    seen = {}
    orig = F._Gate.take

    def spy(self, dy):
        if self is gate:
            seen["handoff"] = self.handoff is not None and self.handoff["part"] is not None
        return orig(self, dy)
    F._Gate.take = spy
    try:
        o.sum().backward()
        torch.cuda.synchronize()
    finally:
        F._Gate.take = orig
    assert seen.get("handoff") is True
    # batch statistics from the tiles == the tensor's
    yy = y.detach().double()
    assert torch.allclose(rm.double(), 0.1 * yy.mean((0, 1, 2)), atol=1e-6)
    assert torch.allclose(rv.double(), 0.9 + 0.1 * yy.reshape(-1, 64).var(0, unbiased=True), rtol=1e-5)


def test_decoder_tail_gates_are_bit_identical():
    """conv3x3(32 -> 16) + ELU -> upconv16 + ELU -> disp_head (models/depth_head.py:60-77): the two ELU backward passes ride in
    upconv16's and disp_head's data-gradient kernels; everything but the bias gradients (another order of the same column sums) bit for bit"""
    F = _F()
    g = torch.Generator().manual_seed(21)
    x = _nhwc(g, 2, 32, 64, 32)
    w0 = (torch.randn(16, 32, 3, 3, generator=g) / 17).to(DEV).requires_grad_()
    w1 = (torch.randn(16, 16, 3, 3, generator=g) / 12).to(DEV).requires_grad_()
    wd = (torch.randn(1, 16, 3, 3, generator=g) / 12).to(DEV).requires_grad_()
    b0, b1 = (torch.randn(16, generator=g) * 0.1).to(DEV).requires_grad_(), (torch.randn(16, generator=g) * 0.1).to(DEV).requires_grad_()
    bd = torch.zeros(1, device=DEV).requires_grad_()
    res = {}
    for excl in (False, True):
        xi = x.clone().requires_grad_()
        y0 = F.conv2d(xi, w0, b0, pad=1, act="elu")
        y1 = F.upconv16(y0, w1, b1, act="elu", exclusive=excl)
        depth, disp = F.disp_head(y1, wd, bd, exclusive=excl)
        gy = torch.randn(depth.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
        res[excl] = [depth.detach()] + list(torch.autograd.grad((depth * gy).sum() + disp.sum(), [xi, w0, w1, wd, bd, b0, b1]))
    _same(res, 2)


@pytest.mark.parametrize("h,w,nms", [(512, 1024, 7), (512, 512, 3), (64, 96, 7)])
def test_panoptic_centre_selection_on_dense_candidates(h, w, nms):
    """misc/post_processing_panoptic.py:31-92 on a heat map of white noise: ~H W / nms^2 local maxima, i.e. 10 k candidates through the
    compact candidate list (round 5) at 512 x 1024 / 7, 29 k -- more than the list holds: the pixel-walk selection -- at 512 x 512 / 3.
    The 200 kept centres, in row-major order, bit for bit against the oracle."""
    import numpy as np
    F = _F()
    from oracle import panoptic
    g = torch.Generator().manual_seed(h + nms)
    ctr = torch.rand(1, 1, h, w, generator=g) * 0.8 + 0.15
    sem = torch.full((1, h, w), 11, dtype=torch.uint8)
    off = torch.zeros(1, 2, h, w)
    pan, inst, centers, state = F.panoptic_fusion(sem.to(DEV), ctr.to(DEV), off.to(DEV), [9, 10, 11, 12, 13], nms_kernel=nms)
    pts = panoptic.find_centers(ctr[0, 0].numpy(), 0.1, nms, 200)
    k = int(state[0, 1])
    assert k == len(pts) and (int(state[0, 0]) > 200 or h * w < 49 * 200)
    assert np.array_equal(centers[0, :k].cpu().numpy(), np.asarray(pts))


@pytest.mark.parametrize("groups", [False, True])
def test_stem_statistics_from_the_halo_kernel_epilogue(groups):
    """Train-mode stem of the pose encoder (models/resnet_encoder.py:118-121): conv_stem74f_kernel leaves (shift, sum, sum of squares)
    per 8 x 32 tile and codeps_bn_finalize_tiles + the pooled pass (running statistics riding in it) replace the statistics pass over
    the stem's output.  Against the classical statistics pass (CODEPS_EPI_STATS=0's path): same batch statistics up to fp32 summation
    order -> running statistics within 1e-6, last feature map / parameter gradients within 2e-3 / 5e-3 relative L2 (every BatchNorm
    of the reference run takes the classical pass; a ReLU may flip for a few elements at zero); the tile statistics themselves against
    float64 statistics of the stem's output."""
    import copy
    import ctypes
    from codeps_amd import models
    from codeps_amd._lib import lib
    F = _F()
    torch.manual_seed(11)
    enc = models.ResnetEncoder(18, False, num_input_images=2).to(DEV).train()
    enc.only_last_feature = True
    ref = copy.deepcopy(enc)
    g = torch.Generator().manual_seed(6)
    frames = [torch.rand(2, 3, 128, 256, generator=g).to(DEV) for _ in range(4)]          # stem output 64 x 128: whole 8 x 32 tiles
    lib().codeps_conv2d_last_kernel.restype = ctypes.c_char_p
    outs, gy = [], None
    seen = {}
    orig = F.stem_conv2d

    def spy(*a, **k):
        z = orig(*a, **k)
        seen["kernel"] = _kfam(lib().codeps_conv2d_last_kernel().decode())
        seen["stats"] = getattr(z, "_codeps_stats", None)
        return z
    keep = F._EPI_PARTS["stats"]
    try:
        for net, on in ((ref, False), (enc, True)):
            F._EPI_PARTS["stats"] = on and keep
            F.stem_conv2d = spy if on else orig
            last = (net.forward_pair_groups([(frames[0], frames[1]), (frames[2], frames[3])]) if groups else net((frames[0], frames[1])))[-1]
            if gy is None:
                gy = torch.randn(last.shape, generator=torch.Generator().manual_seed(9)).to(DEV)
            last.backward(gy)
            outs.append((last.detach().clone(), [p.grad.clone() for p in net.parameters() if p.grad is not None],
                         net.encoder.bn1.running_mean.clone(), net.encoder.bn1.running_var.clone()))
    finally:
        F._EPI_PARTS["stats"] = keep
        F.stem_conv2d = orig
    if not keep:
        pytest.skip("CODEPS_EPI_STATS=0")
    assert seen["kernel"] == "conv_stem74f_kernel" and seen["stats"] is not None and seen["stats"]["rows"] == 256
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))        # noqa: E731
    a, b = outs
    assert rel(b[0], a[0]) <= 2e-3
    assert rel(b[2], a[2]) <= 1e-6 and rel(b[3], a[3]) <= 1e-6
    assert len(a[1]) == len(b[1]) >= 60
    worst = max(rel(gb, ga) for ga, gb in zip(a[1], b[1]))
    assert worst <= 5e-3, worst
    # and against float64 statistics of the stem's output itself
    with torch.no_grad():
        x = F.frame_pairs_to_nhwc([(frames[0], frames[1])], 8, F.STEM_XPAD)
        z = orig(x, enc.encoder.conv1.weight, bn_stats=True)
        st = z._codeps_stats
        mean = torch.empty(1, 64, device=DEV)
        var, inv = torch.empty_like(mean), torch.empty_like(mean)
        ngi, gb0, gb1 = F._img_groups([(0, 2)])
        F.check(lib().codeps_bn_finalize_tiles(F.ptr(st["part"]), st["tile0"], st["tpi"], st["rows"], ngi, gb0, gb1, 64, 1e-5, F.ptr(mean), F.ptr(var),
                                               F.ptr(inv), F.stream_ptr()), "finalize")
        zz = z.double().reshape(-1, 64)
        assert torch.allclose(mean[0].double(), zz.mean(0), rtol=1e-5, atol=1e-7)
        assert torch.allclose(var[0].double(), zz.var(0, unbiased=False), rtol=1e-5)


@pytest.mark.parametrize("case", [(2, 16, 32, 512, 512, 3, True), (4, 32, 64, 1024, 256, 1, False), (4, 16, 32, 2048, 512, 1, True)])
def test_extras_ride_in_the_split_k_finish(case):
    """Small-M, deep-K layers (ResNet-50/101 layer3 / layer4 of the pre-training configurations, models/resnet_encoder.py:115-126) are cut
    into K slices; their epilogue extras then run in the finish pass (splitk_finish_ex_kernel, 64-row tiles): forward statistics against
    float64 statistics of the output, output bit-identical to the launch without extras; BatchNorm + ReLU gate of an exclusive
    consumer's data gradient: the hand-off carries (sum g, sum g xhat) partials, gradients against the separate passes."""
    import ctypes
    from codeps_amd._lib import lib
    F = _F()
    b, h, w, ci, co, k, dgrad_sliced = case               # (case 1: the consumer's K = 256 is too short for K slices -- its own epilogue gates)
    lib().codeps_conv2d_last_finish_kernel.restype = ctypes.c_char_p
    g = torch.Generator().manual_seed(ci + co)
    x = _nhwc(g, b, h, w, ci)
    w0 = (torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5).to(DEV).requires_grad_()
    y0 = F.conv2d(x, w0, None, pad=k // 2)
    y = F.conv2d(x, w0, None, pad=k // 2, bn_stats=True)
    assert lib().codeps_conv2d_last_finish_kernel().decode() == "splitk_finish_ex_kernel"
    st = y._codeps_stats
    assert st is not None and st["rows"] == 64 and torch.equal(y, y0)
    mean = torch.empty(1, co, device=DEV)
    var, inv = torch.empty_like(mean), torch.empty_like(mean)
    ngi, gb0, gb1 = F._img_groups([(0, b)])
    F.check(lib().codeps_bn_finalize_tiles(F.ptr(st["part"]), st["tile0"], st["tpi"], st["rows"], ngi, gb0, gb1, co, 1e-5, F.ptr(mean), F.ptr(var),
                                           F.ptr(inv), F.stream_ptr()), "finalize")
    yy = y.detach().double().reshape(-1, co)
    assert torch.allclose(mean[0].double(), yy.mean(0), rtol=1e-5, atol=1e-6)
    assert torch.allclose(var[0].double(), yy.var(0, unbiased=False), rtol=1e-5)
    # BatchNorm + ReLU behind it, an exclusive consumer of the same depth: its data gradient is cut into K slices too
    w1 = (torch.randn(co, co, k, k, generator=g) / (co * k * k) ** 0.5).to(DEV).requires_grad_()
    res = {}
    for mode in ("separate", "fused"):
        F._EPI = mode == "fused"
        try:
            gamma, beta = torch.full((co,), 1.2, device=DEV).requires_grad_(), torch.full((co,), 0.1, device=DEV).requires_grad_()
            rm, rv = torch.zeros(co, device=DEV), torch.ones(co, device=DEV)
            xi = x.clone().requires_grad_()
            z = F.batch_norm_train(F.conv2d(xi, w0, None, pad=k // 2, bn_stats=True), gamma, beta, rm, rv, act="relu")
            o = F.conv2d(z, w1, None, pad=k // 2, exclusive=True)
            gy = torch.randn(o.shape, generator=torch.Generator().manual_seed(3)).to(DEV)
            seen = {}
            orig = F._Gate.take

            def spy(self, dy, _o=orig, _s=seen):
                if self.kind == "bn":
                    _s["parts"] = self.handoff is not None and self.handoff["part"] is not None
                    _s["rows"] = self.handoff["segs"][0][2] if _s["parts"] else 0
                    _s["finish"] = lib().codeps_conv2d_last_finish_kernel().decode()
                return _o(self, dy)
            F._Gate.take = spy
            try:
                res[mode] = torch.autograd.grad(o, [xi, w0, w1, gamma, beta], gy)
            finally:
                F._Gate.take = orig
            if mode == "fused":
                assert seen.get("parts") is True and seen.get("finish") == "splitk_finish_ex_kernel", seen
                assert (seen["rows"] == 64) == dgrad_sliced or seen["rows"] == 64, seen
        finally:
            F._EPI = True
    rel = lambda a, r: float((a.double() - r.double()).norm() / (r.double().norm() + 1e-30))      # noqa: E731
    for i, (a, r) in enumerate(zip(res["fused"], res["separate"])):
        assert rel(a, r) <= 5e-3, (i, rel(a, r))           # (statistics in another summation order: a ReLU at ~0 may flip)


@pytest.mark.parametrize("case", [
    # (batch, H, W, ci, co, stride): the bottleneck 1 x 1 convolutions of the pre-training configurations (models/resnet_encoder.py:115-126)
    (4, 128, 256, 64, 256, 1), (4, 128, 256, 256, 64, 1), (4, 32, 64, 1024, 256, 1), (4, 16, 32, 512, 2048, 1),
    (4, 64, 128, 256, 512, 2),              # a down-sampling shortcut (stride 2)
    (3, 30, 44, 96, 80, 1), (2, 33, 47, 72, 200, 2), (2, 24, 40, 320, 136, 1),      # ragged: channels that fill no tile, pixel counts that fill no K-step
])
def test_weight_gradient_of_1x1_convolutions_on_the_split_arithmetic(case):
    """`wgrad_1x1s_kernel` (3-piece bf16 operands, pixels as the GEMM's K) against float64: within 3 x torch's own fp32 error, bit-identical
    from launch to launch, and the kernel is the one that ran."""
    import ctypes
    import torch.nn.functional as TF
    from codeps_amd._lib import lib
    F = _F()
    b, h, w, ci, co, stride = case
    g = torch.Generator().manual_seed(ci + co + h)
    x = torch.relu(torch.randn(b, ci, h, w, generator=g))
    wt = torch.randn(co, ci, 1, 1, generator=g) * (2.0 / ci) ** 0.5
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    gy = torch.randn(b, co, ho, wo, generator=g)

    def ref(dtype):
        ww = wt.to(dtype).requires_grad_(True)
        TF.conv2d(x.to(dtype), ww, stride=stride).backward(gy.to(dtype))
        return ww.grad

    dw64, dw32 = ref(torch.float64), ref(torch.float32)
    lib().codeps_conv2d_last_kernel.restype = ctypes.c_char_p
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    if xn.shape[-1] % 16:
        xn = torch.nn.functional.pad(xn, (0, 16 - xn.shape[-1] % 16))
    outs = []
    for _ in range(3):
        ww = wt.to(DEV).detach().requires_grad_(True)
        y = F.conv2d(xn, ww, None, stride=stride)
        gyn = torch.zeros_like(y)
        gyn[..., :co] = gy.permute(0, 2, 3, 1).to(DEV)
        y.backward(gyn)
        assert _kfam(lib().codeps_conv2d_last_kernel().decode()) == "wgrad_1x1s_kernel"
        outs.append(ww.grad.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    l2 = lambda a, r: float((a.double() - r).norm() / r.norm())       # noqa: E731
    e, t = l2(outs[0], dw64), l2(dw32, dw64)
    assert e <= 3 * t + 1e-9, f"{case}: {e:.3e} vs torch fp32 {t:.3e}"


@pytest.mark.parametrize("case", [
    # (batch, H, W, ci, co): the depth decoder's full-resolution tail (models/depth_head.py:60-71) and its data gradients
    (2, 64, 128, 16, 16), (2, 64, 128, 32, 16), (1, 32, 64, 16, 32), (3, 8, 32, 32, 16), (1, 128, 256, 16, 16),
])
def test_16_channel_3x3_layers_on_the_split_arithmetic(case):
    """`conv_halo3n16_kernel` (K-steps of two taps x 16 channels on v_mfma_f32_16x16x32_bf16, weights as register fragments): forward
    with bias + ELU and the data gradient against float64 within 3 x torch's own fp32 error, bit-identical from launch to launch, the
    kernel is the one that ran (forward AND data gradient), and 12 real output channels in a 16-channel tensor."""
    import ctypes
    import torch.nn.functional as TF
    from codeps_amd._lib import lib
    F = _F()
    b, h, w, ci, co = case
    co_real = co - 4 if co == 16 else co
    g = torch.Generator().manual_seed(ci * 3 + co + h)
    x = torch.randn(b, ci, h, w, generator=g)
    wt = torch.randn(co_real, ci, 3, 3, generator=g) * (2.0 / (9 * ci)) ** 0.5
    bias = torch.randn(co_real, generator=g) * 0.1
    gy = torch.randn(b, co_real, h, w, generator=g)

    def ref(dtype):
        xx = x.to(dtype).requires_grad_(True)
        y = TF.elu(TF.conv2d(xx, wt.to(dtype), bias.to(dtype), padding=1))
        y.backward(gy.to(dtype))
        return y.detach(), xx.grad

    (y64, dx64), (y32, dx32) = ref(torch.float64), ref(torch.float32)
    lib().codeps_conv2d_last_kernel.restype = ctypes.c_char_p
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV)       # noqa: E731
    outs = []
    for _ in range(2):
        xn = nhwc(x).requires_grad_(True)
        y = F.conv2d(xn, wt.to(DEV), bias.to(DEV), pad=1, act="elu")
        assert _kfam(lib().codeps_conv2d_last_kernel().decode()) == "conv_halo3n16_kernel"
        gyn = torch.zeros_like(y)
        gyn[..., :co_real] = nhwc(gy)
        dx, = torch.autograd.grad(y, xn, gyn)
        outs.append((y[..., :co_real].detach().permute(0, 3, 1, 2).cpu(), dx.permute(0, 3, 1, 2).cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float(y[..., co_real:].abs().max()) == 0.0 if co_real < co else True
    l2 = lambda a, r: float((a.double() - r).norm() / r.norm())       # noqa: E731
    e, t = l2(outs[0][0], y64), l2(y32, y64)
    assert e <= 3 * t + 1e-9, f"{case} forward: {e:.3e} vs torch fp32 {t:.3e}"
    e, t = l2(outs[0][1], dx64), l2(dx32, dx64)
    assert e <= 3 * t + 1e-9, f"{case} data gradient: {e:.3e} vs torch fp32 {t:.3e}"
    # the data gradient alone (no weight gradient launch after it): its kernel
    xn = nhwc(x).requires_grad_(True)
    y = F.conv2d(xn, wt.to(DEV).requires_grad_(False), None, pad=1)
    torch.autograd.grad(y, xn, torch.ones_like(y))
    assert _kfam(lib().codeps_conv2d_last_kernel().decode()) == "conv_halo3n16_kernel"
