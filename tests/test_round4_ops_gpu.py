"""Round-4 entry points against torch / against the paths they replace: channel arg-max, row concatenation, multi-counter add, the
data gradient with a second gradient added in its epilogue (conv2d fork output), panoptic fusion on strided maps."""
import numpy as np
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


@pytest.mark.parametrize("c,layout", [(14, "nhwc_view"), (2, "nhwc_view"), (19, "nchw"), (14, "ties")])
def test_channel_argmax_equals_torch(c, layout):
    F = _F()
    g = torch.Generator().manual_seed(c)
    b, h, w = 2, 37, 53
    if layout == "nchw":
        x = torch.randn(b, c, h, w, generator=g).to(DEV)
    else:
        cp = (c + 15) // 16 * 16
        base = torch.randn(b, h, w, cp, generator=g)
        if layout == "ties":
            base = torch.round(base * 2) / 2                       # many exact ties: the FIRST maximal index must win
        base[..., c:] = 100.0                                      # padding lanes must never be compared
        x = F.to_nchw(base.to(DEV), c)
    want = torch.argmax(x, dim=1)
    got = F.channel_argmax(x)
    assert got.dtype == torch.int64 and torch.equal(got, want)
    assert torch.equal(got._codeps_u8[0], want.to(torch.uint8)) and got._codeps_u8[1] == got._version
    assert torch.equal(F.channel_argmax(x.detach(), torch.uint8), want.to(torch.uint8))


@pytest.mark.parametrize("sizes", [[(3, 8, 16)], [(1, 5, 7), (2, 5, 7), (4, 5, 7)], [(2, 4)] * 8, [(1, 3)] * 11])
def test_concat_rows_equals_torch_cat(sizes):
    F = _F()
    g = torch.Generator().manual_seed(len(sizes))
    parts = [torch.randn(*s, generator=g).to(DEV) for s in sizes]
    assert torch.equal(F.concat_rows(parts), torch.cat(parts, 0))


def test_counters_add():
    F = _F()
    cs = [torch.full((), i, dtype=torch.int64, device=DEV) for i in range(70)]
    F.counters_add(cs, [i % 5 + 1 for i in range(70)])
    torch.cuda.synchronize()
    assert [int(c) for c in cs] == [i + i % 5 + 1 for i in range(70)]


@pytest.mark.parametrize("shape", [(4, 32, 64, 64, 64, 3, 1), (2, 32, 64, 256, 256, 3, 1), (3, 32, 64, 64, 128, 3, 2), (3, 32, 64, 64, 128, 1, 2),
                                   (2, 16, 32, 256, 64, 1, 1), (1, 32, 64, 32, 16, 3, 1),
                                   (1, 16, 32, 512, 512, 3, 1), (2, 16, 32, 512, 2048, 1, 1)])      # cut into K slices: the finish pass adds
def test_forked_conv_gradient_equals_autograd_sum(shape):
    """y, x' = conv2d(x, fork=True); the gradient arriving through x' is added inside the data-gradient kernel: same bits as autograd's
    own sum of the two gradients (one fp32 add per element either way), for the halo, generic, stride-2 parity-class and 16-column kernels
    and for launches cut into K slices (the finish pass adds)."""
    F = _F()
    b, h, w, ci, co, k, st = shape
    g = torch.Generator().manual_seed(ci + co + k)
    x = torch.randn(b, h, w, ci, generator=g).to(DEV)
    wt = (torch.randn(co, ci, k, k, generator=g) * 0.05).to(DEV)
    other = torch.randn(b, h, w, ci, generator=g).to(DEV)             # d loss / d x through the second consumer
    gy = None
    grads = []
    for fork in (True, False):
        xx = x.clone().requires_grad_(True)
        if fork:
            y, xa = F.conv2d(xx, wt, stride=st, pad=k // 2, fork=True)
        else:
            y, xa = F.conv2d(xx, wt, stride=st, pad=k // 2), xx
        if gy is None:
            gy = torch.randn_like(y)
        ((y * gy).sum() + (xa * other).sum()).backward()
        grads.append(xx.grad.clone())
    torch.cuda.synchronize()
    assert torch.equal(grads[0], grads[1]), f"max |diff| {(grads[0] - grads[1]).abs().max().item():.3e}"
    if b * h * w <= 1024 and co >= 512:
        import ctypes
        from codeps_amd._lib import lib
        lib().codeps_conv2d_last_finish_kernel.restype = ctypes.c_char_p
        assert lib().codeps_conv2d_last_finish_kernel().decode().startswith("splitk_finish")


def test_panoptic_fusion_reads_strided_maps_in_place():
    """Centre / offset maps as channels of padded NHWC tensors (what the instance head hands over) == the same maps as contiguous NCHW
    planes, bit for bit; an int64 semantic map with its uint8 twin needs no conversion and returns an int64 instance map."""
    F = _F()
    g = torch.Generator().manual_seed(5)
    b, h, w = 2, 96, 160
    ctr_nhwc = torch.rand(b, h, w, 16, generator=g).to(DEV)
    off_nhwc = (torch.randn(b, h, w, 16, generator=g) * 6).to(DEV)
    center, offset = F.to_nchw(ctr_nhwc, 1), F.to_nchw(off_nhwc, 2)
    probs = F.to_nchw(torch.rand(b, h, w, 16, generator=g).to(DEV), 14)
    sem = F.channel_argmax(probs)
    a = F.panoptic_fusion(sem, center, offset, [9, 10, 11, 12, 13], void_label=255)
    c = F.panoptic_fusion(sem.to(torch.uint8), center.contiguous(), offset.contiguous(), [9, 10, 11, 12, 13], void_label=255)
    torch.cuda.synchronize()
    assert a[1].dtype == torch.int64 and c[1].dtype == torch.uint8
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1].to(torch.int64)) and torch.equal(a[2], c[2]) and torch.equal(a[3], c[3])
    assert int(a[3][:, 1].min()) > 0, "the case must keep some centres"


@pytest.mark.parametrize("case", [
    # (batch, H, W, c0, c1, up0, co): 4 x 16-divisible maps -> the double-buffered FAST body of wgrad_halo64s_kernel
    (2, 16, 32, 64, 0, False, 64),        # one chunk, one 64-row co tile
    (1, 32, 48, 128, 0, False, 128),      # 2 x 2 tiles of channels
    (2, 16, 32, 80, 0, False, 64),        # second chunk holds 16 of 64 channels (the rest reads zeros through the range check)
    (1, 32, 64, 64, 64, True, 64),        # chunk 0 from the up-sampled source (each source pixel twice), chunk 1 from the skip
    (1, 16, 32, 64, 48, False, 96),       # concat without up-sampling, ragged skip width, co 96 (second co tile half empty)
    (1, 18, 32, 64, 0, False, 64),        # H % 4 != 0 -> the generic body (same kernel name, same bound)
    (1, 16, 32, 32, 32, True, 64),        # ONE chunk: half 0 from the up-sampled source, half 1 from the skip (per-wave sources)
    (2, 16, 32, 64, 0, False, 32),        # <= 32 output channels: the CO32 form of the double-buffered body (waves split the rows)
    (1, 32, 32, 32, 64, True, 32),        # CO32 + a chunk split at 32 (the decoders' half-resolution stage) + a half-empty chunk
])
def test_weight_gradient_fast_bodies_against_float64(case):
    """wgrad_halo64s_kernel picks its body per block (FAST: buffer loads with constant per-thread offsets, double-buffered 2 x 16
    sub-tiles; UP form for an up-sampled chunk; generic otherwise): every form against a float64 convolution gradient
    (reference arithmetic: fp32 convs, models/resnet_encoder.py:115-126), and bit-identical from launch to launch."""
    import torch.nn.functional as TF
    from codeps_amd import functional as F
    from codeps_amd._lib import lib
    b, h, w, c0, c1, up0, co = case
    g = torch.Generator().manual_seed(sum(case))
    x0 = torch.relu(torch.randn(b, c0, h // 2 if up0 else h, w // 2 if up0 else w, generator=g))
    x1 = torch.relu(torch.randn(b, c1, h, w, generator=g)) if c1 else None
    wt = torch.randn(co, c0 + c1, 3, 3, generator=g) * (2.0 / (9 * (c0 + c1))) ** 0.5
    gy = torch.randn(b, co, h, w, generator=g)
    full = TF.interpolate(x0, scale_factor=2, mode="nearest") if up0 else x0
    if c1:
        full = torch.cat([full, x1], 1)

    def ref(dtype):
        ww = wt.to(dtype).requires_grad_(True)
        TF.conv2d(full.to(dtype), ww, padding=1).backward(gy.to(dtype))
        return ww.grad

    dw64, dw32 = ref(torch.float64), ref(torch.float32)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV)       # noqa: E731
    lib().codeps_conv2d_last_kernel.restype = __import__("ctypes").c_char_p
    outs = []
    for _ in range(3):
        ww = wt.to(DEV).detach().requires_grad_(True)
        y = F.conv2d(nhwc(x0), ww, x1=nhwc(x1) if c1 else None, pad=1, up0=up0)
        y.backward(nhwc(gy) if y.shape[-1] == co else torch.nn.functional.pad(nhwc(gy), (0, y.shape[-1] - co)))
        assert _kfam(lib().codeps_conv2d_last_kernel().decode()).startswith("wgrad_halo64s_kernel")
        outs.append(ww.grad.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    l2 = lambda a, r: float((a.double() - r).norm() / r.norm())       # noqa: E731
    e, t = l2(outs[0], dw64), l2(dw32, dw64)
    assert e <= 3 * t + 1e-9, f"{case}: {e:.3e} vs torch fp32 {t:.3e}"


@pytest.mark.parametrize("shape", [(4, 128, 256), (1, 72, 96), (2, 8, 32)])
def test_stem_weight_gradient_double_buffered_against_float64(shape):
    """ResNet conv1 (7x7 / 2 / 3; models/resnet_encoder.py:115) in the double-pixel layout: the double-buffered weight gradient
    (`wgrad_stem74d_kernel`, whole 4 x 16 tiles) against float64; block tile counts even, odd (72 x 96: 27 tiles in blocks of 8) and
    one (8 x 32); bit-identical from launch to launch."""
    import torch.nn.functional as TF
    from codeps_amd import functional as F
    b, h, w = shape
    g = torch.Generator().manual_seed(h + w)
    x = torch.randn(b, 6, h, w, generator=g)
    wt = torch.randn(64, 6, 7, 7, generator=g) * (2.0 / (49 * 6)) ** 0.5
    gy = torch.randn(b, 64, h // 2, w // 2, generator=g)

    def ref(dtype):
        ww = wt.to(dtype).requires_grad_(True)
        TF.conv2d(x.to(dtype), ww, stride=2, padding=3).backward(gy.to(dtype))
        return ww.grad

    dw64, dw32 = ref(torch.float64), ref(torch.float32)
    frames = [x[:, i:i + 3].contiguous().to(DEV) for i in (0, 3)]
    outs = []
    for _ in range(3):
        ww = wt.to(DEV).detach().requires_grad_(True)
        y = F.stem_conv2d(F.images_to_nhwc(frames, 8, F.STEM_XPAD), ww)
        y.backward(gy.permute(0, 2, 3, 1).contiguous().to(DEV))
        outs.append(ww.grad.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    l2 = lambda a, r: float((a.double() - r).norm() / r.norm())       # noqa: E731
    e, t = l2(outs[0], dw64), l2(dw32, dw64)
    assert e <= 3 * t + 1e-9, f"{shape}: {e:.3e} vs torch fp32 {t:.3e}"


@pytest.mark.parametrize("case", [
    # (batch, H, W, ci, co): 3 x 3 / stride 2 / pad 1 on maps whose OUTPUT is whole 4 x 16 tiles -> wgrad_s2_halo_kernel
    (2, 32, 64, 64, 128),        # ResNet-18 layer2.0.conv1 at a small size
    (1, 8, 32, 32, 64),          # one tile, one 32-channel chunk
    (3, 16, 32, 128, 256),       # 4 chunks x 4 co groups, odd tile count per block
    (1, 24, 32, 96, 80),         # co 80: second co group a quarter full; 3 output tile rows
])
def test_stride2_weight_gradient_against_float64(case):
    """The first convolution of ResNet-18/34 stages 2-4 (3 x 3, stride 2; models/resnet_encoder.py:115-126 -> torchvision BasicBlock):
    its weight gradient on the bf16 matrix cores (input columns de-interleaved by parity in LDS, double-buffered sub-tiles) against a
    float64 gradient, and bit-identical from launch to launch."""
    import ctypes
    import torch.nn.functional as TF
    from codeps_amd import functional as F
    from codeps_amd._lib import lib
    b, h, w, ci, co = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.relu(torch.randn(b, ci, h, w, generator=g))
    wt = torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (9 * ci)) ** 0.5
    gy = torch.randn(b, co, h // 2, w // 2, generator=g)

    def ref(dtype):
        ww = wt.to(dtype).requires_grad_(True)
        TF.conv2d(x.to(dtype), ww, stride=2, padding=1).backward(gy.to(dtype))
        return ww.grad

    dw64, dw32 = ref(torch.float64), ref(torch.float32)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV)       # noqa: E731
    lib().codeps_conv2d_last_kernel.restype = ctypes.c_char_p
    outs = []
    for _ in range(3):
        ww = wt.to(DEV).detach().requires_grad_(True)
        y = F.conv2d(nhwc(x), ww, stride=2, pad=1)
        y.backward(nhwc(gy))
        assert _kfam(lib().codeps_conv2d_last_kernel().decode()) == "wgrad_s2_halo_kernel"
        outs.append(ww.grad.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    l2 = lambda a, r: float((a.double() - r).norm() / r.norm())       # noqa: E731
    e, t = l2(outs[0], dw64), l2(dw32, dw64)
    assert e <= 3 * t + 1e-9, f"{case}: {e:.3e} vs torch fp32 {t:.3e}"


@pytest.mark.parametrize("groups,hw", [(False, (64, 96)), (True, (64, 96)), (False, (36, 52))])
def test_fused_stem_bn_relu_pool_is_bit_identical(groups, hw):
    """Train-mode stem of the pose encoder (models/resnet_encoder.py:118-121: bn1 -> relu -> maxpool) as one pass each way
    (`ResnetEncoder.only_last_feature`, C ABI codeps_bn_relu_pool_{fwd,bwd}) against the separate BatchNorm and max-pool ops: last
    feature map, running statistics and every parameter gradient bit for bit; with and without per-pair statistic groups."""
    import copy
    from codeps_amd import models
    torch.manual_seed(3)
    enc = models.ResnetEncoder(18, False, num_input_images=2).to(DEV).train()
    ref = copy.deepcopy(enc)
    enc.only_last_feature = True
    g = torch.Generator().manual_seed(5)
    frames = [torch.rand(2, 3, hw[0], hw[1], generator=g).to(DEV) for _ in range(4)]        # (36 x 52: stem output 18 x 26; 2 x 2 form too)
    gy = None
    outs = []
    for net in (ref, enc):
        if groups:
            feats = net.forward_pair_groups([(frames[0], frames[1]), (frames[2], frames[3])])
        else:
            feats = net((frames[0], frames[1]))
        last = feats[-1]
        if gy is None:
            gy = torch.randn(last.shape, generator=torch.Generator().manual_seed(9)).to(DEV)
        last.backward(gy)
        outs.append((last.detach().clone(), [None if p.grad is None else p.grad.clone() for p in net.parameters()],
                     net.encoder.bn1.running_mean.clone(), net.encoder.bn1.running_var.clone(), int(net.encoder.bn1.num_batches_tracked)))
    assert enc((frames[0], frames[1]))[0] is None and ref((frames[0], frames[1]))[0] is not None
    a, b = outs
    assert torch.equal(a[0], b[0])
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and a[4] == b[4]
    assert sum(ga is not None for ga in a[1]) >= 60
    for ga, gb in zip(a[1], b[1]):
        assert (ga is None) == (gb is None) and (ga is None or torch.equal(ga, gb))


@pytest.mark.parametrize("case", [
    # (batch, H, W, c0, c1, up0, co): <= 32 output channels on whole 8 x 32 tiles, >= 256 tiles -> conv_halo3_kernel in its 4 x 1 wave form
    (4, 128, 128, 64, 0, False, 32),
    (4, 128, 128, 32, 64, True, 32),       # the decoders' half-resolution stage: nearest x2 of 32 channels ++ 64 skip channels
    (2, 128, 256, 96, 0, False, 24),       # 24 real output channels (padded to 32)
])
def test_halo_kernel_with_32_output_columns_against_float64(case):
    """3 x 3 convolutions with <= 32 output channels (models/depth_head.py:60-71 at the half-resolution stage) on the halo-tile kernel:
    forward against float64 within 3 x torch's own fp32 error, launch-to-launch identical."""
    import ctypes
    import torch.nn.functional as TF
    from codeps_amd import functional as F
    from codeps_amd._lib import lib
    b, h, w, c0, c1, up0, co = case
    g = torch.Generator().manual_seed(sum(case))
    x0 = torch.relu(torch.randn(b, c0, h // 2 if up0 else h, w // 2 if up0 else w, generator=g))
    x1 = torch.relu(torch.randn(b, c1, h, w, generator=g)) if c1 else None
    wt = torch.randn(co, c0 + c1, 3, 3, generator=g) * (2.0 / (9 * (c0 + c1))) ** 0.5
    bias = torch.randn(co, generator=g) * 0.1
    full = TF.interpolate(x0, scale_factor=2, mode="nearest") if up0 else x0
    if c1:
        full = torch.cat([full, x1], 1)
    y64 = TF.elu(TF.conv2d(full.double(), wt.double(), bias.double(), padding=1))
    y32 = TF.elu(TF.conv2d(full, wt, bias, padding=1))
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV)       # noqa: E731
    lib().codeps_conv2d_last_kernel.restype = ctypes.c_char_p
    outs = []
    for _ in range(2):
        y = F.conv2d(nhwc(x0), wt.to(DEV), bias.to(DEV), x1=nhwc(x1) if c1 else None, pad=1, up0=up0, act="elu")
        assert _kfam(lib().codeps_conv2d_last_kernel().decode()) == "conv_halo3_kernel<32 columns>"
        outs.append(y[..., :co].permute(0, 3, 1, 2).cpu())
    assert torch.equal(outs[0], outs[1])
    l2 = lambda a, r: float((a.double() - r).norm() / r.norm())       # noqa: E731
    e, t = l2(outs[0], y64), l2(y32, y64)
    assert e <= 3 * t + 1e-9, f"{case}: {e:.3e} vs torch fp32 {t:.3e}"


@pytest.mark.parametrize("case", [
    # the pose encoder's layers at the size of BASELINE configs[2] (12 frame pairs): (kind, batch, H, W, ci, co)
    ("s1", 12, 128, 256, 64, 64),          # layer1: wgrad_halo64s_kernel, double-buffered body
    ("s1", 12, 32, 64, 256, 256),          # layer3
    ("s2", 12, 128, 256, 64, 128),         # layer2.0.conv1: wgrad_s2_halo_kernel
    ("stem", 12, 512, 1024, 6, 64),        # conv1: conv_stem74f_kernel + wgrad_stem74d_kernel
])
def test_full_size_weight_gradients_agree_with_the_f32_mfma_kernels(case):
    """At the headline workload's own sizes the split-operand weight-gradient kernels of this round against this library's f32-MFMA
    kernels on the same tensors (exact fp32 products, the yardstick of tests/test_precision_gpu.py): relative L2 difference at the
    level of fp32 summation order, and linear in the incoming gradient."""
    import ctypes
    from codeps_amd import functional as F
    from codeps_amd._lib import lib
    kind, b, h, w, ci, co = case
    g = torch.Generator().manual_seed(h + ci)
    lib().codeps_conv2d_last_kernel.restype = ctypes.c_char_p
    if kind == "stem":
        frames = [torch.rand(b, 3, h, w, generator=g).to(DEV) for _ in range(2)]
        wt = (torch.randn(co, ci, 7, 7, generator=g) * 0.05).to(DEV)
        gy = torch.randn(b, h // 2, w // 2, co, generator=g).to(DEV)

        def run(wgt, grad):
            ww = wgt.detach().requires_grad_(True)
            y = F.stem_conv2d(F.images_to_nhwc(frames, 8, F.STEM_XPAD), ww)
            y.backward(grad)
            return y.detach(), ww.grad
        expect = "wgrad_stem74s_kernel"
    else:
        x = torch.relu(torch.randn(b, h, w, ci, generator=g)).to(DEV)
        wt = (torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (9 * ci)) ** 0.5).to(DEV)
        st = 2 if kind == "s2" else 1
        gy = torch.randn(b, h // st, w // st, co, generator=g).to(DEV)

        def run(wgt, grad):
            ww = wgt.detach().requires_grad_(True)
            y = F.conv2d(x, ww, stride=st, pad=1)
            y.backward(grad)
            return y.detach(), ww.grad
        expect = "wgrad_s2_halo_kernel" if kind == "s2" else "wgrad_halo64s_kernel"
    y3, dw3 = run(wt, gy)
    assert _kfam(lib().codeps_conv2d_last_kernel().decode()) == expect
    _y, dw3b = run(wt, 0.5 * gy)
    assert torch.equal(dw3b, 0.5 * dw3)                               # scaling by a power of two is exact in every product and sum
    prev = lib().codeps_conv2d_set_arith(0)
    try:
        y0, dw0 = run(wt, gy)
        assert "split" not in _kfam(lib().codeps_conv2d_last_kernel().decode()) and _kfam(lib().codeps_conv2d_last_kernel().decode()) != expect
    finally:
        lib().codeps_conv2d_set_arith(prev)
    l2 = lambda a, r: float((a.double() - r.double()).norm() / r.double().norm())       # noqa: E731
    assert l2(y3, y0) <= 2e-6, l2(y3, y0)
    # (the stem sums 1.57 M pixels per weight: the two kernels' slab / summation orders differ by 5e-6 there, measured)
    assert l2(dw3, dw0) <= (2e-5 if kind == "stem" else 2e-6), l2(dw3, dw0)
