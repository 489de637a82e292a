"""Precision of EVERY split-operand conv kernel variant, pinned per kernel.

The conv engine feeds fp32 operands to the bf16 matrix cores as exact 3-way bf16 splits (six piece products, fp32 accumulate:
csrc/conv.hip).  The claim "indistinguishable from an fp32 convolution" (reference arithmetic: fp32 convs,
models/resnet_encoder.py:115-126) is checked here against float64 for each variant the adaptation step spends its time in --
`conv_halo3_kernel` / `conv_halo3r_kernel` (forward, data gradient, fused up-sample + concat), `conv_igemm_ws_kernel`, `conv_igemm_kernel<split3>`,
the split stem forward (16-channel K-steps), `wgrad_halo64s_kernel`, `wgrad_stem74s_kernel` -- on well-scaled, on
wide-dynamic-range and on cancelling inputs: the error against float64 must stay within 3x of the error torch's own fp32 CPU
convolution makes on the same tensors, or (long accumulation chains) within 1.5x of this library's f32-MFMA kernels on them.  `codeps_conv2d_last_kernel` proves that the shape
really selected the variant under test."""
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _F():
    from codeps_amd import functional as F
    return F


def _last_raw():
    from codeps_amd._lib import lib
    return lib().codeps_conv2d_last_kernel().decode()


def _last():
    """the kernel family of the last launch (without the arithmetic tag: helpers.kernel_family)"""
    from helpers import kernel_family
    return kernel_family(_last_raw())


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(y, c):
    return y[..., :c].permute(0, 3, 1, 2).cpu()


def l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


class _arith:
    """with _arith(0): the conv engine on the f32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32 products, the same k-ordered fp32
    accumulation) -- the yardstick that isolates what the operand split costs."""

    def __init__(self, pieces):
        self.pieces = pieces

    def __enter__(self):
        from codeps_amd._lib import lib
        self.prev = lib().codeps_conv2d_set_arith(self.pieces)

    def __exit__(self, *exc):
        from codeps_amd._lib import lib
        lib().codeps_conv2d_set_arith(self.prev)


@pytest.fixture(autouse=True, params=[16, 3], ids=["f16x2", "bf16x3"])
def split_arith(request):
    """Every test of this module runs under both split-operand arithmetics: 16 = two fp16 pieces, three products, block-local scales (the
    default since round 6), 3 = three bf16 pieces, six products (CODEPS_SPLIT=3).  Kernels without an fp16 form run the bf16 one under 16."""
    with _arith(request.param):
        yield request.param


def _bound(err_torch, err_f32mfma, factor, chain=1.5):
    """An fp32 convolution is allowed its summation order: the reference's (torch CPU: blocked / vectorised sums) and the matrix
    core's (one k-ordered fp32 chain per output, up to 2304 terms here) differ by up to ~4x in error against float64.  The split
    kernels must stay within `factor` x torch's own fp32 error, or -- on long chains -- within 1.5x of this library's f32-MFMA
    kernels, which multiply exactly and accumulate in the same order."""
    return max(factor * err_torch, chain * err_f32mfma) + 1e-9


def _inputs(kind, shape_x, shape_w, seed):
    g = torch.Generator().manual_seed(seed)
    if kind == "normal":                 # post-ReLU activations, He-scaled weights
        x = torch.relu(torch.randn(shape_x, generator=g))
        w = torch.randn(shape_w, generator=g) * (2.0 / (shape_w[1] * shape_w[2] * shape_w[3])) ** 0.5
    elif kind == "wide":                 # magnitudes spread over 2^-12 .. 2^12 (activations) and 2^-6 .. 2^6 (weights)
        x = torch.randn(shape_x, generator=g) * torch.exp2(torch.randint(-12, 13, shape_x, generator=g).float())
        w = torch.randn(shape_w, generator=g) * torch.exp2(torch.randint(-6, 7, shape_w, generator=g).float())
    else:                                # "cancel": near-constant activations against zero-sum filters
        x = 1.0 + 1e-3 * torch.randn(shape_x, generator=g)
        w = torch.randn(shape_w, generator=g)
        w = w - w.mean(dim=(1, 2, 3), keepdim=True)
    return x, w, g


def _reference(x, w, gy, stride, pad, dtype):
    xr, wr = x.detach().clone().to(dtype).requires_grad_(True), w.detach().clone().to(dtype).requires_grad_(True)
    y = TF.conv2d(xr, wr, None, stride, pad)
    y.backward(gy.to(dtype))
    return y.detach(), xr.grad, wr.grad


CASES = {
    # name: (B, H, W, Cin, Cout, k, expected forward kernel, expected dgrad kernel, expected wgrad kernel)
    "halo3_64": (4, 64, 128, 64, 64, 3, "conv_halo3_kernel", "conv_halo3_kernel", "wgrad_halo64s_kernel"),
    "halo3_128": (2, 64, 128, 128, 128, 3, "conv_halo3r_kernel", "conv_halo3r_kernel", "wgrad_halo64s_kernel"),    # 128 blocks of 128 columns: the ring kernel on half the CUs
    "halo3r_256": (2, 64, 128, 128, 256, 3, "conv_halo3r_kernel", None, "wgrad_halo64s_kernel"),      # 128-column tiles: weights by LDS DMA
    "halo3r_dgrad": (2, 64, 128, 256, 256, 3, "conv_halo3r_kernel", "conv_halo3r_kernel", None),
    # grids below one block per CU: the ring kernel on >= 128 CUs without split, or with its channel chunks cut into K slices
    "halo3r_small_grid": (6, 32, 64, 256, 256, 3, "conv_halo3r_kernel", "conv_halo3r_kernel", None),
    "halo3r_splitk": (4, 32, 64, 512, 256, 3, "conv_halo3r_kernel<splitk>", None, None),           # 128 blocks x 2 slices of 8 chunks
    "halo3r_splitk_dgrad": (4, 32, 64, 256, 512, 3, None, "conv_halo3r_kernel<splitk>", None),
    "wgrad_co32": (4, 64, 128, 64, 32, 3, None, None, "wgrad_halo64s_kernel<co32>"),               # <= 32 output channels: pixel-split waves
    "ws_3x3_256": (7, 32, 48, 256, 256, 3, "conv_igemm_ws_kernel", "conv_igemm_ws_kernel", None),      # W % 32 != 0: not a halo-tile shape
    "ws_1x1_2048": (4, 32, 64, 2048, 512, 1, "conv_igemm_ws_kernel", None, None),
    "igemm_1x1": (8, 32, 64, 256, 1024, 1, "conv_igemm_kernel<split3>", "conv_igemm_ws_kernel", None),      # dgrad: K = 1024, one round of 128x128 blocks
    "igemm_1x1_dgrad": (8, 32, 64, 512, 2048, 1, "conv_igemm_kernel<split3>", "conv_igemm_kernel<split3>", None),
}


@pytest.mark.parametrize("kind", ["normal", "wide", "cancel"])
@pytest.mark.parametrize("case", list(CASES))
def test_split_kernels_are_at_fp32_rounding_level(case, kind):
    F = _F()
    torch.set_num_threads(8)
    b, h, w_, ci, co, k, k_fwd, k_dg, k_wg = CASES[case]
    x, wt, g = _inputs(kind, (b, ci, h, w_), (co, ci, k, k), 1000 + len(case) * 7 + len(kind))
    gy = torch.randn(b, co, h, w_, generator=g)
    if kind == "wide":
        gy = gy * torch.exp2(torch.randint(-8, 9, gy.shape, generator=g).float())
    y64, dx64, dw64 = _reference(x, wt, gy, 1, k // 2, torch.float64)
    y32, dx32, dw32 = _reference(x, wt, gy, 1, k // 2, torch.float32)
    wd = wt.to(DEV)
    # forward (+ data gradient only: the weight takes no gradient, so the last launch of backward() is the data gradient)
    dx = nhwc(x).requires_grad_(True)
    y = F.conv2d(dx, wd, pad=k // 2)
    if k_fwd is not None:
        assert _last() == k_fwd, f"forward ran {_last()}, the case is meant to select {k_fwd}"
    y.backward(nhwc(gy))
    if k_dg is not None:
        assert _last() == k_dg, f"data gradient ran {_last()}, the case is meant to select {k_dg}"
    with _arith(0):                    # the same tensors through the f32-MFMA kernels
        fx, fw = nhwc(x).requires_grad_(True), wd.clone().requires_grad_(True)
        fy = F.conv2d(fx, fw, pad=k // 2)
        assert "f32" in _last() or "16" in _last(), _last()
        fy.backward(nhwc(gy))
    m_fwd, m_dx, m_dw = l2(nchw(fy.detach(), co), y64), l2(nchw(fx.grad, ci), dx64), l2(fw.grad.cpu(), dw64)
    # K-sliced variants: the f32-MFMA yardstick of these small-M shapes is itself cut into up to 16 K slices (shorter fp32 chains,
    # smaller error) while the ring kernel uses 2-4 slices of 2304 terms -- same products, longer chains: 2x instead of 1.5x
    ch = 2.0 if "splitk" in case else 1.5
    e_fwd, t_fwd = l2(nchw(y.detach(), co), y64), l2(y32, y64)
    assert e_fwd <= _bound(t_fwd, m_fwd, 3, ch), f"{case}/{kind} forward: {e_fwd:.3e} vs torch fp32 {t_fwd:.3e}, f32 MFMA {m_fwd:.3e}"
    e_dx, t_dx = l2(nchw(dx.grad, ci), dx64), l2(dx32, dx64)
    assert e_dx <= _bound(t_dx, m_dx, 3, ch), f"{case}/{kind} data gradient: {e_dx:.3e} vs torch fp32 {t_dx:.3e}, f32 MFMA {m_dx:.3e}"
    # weight gradient
    ww = wd.clone().requires_grad_(True)
    y2 = F.conv2d(nhwc(x), ww, pad=k // 2)
    y2.backward(nhwc(gy))
    if k_wg is not None:
        assert _last() == k_wg, f"weight gradient ran {_last()}, the case is meant to select {k_wg}"
    e_dw, t_dw = l2(ww.grad.cpu(), dw64), l2(dw32, dw64)
    assert e_dw <= _bound(t_dw, m_dw, 3), f"{case}/{kind} weight gradient ({_last()}): {e_dw:.3e} vs torch fp32 {t_dw:.3e}, f32 MFMA {m_dw:.3e}"


@pytest.mark.parametrize("kind", ["normal", "wide", "cancel"])
@pytest.mark.parametrize("co,kernel", [(64, "conv_halo3_kernel"), (256, "conv_halo3r_kernel")])
def test_halo3_with_fused_upsample_and_concat(kind, co, kernel):
    """The decoder's conv3x3(cat(nearest_x2(x0), x1)) (models/depth_head.py:60-71) on the halo-tile kernels (64 columns: weights
    staged through registers; 128 columns: weights by LDS DMA)."""
    F = _F()
    b, h, w_, c0, c1 = 4, 64, 128, 64, (32 if co == 64 else 64)        # 36 K-steps for the wide case: 128-column tiles
    x0, wt, g = _inputs(kind, (b, c0, h // 2, w_ // 2), (co, c0 + c1, 3, 3), 55)
    x1 = _inputs(kind, (b, c1, h, w_), (1, 1, 1, 1), 56)[0]
    gy = torch.randn(b, co, h, w_, generator=g)
    cat = torch.cat([TF.interpolate(x0, scale_factor=2, mode="nearest"), x1], 1)
    y64, dc64, dw64 = _reference(cat, wt, gy, 1, 1, torch.float64)
    y32, dc32, dw32 = _reference(cat, wt, gy, 1, 1, torch.float32)
    up_t = lambda d: d[:, :c0].reshape(b, c0, h // 2, 2, w_ // 2, 2).sum((3, 5))      # noqa: E731  transpose of nearest x2

    def run(expect):
        a0, a1, ww = nhwc(x0).requires_grad_(True), nhwc(x1).requires_grad_(True), wt.to(DEV).requires_grad_(True)
        y = F.conv2d(a0, ww, x1=a1, pad=1, up0=True)
        assert expect in _last(), _last()
        y.backward(nhwc(gy))
        return (l2(nchw(y.detach(), co), y64), l2(nchw(a1.grad, c1), dc64[:, c0:]), l2(nchw(a0.grad, c0), up_t(dc64)),
                l2(ww.grad.cpu(), dw64))

    got = run(kernel)
    with _arith(0):
        mfma = run("f32")
    torch32 = (l2(y32, y64), l2(dc32[:, c0:], dc64[:, c0:]), l2(up_t(dc32), up_t(dc64)), l2(dw32, dw64))
    for name, e, t, m in zip(("forward", "d skip", "d up-sampled input", "d weight"), got, torch32, mfma):
        assert e <= _bound(t, m, 3), f"{kind} {name}: {e:.3e} vs torch fp32 {t:.3e}, f32 MFMA {m:.3e}"


@pytest.mark.parametrize("kind", ["normal", "wide", "cancel"])
@pytest.mark.parametrize("in_ch", [3, 6])
def test_split_stem_forward_and_weight_gradient(in_ch, kind):
    """ResNet conv1 (7x7 / 2 / 3) in its double-pixel form: the halo-tile forward (`conv_stem74f_kernel`, whole 8 x 32 tiles) and the
    weight gradient `wgrad_stem74s_kernel` (whole 4 x 16 tiles: its double-buffered form)."""
    F = _F()
    b, h, w_ = 4, 128, 256
    x, wt, g = _inputs(kind, (b, in_ch, h, w_), (64, in_ch, 7, 7), 91 + in_ch)
    gy = torch.randn(b, 64, h // 2, w_ // 2, generator=g)
    y64, _, dw64 = _reference(x, wt, gy, 2, 3, torch.float64)
    y32, _, dw32 = _reference(x, wt, gy, 2, 3, torch.float32)
    frames = [x[:, i:i + 3].contiguous().to(DEV) for i in range(0, in_ch, 3)]

    def run(k_fwd, k_wg):
        xin = F.images_to_nhwc(frames, 8, F.STEM_XPAD)
        ww = wt.to(DEV).requires_grad_(True)
        y = F.stem_conv2d(xin, ww)
        assert _last() == k_fwd, _last()
        y.backward(nhwc(gy))
        assert _last() == k_wg, _last()
        return l2(nchw(y.detach(), 64), y64), l2(ww.grad.cpu(), dw64)

    got = run("conv_stem74f_kernel", "wgrad_stem74s_kernel")
    with _arith(0):
        mfma = run("conv_igemm_kernel<f32>", "wgrad_small_kernel")
    for name, e, t, m in zip(("forward", "d weight"), got, (l2(y32, y64), l2(dw32, dw64)), mfma):
        assert e <= _bound(t, m, 3), f"{kind} {name}: {e:.3e} vs torch fp32 {t:.3e}, f32 MFMA {m:.3e}"


@pytest.mark.parametrize("kind", ["normal", "wide"])
def test_co32_weight_gradient_with_a_chunk_that_straddles_the_concat(kind):
    """The decoders' half-resolution stage conv3x3(cat(nearest_x2(x0: 32 ch), x1: 64 ch)) -> 32 channels (models/depth_head.py:60-71):
    its weight gradient on the split kernel's CO32 form; the first 64-channel chunk holds 32 up-sampled + 32 skip channels."""
    F = _F()
    b, h, w_, c0, c1, co = 3, 64, 128, 32, 64, 32
    x0, wt, g = _inputs(kind, (b, c0, h // 2, w_ // 2), (co, c0 + c1, 3, 3), 77)
    x1 = _inputs(kind, (b, c1, h, w_), (1, 1, 1, 1), 78)[0]
    gy = torch.randn(b, co, h, w_, generator=g)
    cat = torch.cat([TF.interpolate(x0, scale_factor=2, mode="nearest"), x1], 1)
    _y64, _dc64, dw64 = _reference(cat, wt, gy, 1, 1, torch.float64)
    _y32, _dc32, dw32 = _reference(cat, wt, gy, 1, 1, torch.float32)

    def run(expect):
        ww = wt.to(DEV).requires_grad_(True)
        y = F.conv2d(nhwc(x0), ww, x1=nhwc(x1), pad=1, up0=True)
        y.backward(nhwc(gy))
        assert expect in _last(), _last()
        return l2(ww.grad.cpu(), dw64)

    e = run("wgrad_halo64s_kernel<co32>")
    with _arith(0):
        m = run("wgrad")
    t = l2(dw32, dw64)
    assert e <= _bound(t, m, 3), f"{kind}: {e:.3e} vs torch fp32 {t:.3e}, f32 MFMA {m:.3e}"
