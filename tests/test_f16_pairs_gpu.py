"""Range behaviour of the two-fp16-piece arithmetic (csrc/conv.hip, split_f16; the default since round 6).  fp16 has five exponent bits, so
every operand travels as x 2^k with a power-of-two scale chosen per weight tensor (the pack) and per block / row tile / pixel for activations
and gradients (running maximum exponent, sums re-based exactly when it grows).  What that has to guarantee, per kernel family:
  * magnitude does not matter: inputs at 1e-30, 1e-6, 1, 1e4 and 1e30 scale give the same relative error against float64 (no overflow to
    inf, no underflow to zero), and scaling an operand by a power of two scales the result bit for bit;
  * a tile whose magnitudes are spread over many octaves (a bright pixel beside a dark region) keeps fp32-level accuracy where it matters:
    the error of every output is small against the scale of ITS tile's terms;
  * an all-zero tile gives exact zeros; fp32 subnormal inputs do not produce NaN;
  * a non-finite input shows up as non-finite output (it is not silently dropped).
Reference arithmetic: fp32 nn.Conv2d throughout models/resnet_encoder.py:115-126, models/depth_head.py:56-78."""
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

#        name            B  H   W    Ci   Co  k  expected kernel family of the forward
SHAPES = {"halo3": (4, 64, 128, 64, 64, 3, "conv_halo3_kernel"),
          "ring": (2, 64, 128, 128, 256, 3, "conv_halo3r_kernel"),
          "igemm_1x1": (2, 32, 64, 256, 512, 1, "conv_igemm_kernel<split3>"),
          "ws_1x1": (4, 32, 64, 2048, 512, 1, "conv_igemm_ws_kernel"),
          "n16": (2, 32, 64, 32, 16, 3, "conv_halo3n16_kernel")}


def _l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def _run(x, w, k):
    from codeps_amd import functional as F
    from codeps_amd._lib import lib
    from helpers import kernel_family
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    y = F.conv2d(xd, wd, pad=k // 2)
    raw = lib().codeps_conv2d_last_kernel().decode()
    g = torch.Generator().manual_seed(5)
    gy = torch.randn(y.shape, generator=g).to(DEV)
    y.backward(gy)
    co, ci = w.shape[:2]
    return (y.detach()[..., :co].permute(0, 3, 1, 2).cpu(), xd.grad[..., :ci].permute(0, 3, 1, 2).cpu(), wd.grad.cpu(),
            gy[..., :co].permute(0, 3, 1, 2).cpu(), kernel_family(raw), raw)


def _ref(x, w, gy, k):
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y = TF.conv2d(xr, wr, None, 1, k // 2)
    y.backward(gy.double())
    return y.detach(), xr.grad, wr.grad


@pytest.mark.parametrize("name", list(SHAPES))
@pytest.mark.parametrize("xs,ws", [(1e-30, 1.0), (1e-6, 1.0), (1.0, 1e-20), (1e4, 1.0), (1e30, 1e-5), (1e15, 1e15)])
def test_magnitude_does_not_matter(name, xs, ws):
    b, h, w_, ci, co, k, fam = SHAPES[name]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(b, ci, h, w_, generator=g)
    wt = torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5
    y1, dx1, dw1, gy, fam1, raw = _run(x, wt, k)
    assert fam1 == fam and "f16x2" in raw, raw                          # the shape selects the kernel under test, on the fp16 arithmetic
    y64, dx64, dw64 = _ref(x, wt, gy, k)
    base = (_l2(y1, y64), _l2(dx1, dx64), _l2(dw1, dw64))
    y2, dx2, dw2, _gy, _f, _r = _run(x * xs, wt * ws, k)
    assert torch.isfinite(y2).all() and torch.isfinite(dx2).all() and torch.isfinite(dw2).all()
    got = (_l2(y2, y64 * (xs * ws)), _l2(dx2, dx64 * ws), _l2(dw2, dw64 * xs))
    for nm, e, e0 in zip(("forward", "data gradient", "weight gradient"), got, base):
        assert e <= 1.5 * e0 + 1e-7, f"{name} {nm} at scales ({xs:g}, {ws:g}): {e:.2e} against {e0:.2e} at scale 1"


@pytest.mark.parametrize("name", list(SHAPES))
def test_power_of_two_scaling_is_exact(name):
    b, h, w_, ci, co, k, _fam = SHAPES[name]
    g = torch.Generator().manual_seed(4)
    x = torch.randn(b, ci, h, w_, generator=g)
    wt = torch.randn(co, ci, k, k, generator=g) * 0.05
    y1, dx1, dw1, _gy, _f, _r = _run(x, wt, k)
    y2, dx2, dw2, _gy2, _f2, _r2 = _run(x * 2.0 ** 9, wt * 2.0 ** -3, k)
    assert torch.equal(y2, y1 * 2.0 ** 6) and torch.equal(dx2, dx1 * 2.0 ** -3) and torch.equal(dw2, dw1 * 2.0 ** 9)


@pytest.mark.parametrize("name", ["halo3", "ring", "igemm_1x1"])
def test_wide_range_inside_one_tile(name):
    """One pixel of every tile 2^20 times larger than the rest: the small outputs (windows that do not touch it) lose their lo pieces to
    fp16's subnormal range gradually -- their error stays below 2^-12 of their own scale and 2^-30 of the tile's largest output, every
    output's below 2^-20 of the largest (what fp32 summation gives an output that cancels large terms)."""
    b, h, w_, ci, co, k, _fam = SHAPES[name]
    g = torch.Generator().manual_seed(6)
    x = torch.randn(b, ci, h, w_, generator=g)
    x[:, :, ::4, ::32] *= 2.0 ** 20
    wt = torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5
    y, _dx, _dw, gy, _f, _r = _run(x, wt, k)
    y64, _a, _b = _ref(x, wt, gy, k)
    err = (y.double() - y64).abs()
    big = float(y64.abs().max())
    assert float(err.max()) <= 2.0 ** -20 * big, (float(err.max()), big)               # fp32-level against the tile's scale
    # outputs whose windows miss the large pixels (rows 0 mod 4, columns 0 mod 32): their terms are 2^20 below the tile's largest
    yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w_), indexing="ij")
    r = k // 2
    near = torch.zeros(h, w_, dtype=torch.bool)
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            near |= (((yy + dy) % 4) == 0) & (((xx + dx) % 32) == 0)
    small = ~near
    e_small, v_small = float(err[:, :, small].max()), float(y64[:, :, small].abs().max())
    assert e_small <= 2.0 ** -12 * v_small and e_small <= 2.0 ** -30 * big, (e_small, v_small, big)


def test_zeros_subnormals_and_non_finite_inputs():
    b, h, w_, ci, co, k, _fam = SHAPES["ring"]
    g = torch.Generator().manual_seed(8)
    wt = torch.randn(co, ci, k, k, generator=g) * 0.05
    y, dx, dw, _gy, _f, _r = _run(torch.zeros(b, ci, h, w_), wt, k)
    assert float(y.abs().max()) == 0.0 and float(dw.abs().max()) == 0.0 and torch.isfinite(dx).all()
    y, dx, dw, _gy, _f, _r = _run(torch.full((b, ci, h, w_), 1e-42), wt, k)                 # fp32 subnormals: finite (they count as zero)
    assert torch.isfinite(y).all() and torch.isfinite(dx).all() and torch.isfinite(dw).all()
    x = torch.randn(b, ci, h, w_, generator=g)
    x[0, 3, 5, 7] = float("inf")
    y, _dx, _dw, _gy, _f, _r = _run(x, wt, k)
    assert not torch.isfinite(y[0, :, 4:7, 6:9]).all(), "an infinite input left no trace in the outputs whose windows contain it"
    x[0, 3, 5, 7] = float("nan")
    y, _dx, _dw, _gy, _f, _r = _run(x, wt, k)
    assert torch.isnan(y[0, :, 5, 7]).any()
