"""Randomised shapes through the non-conv layer kernels (train-mode BatchNorm with residual / ReLU, 3x3/2 max pooling, bilinear
up-sampling) against float64 on the CPU: single pixels, odd sizes, channel counts from one 16-channel group to 2048.  Seeded."""
import random

import pytest

from test_ops_gpu import run_bilinear_case, run_bn_case, run_maxpool_case

pytestmark = pytest.mark.gpu


def _bn_cases(n, seed):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        c = rng.choice([16, 32, 48, 64, 128, 256, 512, 1024, 2048])
        b, h, w = rng.randint(1, 5), rng.randint(1, 40), rng.randint(1, 50)
        if b * h * w < 4 or b * h * w * c > 4e6:
            b, h, w = 2, 6, 7
        out.append((c, rng.choice([None, "relu"]), rng.random() < 0.5, b, h, w))
    return out


@pytest.mark.parametrize("case", _bn_cases(24, 7), ids=lambda c: "-".join(str(v) for v in c))
def test_random_batch_norm_shapes(case):
    c, act, res, b, h, w = case
    run_bn_case(c, act, res, b, h, w, 100 + c + h * w)


@pytest.mark.parametrize("i", range(16))
def test_random_maxpool_shapes(i):
    rng = random.Random(900 + i)
    run_maxpool_case(rng.randint(1, 4), rng.choice([16, 64, 128, 256]), rng.randint(1, 40), rng.randint(1, 60), i)


@pytest.mark.parametrize("i", range(16))
def test_random_bilinear_shapes(i):
    rng = random.Random(1900 + i)
    h, w = rng.randint(1, 24), rng.randint(1, 30)
    oh, ow = h * rng.randint(1, 4) + rng.randint(0, 3), w * rng.randint(1, 4) + rng.randint(0, 3)
    run_bilinear_case(rng.random() < 0.5, rng.randint(1, 3), rng.choice([16, 32, 64, 144]), h, w, oh, ow, i)


@pytest.mark.parametrize("i", range(14))
def test_random_disparity_head_shapes(i):
    """The fused output stage of the depth decoder (reflection-padded 16 -> 1 convolution, sigmoid, disp_to_depth; one pass each
    way) against float64: outputs, data / weight / bias gradients, image sizes down to 2 x 2 (every pixel on the border)."""
    import torch
    import torch.nn.functional as TF
    from codeps_amd import functional as F
    rng = random.Random(2900 + i)
    b = rng.randint(1, 3)
    h, w = [(2, 2), (2, 5), (3, 3), (8, 32), (9, 33), (17, 70)][i] if i < 6 else (rng.randint(2, 40), rng.randint(2, 80))
    g = torch.Generator().manual_seed(500 + i)
    x = torch.randn(b, 16, h, w, generator=g)
    wt = torch.randn(1, 16, 3, 3, generator=g) * 0.2
    bs = torch.randn(1, generator=g) * 0.1
    gd, gs = torch.randn(b, 1, h, w, generator=g), torch.randn(b, 1, h, w, generator=g)
    rx, rw, rb = x.double().requires_grad_(True), wt.double().requires_grad_(True), bs.double().requires_grad_(True)
    disp = torch.sigmoid(TF.conv2d(TF.pad(rx, (1, 1, 1, 1), mode="reflect"), rw, rb))
    depth = 1 / (1 / 100 + (1 / 0.1 - 1 / 100) * disp)
    ((depth * gd.double()).sum() + (disp * gs.double()).sum()).backward()
    dx = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
    dw, db = wt.cuda().requires_grad_(True), bs.cuda().requires_grad_(True)
    d_depth, d_disp = F.disp_head(dx, dw, db)
    ((d_depth * gd.cuda()).sum() + (d_disp * gs.cuda()).sum()).backward()

    def rel(a, ref):
        return float((a.double().cpu() - ref).abs().max() / (ref.abs().max() + 1e-30))
    assert rel(d_disp, disp.detach()) < 2e-6 and rel(d_depth, depth.detach()) < 2e-5
    assert rel(dx.grad.permute(0, 3, 1, 2), rx.grad) < 2e-5
    assert rel(dw.grad, rw.grad) < 2e-5 and rel(db.grad, rb.grad) < 2e-5


@pytest.mark.parametrize("i", range(16))
def test_random_upconv16_shapes(i):
    """The decoders' last up-convolution as four 2 x 2 convolutions of the half-resolution tensor (pre-summed taps, f32 MFMA)
    against float64 nearest-x2 + 3 x 3 convolution + ELU: outputs and data / weight / bias gradients; source sizes from one pixel
    to several tiles with ragged edges."""
    import torch
    import torch.nn.functional as TF
    from codeps_amd import functional as F
    rng = random.Random(3900 + i)
    b = rng.randint(1, 3)
    hs, ws = [(1, 1), (1, 7), (2, 2), (8, 16), (9, 17), (16, 32), (5, 40)][i] if i < 7 else (rng.randint(1, 30), rng.randint(1, 50))
    act = [None, "elu", "relu"][i % 3]
    g = torch.Generator().manual_seed(700 + i)
    x = torch.randn(b, 16, hs, ws, generator=g)
    wt = torch.randn(16, 16, 3, 3, generator=g) / 12.0
    bs = torch.randn(16, generator=g) * 0.1 if i % 4 else None
    gy = torch.randn(b, 16, 2 * hs, 2 * ws, generator=g)
    rx, rw = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    rb = bs.double().requires_grad_(True) if bs is not None else None
    pre = TF.conv2d(TF.interpolate(rx, scale_factor=2, mode="nearest"), rw, rb, padding=1)
    y = TF.elu(pre) if act == "elu" else (TF.relu(pre) if act == "relu" else pre)
    gy64 = gy.double() * ((pre.detach().abs() > 1e-5) if act == "relu" else 1.0)      # no gradient on the ReLU kink
    (y * gy64).sum().backward()
    dx = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
    dw = wt.cuda().requires_grad_(True)
    db = bs.cuda().requires_grad_(True) if bs is not None else None
    out = F.upconv16(dx, dw, db, act)
    assert out.shape == (b, 2 * hs, 2 * ws, 16)
    out.backward(gy64.float().permute(0, 2, 3, 1).contiguous().cuda())

    def rel(a, ref):
        return float((a.detach().double().cpu() - ref).abs().max() / (ref.abs().max() + 1e-30))
    assert rel(out.permute(0, 3, 1, 2), y.detach()) < 5e-6
    assert rel(dx.grad.permute(0, 3, 1, 2), rx.grad) < 2e-5
    assert rel(dw.grad, rw.grad) < 2e-5
    if bs is not None:
        assert rel(db.grad, rb.grad) < 2e-5
    # and against the generic engine's fused-gather form
    ref = F.conv2d(dx.detach(), dw.detach(), None if db is None else db.detach(), pad=1, act=act, up0=True)
    assert rel(out, ref.detach().double().cpu()) < 5e-6
