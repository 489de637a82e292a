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
