"""Randomised shapes through the loss kernels and the panoptic fusion against the oracle (CPU, float64 where the reference is
float): odd image sizes, single rows, every top-k fraction, class weights, void pixels; seeded."""
import random

import numpy as np
import pytest
import torch

import inputs as gi
from helpers import close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
THINGS = [9, 10, 11, 12, 13]


def nhwc(x, cp=16):
    b, c, h, w = x.shape
    y = torch.zeros(b, h, w, cp)
    y[..., :c] = x.permute(0, 2, 3, 1)
    return y.to(DEV)


def _shapes(n, seed, lo=2):
    rng = random.Random(seed)
    return [(rng.randint(1, 3), rng.randint(lo, 40), rng.randint(lo, 70), rng.randint(0, 10 ** 6)) for _ in range(n)]


@pytest.mark.parametrize("b,h,w,seed", _shapes(12, 11))
def test_random_semantic_and_instance_losses(b, h, w, seed):
    from codeps_amd import functional as F
    from oracle import losses
    rng = random.Random(seed)
    c = rng.choice([2, 5, 14, 16])
    topk = rng.choice([0.05, 0.2, 0.5, 1.0])
    cw = [rng.uniform(0.5, 2.0) for _ in range(c)] if rng.random() < 0.5 else None
    probs = torch.softmax(gi.normal((b, c, h, w), seed, 2.0), 1)
    target, pw = gi.semantic_labels(b, h, w, seed, c), gi.pixel_weights(b, h, w, seed)
    p = nhwc(probs).requires_grad_(True)
    loss = F.semantic_loss(p, c, target.to(DEV), pw.to(DEV), None if cw is None else torch.tensor(cw, device=DEV), topk)
    loss.backward()
    rp = probs.double().requires_grad_(True)
    ref = losses.semantic_loss(rp, target.long(), pw.double(), None if cw is None else torch.tensor(cw, dtype=torch.float64), topk)
    ref.backward()
    close(loss, ref.float(), 1e-6, 3e-5, "semantic loss")
    # top-k ties at the k-th loss value may select different (equal-loss) pixels: compare gradient mass and agreement
    g, rg = p.grad[..., :c].permute(0, 3, 1, 2).cpu().double(), rp.grad
    assert abs(float(g.sum() - rg.sum())) <= 1e-4 * max(1.0, float(rg.abs().sum()))
    assert float(((g - rg).abs() > 1e-5 + 1e-3 * rg.abs()).double().mean()) < 5e-3
    cp = nhwc(gi.normal((b, 1, h, w), seed + 1)).requires_grad_(True)
    op = nhwc(gi.normal((b, 2, h, w), seed + 2, 10.0)).requires_grad_(True)
    ct, ot = gi.center_heatmap(b, h, w, seed), gi.offsets(b, h, w, seed)
    lc, lo = F.center_loss(cp, ct.to(DEV)), F.offset_loss(op, ot.to(DEV))
    (lc + lo).backward()
    close(lc, losses.center_loss(cp.detach()[..., :1].permute(0, 3, 1, 2).cpu().double(), ct.double()).float(), 1e-6, 2e-5, "center")
    close(lo, losses.offset_loss(op.detach()[..., :2].permute(0, 3, 1, 2).cpu().double(), ot.double()).float(), 1e-6, 2e-5, "offset")


@pytest.mark.parametrize("b,h,w,seed", _shapes(10, 23, lo=3))
def test_random_smoothness(b, h, w, seed):
    from codeps_amd import functional as F
    from oracle import losses
    img = gi.triplet(b, h, w, seed)[0]
    disp = gi.uniform((b, 1, h, w), seed + 4, 0.01, 1.0)
    d = disp.to(DEV).requires_grad_(True)
    loss = F.smoothness_loss(img.to(DEV), d)
    loss.backward()
    rd = disp.double().requires_grad_(True)
    ref = losses.smoothness_loss(img.double(), rd)
    ref.backward()
    close(loss, ref.float(), 1e-7, 3e-5, "smoothness")
    close(d.grad, rd.grad.float(), 1e-8, 3e-3, "d disp")


@pytest.mark.parametrize("b,h,w,seed", _shapes(12, 37, lo=8))
def test_random_panoptic_fusion(b, h, w, seed):
    from codeps_amd import functional as F
    from oracle import panoptic
    rng = random.Random(seed)
    sem = gi.semantic_labels(b, h, w, seed, 14, block=rng.choice([2, 4, 8]))
    ctr = gi.center_heatmap(b, h, w, seed, rng.randint(1, 12), rng.choice([2.0, 4.0, 8.0]))
    off = gi.offsets(b, h, w, seed, rng.choice([4.0, 16.0]))
    thr = rng.choice([0.1, 0.3])
    pan, inst, centers, state = F.panoptic_fusion(sem.to(DEV), ctr.to(DEV), off.to(DEV), THINGS, 1000, 255, thr, 7, 200)
    rp, ri, pts = panoptic.panoptic_fusion(sem.numpy(), ctr.numpy(), off.numpy(), THINGS, thr)
    assert np.array_equal(pan.cpu().numpy(), rp)
    assert np.array_equal(inst.cpu().numpy(), ri)
    for i in range(b):
        assert np.array_equal(centers[i, :int(state[i, 1])].cpu().numpy(), pts[i])
