"""HIP loss kernels, pose matrix, panoptic fusion and Adam/EMA against the reference goldens / oracle."""
import numpy as np
import pytest
import torch

import inputs as gi
from helpers import close, load, t

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def nhwc(x, cp=16):
    b, c, h, w = x.shape
    y = torch.zeros(b, h, w, cp)
    y[..., :c] = x.permute(0, 2, 3, 1)
    return y.to(DEV)


def test_smoothness_golden():
    from codeps_amd import functional as F
    for tag in ("small", "odd"):
        z, meta = load("recon_" + tag)
        b, h, w, seed = meta["b"], meta["h"], meta["w"], meta["seed"]
        img = gi.triplet(b, h, w, seed)[0].to(DEV)
        disp = gi.uniform((b, 1, h, w), seed + 4, 0.01, 1.0).to(DEV).requires_grad_(True)
        loss = F.smoothness_loss(img, disp)
        (loss * 2.0).backward()
        close(loss, z["smooth"], 1e-7, 2e-5, "smoothness")
        close(disp.grad / 2.0, z["ddisp"], 1e-8, 2e-3, "d disp")


def test_semantic_and_instance_losses_golden():
    from codeps_amd import functional as F
    z, meta = load("sem_inst_losses")
    b, h, w, c, seed = meta["b"], meta["h"], meta["w"], meta["c"], meta["seed"]
    probs = torch.softmax(gi.normal((b, c, h, w), seed, 2.0), 1)
    target, pw = gi.semantic_labels(b, h, w, seed, c).to(DEV), gi.pixel_weights(b, h, w, seed).to(DEV)
    for name, topk, cw in (("topk", 0.2, [1.0] * c), ("full", 1.0, None), ("cw", 0.2, z["class_weights"].tolist())):
        p = nhwc(probs).requires_grad_(True)
        cwt = None if cw is None else torch.tensor(cw, device=DEV)
        loss = F.semantic_loss(p, c, target, pw, cwt, topk)
        loss.backward()
        close(loss, z["sem_" + name], 1e-6, 2e-5, name)
        close(p.grad[..., :c].permute(0, 3, 1, 2), z[f"sem_{name}_dprobs"], 1e-9, 1e-3, name + " grad")
        assert p.grad[..., c:].abs().max().item() == 0
    cp = nhwc(gi.normal((b, 1, h, w), seed + 1)).requires_grad_(True)
    op = nhwc(gi.normal((b, 2, h, w), seed + 2, 10.0)).requires_grad_(True)
    tp = nhwc(torch.softmax(gi.normal((b, 2, h, w), seed + 3), 1)).requires_grad_(True)
    lc = F.center_loss(cp, gi.center_heatmap(b, h, w, seed).to(DEV))
    lo = F.offset_loss(op, gi.offsets(b, h, w, seed).to(DEV))
    lm = F.binary_mask_loss(tp, gi.semantic_labels(b, h, w, seed + 4, 2).to(DEV))
    (lc + lo + lm).backward()
    for a, k in ((lc, "center"), (lo, "offset"), (lm, "mask")):
        close(a, z[k], 1e-6, 2e-5, k)
    close(cp.grad[..., :1].permute(0, 3, 1, 2), z["dcenter"], 1e-10, 1e-4)
    close(op.grad[..., :2].permute(0, 3, 1, 2), z["doffset"], 1e-10, 1e-4)
    close(tp.grad[..., :2].permute(0, 3, 1, 2), z["dmask"], 1e-10, 1e-3)


def test_semantic_loss_ignore_labels_golden():
    """SemanticLoss(ignore_labels=[...]) (algos/semantic_seg.py:34-40,50-55): in-place remap of the targets, CE over the
    preserved channels indexed with the original label values, filtered class weights."""
    from codeps_amd.algos import SemanticLoss
    z, meta = load("sem_inst_losses")
    b, h, w, c, seed = meta["b"], meta["h"], meta["w"], meta["c"], meta["seed"]
    ign = z["ignore_labels"].tolist()
    pw = gi.pixel_weights(b, h, w, seed).to(DEV)
    for name, topk in (("ign_topk", 0.2), ("ign_full", 1.0)):
        probs = torch.softmax(gi.normal((b, c, h, w), seed, 2.0), 1).to(DEV).requires_grad_(True)
        tgt = gi.semantic_labels(b, h, w, seed + 9, 11).to(DEV)
        sl = SemanticLoss(device=DEV, class_weights=z["class_weights"].tolist(), top_k_percent_pixels=topk, ignore_labels=ign)
        loss = sl(probs, tgt, pw)
        loss.backward()
        close(loss, z["sem_" + name], 1e-6, 2e-5, name)
        close(probs.grad, z[f"sem_{name}_dprobs"], 1e-9, 1e-3, name + " grad")
        assert np.array_equal(tgt.cpu().numpy(), z[f"sem_{name}_target_after"]), "targets are remapped in place like the reference"
        assert float(probs.grad[:, ign].abs().max()) == 0.0, "dropped channels receive no gradient"


def test_topk_properties_full_size():
    """512x1024 batch: value equals sort-based top-k mean; heavy ties (80 % zero-loss pixels) handled."""
    from codeps_amd import functional as F
    b, h, w, c = 2, 512, 1024, 14
    g = torch.Generator().manual_seed(0)
    probs = torch.softmax(torch.randn(b, h, w, c, generator=g) * 2, -1)
    p = torch.zeros(b, h, w, 16)
    p[..., :c] = probs
    target = torch.randint(0, c, (b, h, w), generator=g).to(torch.uint8)
    target[torch.rand(b, h, w, generator=g) < 0.85] = 255          # fewer valid pixels than k -> threshold is 0
    pd = p.to(DEV).requires_grad_(True)
    loss = F.semantic_loss(pd, c, target.to(DEV), None, None, 0.2)
    loss.backward()
    ref = torch.nn.functional.cross_entropy(probs.permute(0, 3, 1, 2), target.long(), ignore_index=255, reduction="none")
    k = int(0.2 * ref.numel())
    want = torch.topk(ref.flatten(), k)[0].mean()
    close(loss, want, 1e-7, 2e-5)
    assert torch.isfinite(pd.grad).all() and (pd.grad[target.to(DEV) == 255] == 0).all()


def test_pose_matrix_against_oracle():
    from codeps_amd import functional as F
    from oracle import heads
    g = torch.Generator().manual_seed(4)
    feat = torch.randn(3, 12, 5, 7, generator=g) * 30
    for inv in (False, True):
        rf = feat.double().requires_grad_(True)
        v = 0.01 * rf.mean(3).mean(2).view(-1, 2, 6)
        T = heads.transformation_from_parameters(v[:, 0, :3], v[:, 0, 3:], inv)
        gT = torch.randn(T.shape, generator=g).double()
        (T * gT).sum().backward()
        df = nhwc(feat).requires_grad_(True)
        Td = F.pose_matrix(df, inv)
        close(Td, T.detach().float(), 1e-6, 1e-5, "T")
        (Td * gT.float().to(DEV)).sum().backward()
        close(df.grad[..., :12].permute(0, 3, 1, 2), rf.grad.float(), 1e-9, 2e-3, "d feat")


def test_panoptic_fusion_bit_exact_golden():
    from codeps_amd import functional as F
    z, meta = load("panoptic")
    for name in meta["cases"]:
        sem, ctr, off = t(z[name + "_sem"]).to(DEV), t(z[name + "_ctr"]).to(DEV), t(z[name + "_off"]).to(DEV)
        pan, inst, centers, state = F.panoptic_fusion(sem, ctr, off, meta["thing_list"])
        assert pan.dtype == torch.int16 and inst.dtype == sem.dtype
        assert np.array_equal(pan.cpu().numpy(), z[name + "_pan"]), f"{name}: panoptic ids differ"
        assert np.array_equal(inst.cpu().numpy(), z[name + "_inst"]), f"{name}: instance ids differ"
        for i in range(sem.shape[0]):
            k = int(state[i, 1])
            assert np.array_equal(centers[i, :k].cpu().numpy(), z[f"{name}_pts{i}"]), f"{name}: centres differ"


def test_panoptic_fusion_against_oracle_full_size():
    from codeps_amd import functional as F
    from oracle import panoptic
    b, h, w = 1, 512, 1024
    sem = gi.semantic_labels(b, h, w, 3, 14, block=32)
    ctr = gi.center_heatmap(b, h, w, 3, 40, 8.0)
    off = gi.offsets(b, h, w, 3, 64.0)
    pan, inst, centers, state = F.panoptic_fusion(sem.to(DEV), ctr.to(DEV), off.to(DEV), [9, 10, 11, 12, 13])
    rp, ri, pts = panoptic.panoptic_fusion(sem.numpy(), ctr.numpy(), off.numpy(), [9, 10, 11, 12, 13])
    assert np.array_equal(centers[0, :int(state[0, 1])].cpu().numpy(), pts[0])
    assert np.array_equal(inst.cpu().numpy(), ri) and np.array_equal(pan.cpu().numpy(), rp)
    # idempotence of the NMS: fusing again gives the identical result (no atomically-ordered state)
    pan2, inst2, _, _ = F.panoptic_fusion(sem.to(DEV), ctr.to(DEV), off.to(DEV), [9, 10, 11, 12, 13])
    assert torch.equal(pan, pan2) and torch.equal(inst, inst2)


def test_adam_ema_step():
    from codeps_amd import functional as F
    from oracle import network
    g = torch.Generator().manual_seed(2)
    n = 100003
    p, m, v = torch.randn(n, generator=g), torch.zeros(n), torch.zeros(n)
    ema = p.clone()
    dp, dm, dv, de = (x.clone().to(DEV) for x in (p, m, v, ema))
    for step in range(1, 4):
        grad = torch.randn(n, generator=g)
        network.adam_step(p, grad, m, v, step)
        network.ema_update(ema, p, 0.99)
        F.adam_ema_step(dp, grad.to(DEV), dm, dv, step, ema_params=de, ema_alpha=0.99)
    close(dp, p, 1e-6, 1e-5, "param")
    close(de, ema, 1e-6, 1e-5, "ema")
    # reference semantics: identical to torch.optim.Adam
    q = torch.nn.Parameter(torch.ones(5))
    opt = torch.optim.Adam([q], lr=1e-4)
    q.grad = torch.arange(5.0)
    opt.step()
    dq, dmq, dvq = torch.ones(5, device=DEV), torch.zeros(5, device=DEV), torch.zeros(5, device=DEV)
    F.adam_ema_step(dq, torch.arange(5.0).to(DEV), dmq, dvq, 1)
    close(dq, q.detach(), 1e-7, 1e-6)


def test_degenerate_inputs_match_oracle():
    """Edge cases of the domain: no thing pixels, no centre above the threshold, every thing pixel but a single centre,
    an all-ignored semantic target, batch 1 at the smallest legal size."""
    from codeps_amd import functional as F
    from oracle import losses, panoptic
    things = [9, 10, 11, 12, 13]
    b, h, w = 2, 24, 40
    sem = gi.semantic_labels(b, h, w, 5, 14, block=8)
    ctr = gi.center_heatmap(b, h, w, 5, 3, 4.0)
    off = gi.offsets(b, h, w, 5, 16.0)
    cases = {
        "no_things": (torch.clamp(sem, max=8), ctr, off),
        "no_centres": (sem, ctr * 0.0, off),
        "all_things_one_centre": (torch.full_like(sem, 11), ctr, off * 0.0),
    }
    for name, (s_, c_, o_) in cases.items():
        pan, inst, centers, state = F.panoptic_fusion(s_.to(DEV), c_.to(DEV), o_.to(DEV), things)
        rp, ri, pts = panoptic.panoptic_fusion(s_.numpy(), c_.numpy(), o_.numpy(), things)
        assert np.array_equal(pan.cpu().numpy(), rp), name
        assert np.array_equal(inst.cpu().numpy(), ri), name
        for i in range(b):
            assert np.array_equal(centers[i, :int(state[i, 1])].cpu().numpy(), pts[i]), name
    # semantic loss: every pixel ignored -> the reference's top-k mean over zero losses is 0 with zero gradient
    c = 14
    probs = torch.softmax(gi.normal((1, c, 8, 16), 3, 2.0), 1)
    target = torch.full((1, 8, 16), 255, dtype=torch.uint8)
    p = nhwc(probs).requires_grad_(True)
    loss = F.semantic_loss(p, c, target.to(DEV), None, None, 0.2)
    loss.backward()
    ref = losses.semantic_loss(probs.double(), target.long(), torch.ones(1, 8, 16, dtype=torch.float64), None, 0.2)
    close(loss, ref.float(), 1e-7, 1e-6, "all-ignored semantic loss")
    assert float(p.grad.abs().max()) == 0.0
