"""The composed network at the HEADLINE size against the oracle (missing item 3): one 512 x 1024 image through the frozen
ResNet-101 encoder (eval-mode BatchNorm folded into the conv epilogues, the layer3 chain kernel, the kernel variants the cost model
picks at this size), DepthHead and SemanticHead (GUDA fusion), compared with `oracle.network` / `oracle.heads` on the same weights in
fp32 -- north_star's bar: depth / semantic LOGITS within 1e-3 -- and in float64 (how much of the difference is the fp32 reference's own
rounding).  models/resnet_encoder.py:115-126, models/depth_head.py:56-78, models/semantic_head.py:46-76."""
import numpy as np
import pytest
import torch

import inputs as gi
from helpers import record_bar

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


_TRUNK = {}


def _trunk(temper):
    """ResNet-101 with random weights at 512 x 1024 on the GPU, its state dict as the oracle receives it, and the image triplet.
    temper: factor on every bn3.weight (0.25 = residual branches at a quarter of the identity's scale, as in a trained zero-init-residual
    trunk: 33 random full-scale bottlenecks make a chaotic map whose fp32 evaluations differ by 1e-3 from float64; 1.0 = as drawn)."""
    if temper in _TRUNK:
        return _TRUNK[temper]
    from codeps_amd.models import ResnetEncoder
    from oracle import resnet as oresnet
    layers, h, w = 101, 512, 1024
    sd = gi.fill_state_dict(gi.shapes_of(oresnet.make_state_dict(layers, 3, 0)), 77 + layers)
    for k in sd:
        if k.endswith("bn3.weight"):
            sd[k] = sd[k] * temper
    enc = ResnetEncoder(layers, False, 1, 3)
    enc.load_state_dict({"encoder." + k: v for k, v in sd.items()})
    enc.to(DEV)
    imgs = gi.triplet(1, h, w, 5)
    x = imgs[0]
    # 101 layers of random weights with unit running statistics blow the activations up to 1e8: calibrate the running statistics
    # with ONE train-mode pass (momentum 1: running = batch statistics of this image), as a trained checkpoint's would be -- the
    # weights under test are whatever that leaves in the state dict, which the oracle then receives
    for m in enc.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 1.0
    enc.train()
    with torch.no_grad():
        enc(x.to(DEV))
    enc.eval()
    sd = {k[len("encoder."):]: v.detach().cpu().clone() for k, v in enc.state_dict().items()}
    _TRUNK[temper] = (enc, sd, x, imgs)
    return _TRUNK[temper]


_OFEATS = {}


def _oracle_features(temper, dt):
    if (temper, dt) not in _OFEATS:
        from oracle import resnet as oresnet
        _enc, sd, x, _ = _trunk(temper)
        cast = lambda t: t.to(dt) if t.is_floating_point() else t      # noqa: E731
        with torch.no_grad():
            _OFEATS[(temper, dt)] = oresnet.features({k: cast(v) for k, v in sd.items()}, x.to(dt), 101, training=False)
    return _OFEATS[(temper, dt)]


def _heads(far):
    """DepthHead / SemanticHead with drawn weights; far: the full-resolution disparity bias lowered by 9 so that depth = 1 / (0.01 + 9.99
    sigmoid(logit)) reaches the far end of its range (>= 80 m), where d depth / d logit ~ depth."""
    from codeps_amd.models import DepthHead, SemanticHead
    enc = _trunk(0.25)[0]
    ch, ncls = enc.num_ch_enc, 14
    depth, sem = DepthHead(ch, True), SemanticHead(ch, ncls, True, True)
    dsd = gi.fill_state_dict(gi.shapes_of(depth.state_dict()), 401)
    ssd = gi.fill_state_dict(gi.shapes_of(sem.state_dict()), 402)
    if far:
        dsd["dispconvs.0.1.bias"] = dsd["dispconvs.0.1.bias"] - 9.0
    depth.load_state_dict(dsd)
    sem.load_state_dict(ssd)
    return depth.to(DEV), sem.to(DEV), dsd, ssd, ncls


def _run_threads():
    torch.set_num_threads(min(32, torch.get_num_threads() if torch.get_num_threads() > 8 else 32))


@pytest.mark.parametrize("far", [False, True], ids=["near", "far_range"])
def test_r101_depth_and_semantic_logits_at_512x1024_against_oracle(far):
    from oracle import heads as oheads
    _run_threads()
    enc, sd, x, _imgs = _trunk(0.25)
    depth, sem, dsd, ssd, ncls = _heads(far)
    depth.eval()
    sem.eval()
    depth.logit_log, sem.logit_log = [], []
    with torch.no_grad():
        feats = enc(x.to(DEV))
        d_hip = depth(feats)
        p_hip = sem(feats)
    torch.cuda.synchronize()
    dl_hip, sl_hip = depth.logit_log[0].cpu(), sem.logit_log[0].cpu()

    def oracle(dt):
        cast = lambda t: t.to(dt) if t.is_floating_point() else t      # noqa: E731
        with torch.no_grad():
            of = _oracle_features(0.25, dt)
            dl, sl = [], []
            od, _ = oheads.depth_head({k: cast(v) for k, v in dsd.items()}, of, logits=dl)
            op = oheads.semantic_head({k: cast(v) for k, v in ssd.items()}, of, True, logits_out=sl)
        return of, dl[0], sl[0], od, op

    of32, dl32, sl32, od32, op32 = oracle(torch.float32)
    of64, dl64, sl64, _od64, op64 = oracle(torch.float64)
    rec, bad = {}, []
    # encoder features first (a logit difference is only meaningful on equal features): against float64, with the fp32 oracle's own
    # distance from float64 as the yardstick of how ill-conditioned 101 random layers are at this size
    for i, (a, b32, b64) in enumerate(zip(feats, of32, of64)):
        rng = float(b64.abs().max())
        e_hip, e_o32 = float((a.cpu().double() - b64).abs().max()) / rng, float((b32.double() - b64).abs().max()) / rng
        rec[f"feat{i}_err_vs_fp64_rel"], rec[f"feat{i}_oracle_fp32_err_vs_fp64_rel"] = e_hip, e_o32
        if e_hip > max(2e-4, 3 * e_o32):
            bad.append(f"encoder feature {i}: {e_hip:.2e} of its range from float64 (fp32 oracle: {e_o32:.2e})")
    e_d = float((dl_hip - dl32).abs().max())
    e_s = float((sl_hip - sl32).abs().max())
    rec.update(depth_logit_max_abs_err=e_d, depth_logit_max_abs=float(dl32.abs().max()), semantic_logit_max_abs_err=e_s,
               semantic_logit_max_abs=float(sl32.abs().max()), logit_bar="1e-3 absolute vs the fp32 oracle (north_star)")
    rec.update(depth_logit_err_vs_fp64=float((dl_hip.double() - dl64).abs().max()),
               oracle_fp32_depth_logit_err_vs_fp64=float((dl32.double() - dl64).abs().max()),
               semantic_logit_err_vs_fp64=float((sl_hip.double() - sl64).abs().max()),
               oracle_fp32_semantic_logit_err_vs_fp64=float((sl32.double() - sl64).abs().max()))
    arg_hip, arg32 = p_hip.argmax(1).cpu(), op32.argmax(1)
    rec["semantic_argmax_differing_pixels"] = int((arg_hip != arg32).sum())
    rec["depth_max_abs_err_m"] = float((d_hip.cpu() - od32).abs().max())
    rec["depth_max_m"] = float(od32.max())
    record_bar("headline_512x1024_r101_depth_semantic" + ("_far_range" if far else ""), **rec)
    if far:
        assert rec["depth_max_m"] >= 80.0, f"the far-range variant only reaches {rec['depth_max_m']:.1f} m"
    assert not bad, "; ".join(bad)
    # north_star: logits within 1e-3 of the fp32 reference.  Where the fp32 reference is itself further than that from the exact
    # (float64) network, the claim is stated the way tests/helpers.depth_bar states it: this path's distance from float64 is at most
    # 4 x the fp32 oracle's own, and never beyond 1e-3 + 1e-4 |logit| of the fp32 oracle
    for name, e32, hip, o32, o64 in (("disparity", e_d, dl_hip, dl32, dl64), ("semantic", e_s, sl_hip, sl32, sl64)):
        if e32 <= 1e-3:
            continue
        e_hip, e_ref = float((hip.double() - o64).abs().max()), float((o32.double() - o64).abs().max())
        assert e_hip <= 4 * e_ref, f"{name} logits: {e_hip:.3e} from float64, the fp32 oracle {e_ref:.3e}"
        assert float(((hip - o32).abs() - 1e-4 * o32.abs()).max()) <= 1e-3, f"{name} logits differ by {e32:.3e} from the fp32 oracle"
    # arg-max: equal wherever the fp32 oracle's own top-2 margin exceeds twice the logit error
    top2 = sl32.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 2 * max(e_s, 1e-6)
    assert bool((arg_hip == arg32)[safe].all()), "semantic arg-max differs at a pixel whose margin exceeds the logit error"
    # metres: 1e-3 absolute, or the conditioning of disp_to_depth (d depth / d logit ~ depth at the far end)
    cap = float(((d_hip.cpu() - od32).abs() - 1e-4 * od32.abs()).max())
    assert cap <= 1e-3, f"depth differs by {rec['depth_max_abs_err_m']:.3e} m: beyond 1e-3 + 1e-4 |depth|"


def test_untempered_trunk_logit_error_is_recorded():
    """The same comparison with bn3.weight x 1 (the trunk as drawn: 33 full-scale random bottlenecks).  RECORDED, not asserted beyond
    finiteness: the map is chaotic at this depth, the fp32 oracle itself sits ~1e-3 of the feature range from float64 and nothing about the
    kernels can be read off a bar here -- the number belongs in profiles/rNN_parity_bars.json beside the tempered ones."""
    from oracle import heads as oheads
    _run_threads()
    enc, sd, x, _imgs = _trunk(1.0)
    depth, sem, dsd, ssd, _ncls = _heads(False)
    depth.eval()
    depth.logit_log = []
    with torch.no_grad():
        feats = enc(x.to(DEV))
        depth(feats)
    torch.cuda.synchronize()
    dl_hip = depth.logit_log[0].cpu()
    rec = {}
    dls = {}
    for dt, tag in ((torch.float32, "fp32"), (torch.float64, "fp64")):
        of = _oracle_features(1.0, dt)
        dl = []
        with torch.no_grad():
            oheads.depth_head({k: (v.to(dt) if v.is_floating_point() else v) for k, v in dsd.items()}, of, logits=dl)
        dls[tag] = dl[0]
        rng = float(of[-1].abs().max())
        rec[f"feat4_err_vs_oracle_{tag}_rel"] = float((feats[-1].cpu().double() - of[-1].double()).abs().max()) / rng
    rec["oracle_fp32_feat4_err_vs_fp64_rel"] = float((_oracle_features(1.0, torch.float32)[-1].double() - _oracle_features(1.0, torch.float64)[-1]).abs().max()) / \
        float(_oracle_features(1.0, torch.float64)[-1].abs().max())
    rec["depth_logit_max_abs_err_vs_oracle_fp32"] = float((dl_hip - dls["fp32"]).abs().max())
    rec["depth_logit_err_vs_fp64"] = float((dl_hip.double() - dls["fp64"]).abs().max())
    rec["oracle_fp32_depth_logit_err_vs_fp64"] = float((dls["fp32"].double() - dls["fp64"]).abs().max())
    record_bar("headline_512x1024_r101_untempered_recorded_only", **rec)
    assert all(np.isfinite(v) for v in rec.values())


def test_heads_and_losses_backward_at_512x1024_against_oracle():
    """The composed BACKWARD at the headline size (missing item 2): B = 1, 512 x 1024, frozen ResNet-101 features ->
    DepthHead + SemanticHead (train mode) -> photometric reconstruction + edge-aware smoothness + hard-pixel-mined semantic loss ->
    per-parameter gradients, against `oracle.heads` + `oracle.losses` on the SAME features (the HIP encoder's, so that the trunk's rounding
    does not enter) in fp32 and float64: relative L2 per module against float64 at most 3 x the fp32 oracle's own.  These are the
    data-gradient / weight-gradient kernel variants the cost model picks at this size (halo / ring / generic / 16-channel tail, GUDA gather
    backward), composed.  models/depth_head.py:56-78, algos/depth.py:239-326, algos/semantic_seg.py:45-67."""
    from codeps_amd import functional as F
    from oracle import heads as oheads, losses as olosses
    _run_threads()
    enc, _sd, x, imgs = _trunk(0.25)
    depth, sem, dsd, ssd, ncls = _heads(False)
    depth.train()
    sem.train()
    h, w = 512, 1024
    intr = gi.intrinsics(1, h, w, 5)
    poses = [gi.pose(1, 6), gi.pose(1, 7)]
    g = torch.Generator().manual_seed(9)
    noise = [torch.randn(1, 2, h >> s, w >> s, generator=g) * 1e-5 for s in range(5)]
    target = gi.semantic_labels(1, h, w, 11, ncls)                 # (blocks of 8 x 8 pixels, 5 % void = 255)
    pw = gi.pixel_weights(1, h, w, 12)
    w_photo, w_smth, w_sem = 1.0, 1e-3, 1.0
    with torch.no_grad():
        feats = [f.detach() for f in enc(x.to(DEV))]
    d, disp = depth(feats, return_disparity=True)
    probs = sem.forward_nhwc(feats)
    dev_imgs = [i.to(DEV) for i in imgs]
    l_photo = F.photometric_loss(dev_imgs, d, [p.to(DEV) for p in poses], intr.to(DEV), 5, 0.85, torch.cat([n.reshape(-1) for n in noise]).to(DEV))
    l_smth = F.smoothness_loss(dev_imgs[0], disp)
    l_sem = F.semantic_loss(probs, ncls, target.to(DEV), pw.to(DEV), None, 0.2)
    (w_photo * l_photo + w_smth * l_smth + w_sem * l_sem).backward()
    torch.cuda.synchronize()
    hip_losses = {"photo": float(l_photo), "smth": float(l_smth), "sem": float(l_sem)}
    feats_cpu = [f.cpu() for f in feats]

    def oracle(dt):
        osd = {"d": {k: v.clone().to(dt).requires_grad_(True) for k, v in dsd.items()}, "s": {k: v.clone().to(dt).requires_grad_(True) for k, v in ssd.items()}}
        of = [f.to(dt) for f in feats_cpu]
        od, odisp = oheads.depth_head(osd["d"], of)
        op = oheads.semantic_head(osd["s"], of, True)
        lp = olosses.reconstruction_loss(intr.to(dt), [i.to(dt) for i in imgs], od, [p.to(dt) for p in poses], 5, [n.to(dt) for n in noise])
        ls = olosses.smoothness_loss(imgs[0].to(dt), odisp)
        lm = olosses.semantic_loss(op, target.long(), pw.to(dt), None, 0.2)
        (w_photo * lp + w_smth * ls + w_sem * lm).backward()
        return {"photo": float(lp), "smth": float(ls), "sem": float(lm)}, osd

    o32, g32 = oracle(torch.float32)
    _o64, g64 = oracle(torch.float64)
    rec = {}
    for k in hip_losses:
        rec[f"loss_{k}_rel_err_vs_oracle_fp32"] = abs(hip_losses[k] - o32[k]) / max(abs(o32[k]), 1e-30)
        assert rec[f"loss_{k}_rel_err_vs_oracle_fp32"] <= 1e-4, f"loss {k}: hip {hip_losses[k]!r} vs oracle {o32[k]!r}"
    bad = []
    for tag, mod in (("d", depth), ("s", sem)):
        num_h = num_o = den = 0.0
        worst = (0.0, "")
        for k, p in mod.named_parameters():
            if k not in g64[tag] or g64[tag][k].grad is None:
                continue
            assert p.grad is not None, f"{tag}.{k}: the oracle has a gradient, the HIP path has none"
            r64 = g64[tag][k].grad
            eh, eo, dn = float((p.grad.detach().cpu().double() - r64).norm()), float((g32[tag][k].grad.double() - r64).norm()), float(r64.norm())
            num_h += eh ** 2
            num_o += eo ** 2
            den += dn ** 2
            if dn > 0 and eh / dn > worst[0]:
                worst = (eh / dn, k)
        name = "depth_head" if tag == "d" else "semantic_head"
        e_hip, e_o32 = (num_h / den) ** 0.5, (num_o / den) ** 0.5
        rec[f"{name}_grad_rel_l2_vs_fp64"], rec[f"{name}_oracle_fp32_grad_rel_l2_vs_fp64"] = e_hip, e_o32
        rec[f"{name}_worst_parameter_rel_l2_vs_fp64"] = worst[0]
        print(f"[headline backward] {name}: gradient rel-L2 vs float64: hip {e_hip:.2e}, fp32 oracle {e_o32:.2e}; worst parameter {worst[1]} {worst[0]:.2e}")
        if e_hip > max(2e-4, 3.0 * e_o32):
            bad.append(f"{name}: gradient error vs float64 {e_hip:.2e}, the fp32 oracle's own {e_o32:.2e}")
    record_bar("headline_512x1024_heads_losses_backward", **rec)
    assert not bad, "; ".join(bad)
