"""The CPU oracle (oracle/) against golden vectors produced by the REAL reference
(tests/golden/make_golden.py).  This is what pins the oracle."""
import json

import numpy as np
import pytest
import torch

import inputs as gi
from helpers import close, grad_stats, head_shapes, keys_of, load, t
from oracle import heads, losses, network, panoptic, resnet

torch.set_num_threads(1)


def _run_head(z, meta, name, fn, pseed, feats):
    shapes = keys_of(z, name)
    sd = gi.fill_state_dict(shapes, pseed)
    params = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k]
    for k in params:
        sd[k].requires_grad_(True)
    fs = [f.clone().requires_grad_(True) for f in feats]
    outs = fn(sd, fs)
    loss = 0
    for i, o in enumerate(outs):
        close(o, z[f"{name}_out{i}"], 2e-5, 1e-5, f"{name}_out{i}")
        loss = loss + (o * gi.normal(tuple(o.shape), pseed * 7 + i)).sum()
    loss.backward()
    st = grad_stats([sd[k].grad for k in params])
    ref = z[name + "_pgrad_stats"]
    scale = np.abs(ref[:, 1:2]) + 1e-6
    assert (np.abs(st - ref) / scale).max() < 2e-3, f"{name}: param-grad stats differ"
    for i, f in enumerate(fs):
        if f"{name}_dfeat{i}" in z.files:
            close(f.grad, z[f"{name}_dfeat{i}"], 1e-4, 1e-4, f"{name}_dfeat{i}")
    return sd


@pytest.mark.parametrize("tag", ["r18", "r50"])
def test_heads(tag):
    z, meta = load("heads_" + tag)
    seed = meta["seed"]
    feats = gi.feature_pyramid(meta["num_ch_enc"], meta["b"], meta["h"], meta["w"], seed)
    sd = _run_head(z, meta, "depth", lambda s, f: heads.depth_head(s, f), seed + 1, feats)
    close(sd["dispconvs.0.1.weight"].grad, z["depth_dw_disp"], 1e-4, 1e-4)
    close(sd["upconvs_1.0.0.weight"].grad, z["depth_dw_up10"], 1e-4, 1e-4)
    close(sd["upconvs_0.4.0.bias"].grad, z["depth_db_up04"], 1e-4, 1e-4)
    assert sd["dispconvs.1.1.weight"].grad is None        # dead scales 1..3
    sd = _run_head(z, meta, "sem", lambda s, f: [heads.semantic_head(s, f, True)], seed + 2, feats)
    close(sd["semconv_guda.weight"].grad, z["sem_dw_guda"], 1e-4, 1e-4)
    if tag == "r50":
        return
    _run_head(z, meta, "semng", lambda s, f: [heads.semantic_head(s, f, False)], seed + 3, feats)
    _run_head(z, meta, "inst", lambda s, f: list(heads.instance_head(s, f, True, False)), seed + 4, feats)
    sd = _run_head(z, meta, "insttrain", lambda s, f: list(heads.instance_head(s, f, True, True)), seed + 4,
                   feats)
    bn = np.concatenate([v.detach().numpy().ravel() for k, v in sd.items()
                         if k.endswith("running_mean") or k.endswith("running_var")])
    close(bn, z["insttrain_bn_stats"], 1e-5, 1e-5, "BN running stats")
    _run_head(z, meta, "instnomask", lambda s, f: list(heads.instance_head(s, f, False, False))[:2], seed + 5,
              feats)
    _run_head(z, meta, "pose", lambda s, f: [heads.pose_head(s, f, False)], seed + 6, feats)
    _run_head(z, meta, "poseinv", lambda s, f: [heads.pose_head(s, f, True)], seed + 6, feats)


def test_geometry():
    z, meta = load("geometry")
    b, h, w, seed = meta["b"], meta["h"], meta["w"], meta["seed"]
    intr = gi.intrinsics(b, h, w, seed)
    close(losses.scale_intrinsics(intr, w, h, w // 2, h // 2)[0], z["scaled_intr"].astype(np.float32), 0, 1e-6)
    depth, T, src = gi.depth_map(b, h, w, seed), gi.pose(b, seed), gi.smooth_image(b, h, w, seed + 1)
    pts = losses.backproject(intr, torch.ones(b, 1, h, w))[1]       # depth 1 -> ray / ray_z
    close(pts / pts.norm(dim=0, keepdim=True), z["ray"], 1e-6, 1e-6, "viewing ray")
    close(losses.warp_grid(intr, depth, T), z["grid"], 2e-5, 1e-5, "grid")
    close(losses.warp_image(intr, src, depth, T), z["warped"], 1e-4, 0, "warped")
    close(losses.warp_image(intr, src, depth, torch.eye(4).repeat(b, 1, 1)), z["warped_identity"], 1e-4)
    close(losses.warp_image(intr, src, depth, torch.eye(4).repeat(b, 1, 1)), src, 2e-4, 0, "identity warp")
    a, c = gi.smooth_image(b, h, w, seed + 2), gi.smooth_image(b, h, w, seed + 3)
    close(losses.ssim_loss(a, c), z["ssim"], 1e-6, 1e-5, "ssim")
    close(losses.ssim_loss(a, a), z["ssim_self"], 1e-6)


@pytest.mark.parametrize("tag", ["small", "odd"])
def test_reconstruction_and_smoothness(tag):
    z, meta = load("recon_" + tag)
    b, h, w, seed, ns = meta["b"], meta["h"], meta["w"], meta["seed"], meta["num_scales"]
    intr, imgs = gi.intrinsics(b, h, w, seed), gi.triplet(b, h, w, seed)
    depth = gi.depth_map(b, h, w, seed).requires_grad_(True)
    poses = [gi.pose(b, seed + 1).requires_grad_(True), gi.pose(b, seed + 2).requires_grad_(True)]
    noise = [t(z[f"noise{s}"]) for s in range(ns)]
    loss = losses.reconstruction_loss(intr, imgs, depth, poses, ns, noise)
    loss.backward()
    close(loss, z["loss"], 1e-6, 1e-5, "recon loss")
    close(depth.grad, z["ddepth"], 1e-8, 1e-3, "d loss / d depth")
    close(poses[0].grad, z["dT0"], 1e-5, 1e-3, "dT0")
    close(poses[1].grad, z["dT1"], 1e-5, 1e-3, "dT1")
    # same generator stream as the reference => same noise
    torch.manual_seed(1000 + seed)
    loss2 = losses.reconstruction_loss(intr, imgs, depth.detach(), [p.detach() for p in poses], ns, None)
    close(loss2, z["loss"], 1e-6, 1e-5, "recon loss (drawn noise)")
    disp = gi.uniform((b, 1, h, w), seed + 4, 0.01, 1.0).requires_grad_(True)
    sm = losses.smoothness_loss(imgs[0], disp)
    sm.backward()
    close(sm, z["smooth"], 1e-6, 1e-5)
    close(disp.grad, z["ddisp"], 1e-7, 1e-3)


def test_semantic_and_instance_losses():
    z, meta = load("sem_inst_losses")
    b, h, w, c, seed = meta["b"], meta["h"], meta["w"], meta["c"], meta["seed"]
    probs = torch.softmax(gi.normal((b, c, h, w), seed, 2.0), 1).requires_grad_(True)
    target, pw = gi.semantic_labels(b, h, w, seed, c), gi.pixel_weights(b, h, w, seed)
    for name, topk, cw in (("topk", 0.2, [1.0] * c), ("full", 1.0, [1.0] * c),
                           ("cw", 0.2, z["class_weights"].tolist())):
        probs.grad = None
        loss = losses.semantic_loss(probs, target, pw, cw, topk)
        loss.backward()
        close(loss, z["sem_" + name], 1e-6, 1e-5, name)
        close(probs.grad, z[f"sem_{name}_dprobs"], 1e-8, 1e-4, name + " grad")
    ign = z["ignore_labels"].tolist()                       # algos/semantic_seg.py:50-55
    for name, topk in (("ign_topk", 0.2), ("ign_full", 1.0)):
        probs.grad = None
        tgt = gi.semantic_labels(b, h, w, seed + 9, 11)
        loss = losses.semantic_loss(probs, tgt, pw, z["class_weights"].tolist(), topk, ignore_labels=ign)
        loss.backward()
        close(loss, z["sem_" + name], 1e-6, 1e-5, name)
        close(probs.grad, z[f"sem_{name}_dprobs"], 1e-8, 1e-4, name + " grad")
        assert np.array_equal(tgt.numpy(), z[f"sem_{name}_target_after"]), "targets are remapped in place"
    cp = gi.normal((b, 1, h, w), seed + 1).requires_grad_(True)
    op = gi.normal((b, 2, h, w), seed + 2, 10.0).requires_grad_(True)
    tp = torch.softmax(gi.normal((b, 2, h, w), seed + 3), 1).requires_grad_(True)
    lc = losses.center_loss(cp, gi.center_heatmap(b, h, w, seed))
    lo = losses.offset_loss(op, gi.offsets(b, h, w, seed))
    lm = losses.binary_mask_loss(tp, gi.semantic_labels(b, h, w, seed + 4, 2))
    (lc + lo + lm).backward()
    for a, k in ((lc, "center"), (lo, "offset"), (lm, "mask")):
        close(a, z[k], 1e-6, 1e-5, k)
    close(cp.grad, z["dcenter"], 1e-9, 1e-4)
    close(op.grad, z["doffset"], 1e-9, 1e-4)
    close(tp.grad, z["dmask"], 1e-9, 1e-4)


def test_panoptic_fusion_bit_exact():
    z, meta = load("panoptic")
    for name in meta["cases"]:
        pan, inst, pts = panoptic.panoptic_fusion(z[name + "_sem"], z[name + "_ctr"], z[name + "_off"],
                                                  meta["thing_list"])
        assert pan.dtype == np.int16 and inst.dtype == z[name + "_inst"].dtype
        assert np.array_equal(pan, z[name + "_pan"]), f"{name}: panoptic ids differ"
        assert np.array_equal(inst, z[name + "_inst"]), f"{name}: instance ids differ"
        for i, p in enumerate(pts):
            assert np.array_equal(p, z[f"{name}_pts{i}"]), f"{name}: centres differ"


def _sds(seed, use_thing_mask):
    """Rebuild the state dicts the golden generator used (same seeds, same key order)."""
    sds = {"backbone_po_depth": gi.fill_state_dict(gi.shapes_of(resnet.make_state_dict(18, 3, 0, False)), seed + 1),
           "backbone_pose_sflow": gi.fill_state_dict(gi.shapes_of(resnet.make_state_dict(18, 6, 0, False)), seed + 2)}
    for i, name in enumerate(("depth_head", "pose_head", "semantic_head", "instance_head")):
        key = name if name != "instance_head" or use_thing_mask else "instance_head_nomask"
        sds[name] = gi.fill_state_dict(head_shapes(key), seed + 10 + i)
    return sds


def _sample(b, h, w, seed, labels=True):
    f0, fm, fp = gi.triplet(b, h, w, seed)
    d = {"rgb": {0: f0, -1: fm, 1: fp}, "camera_model": gi.intrinsics(b, h, w, seed)}
    if labels:
        d.update(semantic=gi.semantic_labels(b, h, w, seed), semantic_weights=gi.pixel_weights(b, h, w, seed),
                 center=gi.center_heatmap(b, h, w, seed), offset=gi.offsets(b, h, w, seed),
                 thing_mask=gi.semantic_labels(b, h, w, seed + 1, 2, void_frac=0.02).unsqueeze(1))
    return d


def _req_grad(sds):
    groups = {}
    for name, sd in sds.items():
        ps = [k for k, v in sd.items() if v.is_floating_point() and "running" not in k]
        for k in ps:
            sd[k].requires_grad_(True)
        groups[name] = ps
    return groups


def _draw_noise(seed, shapes):
    torch.manual_seed(seed)
    return [torch.randn(s) * .00001 for s in shapes]


def test_network_train_step():
    z, meta = load("network_train")
    b, h, w, seed = meta["b"], meta["h"], meta["w"], meta["seed"]
    sds = _sds(seed, True)
    groups = _req_grad(sds)
    cfg = dict(nof_layers=18, use_thing_mask=True)
    noise = _draw_noise(meta["noise_seed"], [(b, 2, h // 2 ** s, w // 2 ** s) for s in range(5)])
    out, res = network.train_forward(sds, cfg, _sample(b, h, w, seed), noise)
    total = sum(meta["weights"][k] * v for k, v in out.items())
    total.backward()
    for k, v in out.items():
        close(v, z["loss_" + k], 1e-6, 2e-5, "loss " + k)
    close(total, z["total"], 1e-5, 2e-5)
    close(res["depth"], z["res_depth"], 1e-3, 0, "depth")
    close(res["center"], z["res_center"], 1e-4, 1e-4)
    close(res["offset"], z["res_offset"], 1e-4, 1e-4)
    for k in ("semantic", "panoptic", "instance", "thing_mask"):
        assert np.array_equal(res[k].numpy(), z["res_" + k]), k
    for name, ps in groups.items():
        st = grad_stats([sds[name][k].grad for k in ps])
        ref = z["pgrad_" + name]
        assert (np.abs(st - ref) / (np.abs(ref[:, 1:2]) + 1e-6)).max() < 5e-3, name


def test_config0_depth_only_train_step():
    """BASELINE configs[0] (one 192x640 triplet, ResNet-18, depth head only) as the reference's CodepsNet computes it."""
    z, meta = load("config0")
    b, h, w, seed = meta["b"], meta["h"], meta["w"], meta["seed"]
    sds = _sds(seed, False)
    for name in ("semantic_head", "instance_head"):
        sds.pop(name)
    groups = _req_grad(sds)
    noise = _draw_noise(meta["noise_seed"], [(b, 2, h // 2 ** s, w // 2 ** s) for s in range(5)])
    out, res = network.train_forward(sds, dict(nof_layers=18), _sample(b, h, w, seed, labels=False), noise, do_panoptic=False)
    assert list(out) == ["depth_recon", "depth_smth"]
    total = sum(meta["weights"][k] * v for k, v in out.items())
    total.backward()
    for k, v in out.items():
        close(v, z["loss_" + k], 1e-6, 2e-5, "loss " + k)
    close(total, z["total"], 1e-5, 2e-5)
    close(res["depth"], z["res_depth"], 1e-3, 0, "depth")
    for name, ps in groups.items():
        st = grad_stats([sds[name][k].grad for k in ps])
        ref = z["pgrad_" + name]
        assert (np.abs(st - ref) / (np.abs(ref[:, 1:2]) + 1e-6)).max() < 5e-3, name


def test_network_adapt_step():
    z, meta = load("network_adapt")
    ht, wt, hs, ws, seed = meta["ht"], meta["wt"], meta["hs"], meta["ws"], meta["seed"]
    sds = _sds(seed, False)
    groups = _req_grad(sds)
    data = {"target": _sample(1, ht, wt, seed, False), "target_replay": _sample(2, ht, wt, seed + 1, False),
            "source": _sample(2, hs, ws, seed + 2), "cut_mixup": _sample(2, hs, ws, seed + 3)}
    # the reference drew noise per key in the order source, target, target_replay (algos/depth.py:514)
    torch.manual_seed(meta["noise_seed"])
    noise = {}
    for key, (bb, hh, ww) in (("source", (2, hs, ws)), ("target", (1, ht, wt)), ("target_replay", (2, ht, wt))):
        noise[key] = [torch.randn(bb, 2, hh // 2 ** s, ww // 2 ** s) * .00001 for s in range(5)]
    out, res = network.adapt_forward(sds, dict(nof_layers=18), data, noise)
    total = sum(meta["weights"][k] * out[k] for k in meta["weights"])
    total.backward()
    for k in meta["weights"]:
        close(out[k], z["loss_" + k], 1e-6, 2e-5, k)
    close(res["depth"], z["res_depth"], 1e-3)
    close(res["T"][0], z["res_T0"], 1e-6, 1e-5)
    assert np.array_equal(res["semantic"].numpy(), z["res_semantic"])
    for name in ("depth_head", "pose_head", "semantic_head", "backbone_pose_sflow"):
        st = grad_stats([sds[name][k].grad for k in groups[name]])
        ref = z["pgrad_" + name]
        assert (np.abs(st - ref) / (np.abs(ref[:, 1:2]) + 1e-6)).max() < 5e-3, name


def test_network_adapt_real():
    """The oracle's adaptation step against the reference's own CodepsNet.adapt incl. Mixup.do_mixup on a raw item."""
    z, meta = load("network_adapt_real")
    ht, wt, hs, ws, seed = meta["ht"], meta["wt"], meta["hs"], meta["ws"], meta["seed"]
    sds = _sds(seed, False)
    groups = _req_grad(sds)
    for k, v in sds["instance_head"].items():                  # the golden's instance head is trainable
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    data = {"target": _sample(1, ht, wt, seed, False), "target_replay": _sample(2, ht, wt, seed + 1, False),
            "source": _sample(2, hs, ws, seed + 2), "cut_mixup": gi.mixup_item(2, hs, ws, ht, wt, seed + 3)}
    torch.manual_seed(meta["noise_seed"])
    noise = {}
    for key, (bb, hh, ww) in (("source", (2, hs, ws)), ("target", (1, ht, wt)), ("target_replay", (2, ht, wt))):
        noise[key] = [torch.randn(bb, 2, hh // 2 ** s, ww // 2 ** s) * .00001 for s in range(5)]
    np.random.seed(meta["np_seed"])                             # the reference's cell draw, one per batch item (mixup.py:108)
    cells = [np.random.choice(8, 2, replace=False) for _ in range(2)]
    out, res = network.adapt_forward(sds, dict(nof_layers=18), data, noise, {"cut_mixup": cells})
    assert np.array_equal(res["mix_cut_mixup"]["semantic"].numpy(), z["mix_semantic"])
    close(res["mix_cut_mixup"]["rgb"][0], z["mix_rgb"], 0, 0, "mixed image")
    total = sum(meta["weights"][k] * out[k] for k in meta["live"])
    total.backward()
    for k in meta["loss_keys"]:
        close(out[k], z["loss_" + k], 1e-6, 2e-5, k)
    close(total, z["total"], 1e-5, 2e-5)
    close(res["depth"], z["res_depth"], 1e-3)
    assert np.array_equal(res["semantic"].numpy(), z["res_semantic"])
    close(res["center"], z["res_center"], 1e-5 * float(np.abs(z["res_center"]).max()), 1e-5)
    assert np.array_equal(res["panoptic"].numpy(), z["res_panoptic"]) and np.array_equal(res["instance"].numpy(), z["res_instance"])
    close(res["image_features"], z["res_image_features"], 1e-5 * float(np.abs(z["res_image_features"]).max()), 1e-5)
    for name in ("depth_head", "pose_head", "semantic_head", "backbone_pose_sflow"):
        st = grad_stats([sds[name][k].grad for k in groups[name]])
        ref = z["pgrad_" + name]
        assert (np.abs(st - ref) / (np.abs(ref[:, 1:2]) + 1e-6)).max() < 5e-3, name


def test_eval_and_infer_modes():
    """the oracle's "eval" / "infer" composition against the reference's own CodepsNet.forward, incl. the depth
    statistics, the confusion matrix, panoptic fusion and the pseudo labels of the per-frame evaluation passes."""
    z, meta = load("eval_infer")
    h, w, hs, ws, seed = meta["h"], meta["w"], meta["hs"], meta["ws"], meta["seed"]
    sds = _sds(seed, False)
    teacher = _sds(seed, False)
    for k, v in teacher["semantic_head"].items():
        if v.is_floating_point():
            v.mul_(meta["teacher_scale"])
    cfg = dict(nof_layers=18, depth_ranges=[0.1, 80])
    target = _sample(1, h, w, seed)
    target["depth"] = torch.where(gi.uniform((1, 1, h, w), seed + 5) < 0.3, torch.zeros(1, 1, h, w), gi.depth_map(1, h, w, seed + 4))[:, 0]
    with torch.no_grad():
        r_eval, s_eval = network.eval_forward(teacher, cfg, target, "eval", True, True)
        r_inf, s_inf = network.eval_forward(sds, cfg, target, "infer", True)
        mix = gi.mixup_item(2, hs, ws, h, w, seed + 3)
        r_pl, _ = network.eval_forward(teacher, cfg, mix, "eval", False)
        r_st, _ = network.eval_forward(sds, cfg, mix, "eval", False)
    assert s_inf["sem_conf"] is None and len(s_inf) == 1
    for tag, r in (("eval", r_eval), ("infer", r_inf)):
        close(r["depth"], z[tag + "_depth"], 1e-3)
        close(r["semantic_soft"], z[tag + "_semantic_soft"], 1e-6, 1e-5)
        for k in ("semantic", "panoptic", "instance"):
            assert np.array_equal(r[k].numpy(), z[f"{tag}_{k}"]), (tag, k)
        for k in ("center", "offset"):
            close(r[k], z[f"{tag}_{k}"], 1e-5 * float(np.abs(z[f"{tag}_{k}"]).max()), 1e-5)
    assert np.array_equal(np.asarray(s_eval["sem_conf"]), z["eval_sem_conf"])
    keys = json.loads(str(z["eval_stat_keys"]))
    assert [k for k in s_eval if k != "sem_conf"] == keys
    close(np.array([float(s_eval[k]) for k in keys]), z["eval_stats"], 1e-6, 2e-5, "depth statistics")
    assert np.array_equal(r_pl["semantic"].numpy(), z["mix_semantic_pred_tgt_ema"])
    close(r_st["center"], z["mix_center_pred_tgt_ema"], 1e-5 * float(np.abs(z["mix_center_pred_tgt_ema"]).max()), 1e-5)
    close(r_st["offset"], z["mix_offset_pred_tgt_ema"], 1e-5 * float(np.abs(z["mix_offset_pred_tgt_ema"]).max()), 1e-5)


def test_resnet_self_consistency():
    """Trunk is 'parity unpinned' (no torchvision): pin what can be pinned."""
    # published torchvision conv GMACs @224^2 (fc excluded): 1.81 / 3.66 / 4.09 / 7.80 / 11.5
    for n, g in ((18, 1.81), (50, 4.09), (101, 7.80)):
        assert abs(resnet.conv_macs(n, 224, 224) / 1e9 - g) < 0.03, n
    for n, m in ((18, 11.18), (50, 23.51), (101, 42.50)):
        sd = resnet.make_state_dict(n, 3, 0, with_fc=False)
        assert abs(sum(v.numel() for k, v in sd.items() if v.dim() > 0 and "running" not in k) / 1e6 - m) < 0.06
    sd = resnet.make_state_dict(18, 6, 1)
    assert sd["conv1.weight"].shape == (64, 6, 7, 7) and "layer2.0.downsample.0.weight" in sd
    assert "layer1.0.downsample.0.weight" not in sd
    sd50 = resnet.make_state_dict(50, 3, 1)
    assert sd50["layer1.0.downsample.0.weight"].shape == (256, 64, 1, 1)
    assert sd50["layer2.0.conv2.weight"].shape == (128, 128, 3, 3)
    f = resnet.features(sd50, torch.randn(1, 3, 64, 96), 50)
    assert [tuple(x.shape[1:]) for x in f] == [(64, 32, 48), (256, 16, 24), (512, 8, 12), (1024, 4, 6), (2048, 2, 3)]
