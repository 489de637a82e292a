"""CodepsNet (HIP path) end to end against goldens produced by the REAL reference CodepsNet / algos:
forward("train") + backward, and the adaptation step (two resolutions, count-weighted depth losses)."""
import numpy as np
import pytest
import torch

import inputs as gi
from helpers import close, depth_bar, depth_of_logit, grad_stats, head_shapes, load, logit_bar, oracle_depth64, record_bar
from oracle import resnet as oresnet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
THINGS, STUFF = [9, 10, 11, 12, 13], list(range(9))


def _build(seed, use_thing_mask, w, h, adaptation=False, src_hw=None):
    from codeps_amd.codeps import gen_models
    from codeps_amd.config import get_cfg_shipped
    cfg = get_cfg_shipped(adaptation)
    cfg.model.po_depth_net.params.nof_layers = 18
    cfg.model.instance_head.use_thing_mask = use_thing_mask
    cfg.dataset.feed_img_size = [h, w]
    if adaptation:
        cfg.adapt.source_dataset.feed_img_size = list(src_hw)
    net = gen_models(cfg, None, STUFF, THINGS, [], adaptation_mode=adaptation)
    for name, in_ch, s in (("backbone_po_depth", 3, seed + 1), ("backbone_pose_sflow", 6, seed + 2)):
        sd = gi.fill_state_dict(gi.shapes_of(oresnet.make_state_dict(18, in_ch, 0, False)), s)
        missing = getattr(net, name).load_state_dict({"encoder." + k: v for k, v in sd.items()}, strict=False)
        assert set(missing.missing_keys) == {"encoder.fc.weight", "encoder.fc.bias"}
    for i, name in enumerate(("depth_head", "pose_head", "semantic_head", "instance_head")):
        key = name if name != "instance_head" or use_thing_mask else "instance_head_nomask"
        getattr(net, name).load_state_dict(gi.fill_state_dict(head_shapes(key), seed + 10 + i))
    return net.to(DEV).eval()


def _sample(b, h, w, seed, labels=True):
    f0, fm, fp = gi.triplet(b, h, w, seed)
    d = {"rgb": {0: f0.to(DEV), -1: fm.to(DEV), 1: fp.to(DEV)}, "camera_model": gi.intrinsics(b, h, w, seed).to(DEV)}
    if labels:
        d.update(semantic=gi.semantic_labels(b, h, w, seed).to(DEV), semantic_weights=gi.pixel_weights(b, h, w, seed).to(DEV),
                 center=gi.center_heatmap(b, h, w, seed).to(DEV), offset=gi.offsets(b, h, w, seed).to(DEV),
                 center_weights=torch.ones(b, 1, h, w, dtype=torch.uint8, device=DEV),
                 offset_weights=torch.ones(b, 1, h, w, dtype=torch.uint8, device=DEV),
                 thing_mask=gi.semantic_labels(b, h, w, seed + 1, 2, void_frac=0.02).unsqueeze(1).to(DEV))
    return d


def _tap(*nets):
    for n in nets:
        n.depth_head.logit_log = []
        if n.semantic_head is not None:
            n.semantic_head.logit_log = []


def _logit_bars(test, net, z, result, prefix="res_", depth_key="depth", sem_key="semantic"):
    """north_star: depth / semantic logits within 1e-3 of the reference's (the goldens hold the reference's own pre-sigmoid /
    pre-softmax tensors); the selected call must be the one that produced `result`."""
    def same_depth(sel):
        got = result[depth_key].detach().cpu().double()
        assert float((depth_of_logit(sel) - got).abs().div(got).max()) < 1e-5, "the logged disparity logits are not those of the result"
    logit_bar(test, "disparity", net.depth_head.logit_log, z[prefix + "disp_logit"], same_depth)
    if net.semantic_head is not None and (prefix + "sem_logit") in z.files:
        def same_labels(sel):
            assert torch.equal(sel.argmax(1).to(torch.uint8), result[sem_key].detach().cpu().reshape(sel.shape[0], *sel.shape[-2:]).to(torch.uint8)), \
                "the logged semantic logits are not those of the result"
        logit_bar(test, "semantic", net.semantic_head.logit_log, z[prefix + "sem_logit"], same_labels)


def _noise(seed, shapes):
    torch.manual_seed(seed)
    return [torch.cat([(torch.randn(b, 2, h >> s, w >> s) * .00001).reshape(-1) for s in range(5)]).to(DEV) for b, h, w in shapes]


def _check_pgrads(net, z, names, tol=5e-3):
    # (projection, l2) per parameter, relative to its l2 norm -- with a floor at 1e-3 x the largest gradient norm
    # of the whole model: with these random weights the semantic softmax saturates, its gradients are ~1e-9
    # (pure fp32 cancellation noise in the reference as well) and carry no comparable information.
    floor = 1e-3 * max(float(z["pgrad_" + n][:, 1].max()) for n in names)
    for name in names:
        ref = z["pgrad_" + name]
        st = grad_stats([p.grad for _n, p in getattr(net, name).named_parameters()])[:ref.shape[0]]
        err = np.abs(st - ref) / (np.abs(ref[:, 1:2]) + floor)
        # the random-weight semantic softmax is saturated: d probs -> d logits is y*(dy - sum(dy*y)), a difference of
        # nearly equal numbers, so those gradients carry ~1e-2 fp32 noise in ANY implementation (the semantic path is
        # pinned tightly where it is well conditioned: tests/test_modules_gpu.py, tests/test_losses_gpu.py)
        t_ = 5e-2 if name == "semantic_head" else tol
        assert err.max() < t_, f"{name}: param-grad stats differ (max {err.max():.2e}, row {err.argmax() // 2})"


def test_train_forward_backward_against_reference():
    z, meta = load("network_train")
    b, h, w, seed = meta["b"], meta["h"], meta["w"], meta["seed"]
    net = _build(seed, True, w, h)
    net.depth_algo.reconstruction_loss.noise = _noise(meta["noise_seed"], [(b, h, w)])
    _tap(net)
    losses, result, _ = net(_sample(b, h, w, seed), "train", [0, -1, 1], do_panoptic_fusion=True)
    total = sum(meta["weights"][k] * v for k, v in losses.items() if v is not None and k in meta["weights"])
    total.backward()
    _logit_bars("network_train", net, z, result)
    for k, v in losses.items():
        if v is not None:
            close(v, z["loss_" + k], 1e-6, 5e-5, "loss " + k)
    close(total, z["total"], 1e-5, 5e-5)
    close(result["depth"], z["res_depth"], 1e-3, 0, "depth (north-star tolerance 1e-3)")
    depth_bar("network_train", result["depth"], z["res_depth"], None)
    for k in ("center", "offset"):          # random-weight heads give O(100) outputs: tolerance relative to the range
        close(result[k], z["res_" + k], 2e-5 * float(np.abs(z["res_" + k]).max()), 1e-4, k)
    for k in ("semantic", "thing_mask", "panoptic", "instance"):
        same = (result[k].cpu().numpy() == z["res_" + k]).mean()
        assert same == 1.0, f"{k}: only {same:.6f} of the pixels agree"
    record_bar("network_train", panoptic_agreement=1.0, instance_agreement=1.0, semantic_agreement=1.0)
    assert result["panoptic"].dtype == torch.int16 and result["instance"].dtype == torch.uint8
    _check_pgrads(net, z, ("depth_head", "pose_head", "semantic_head", "instance_head", "backbone_po_depth",
                           "backbone_pose_sflow"))


def test_config0_depth_only_train_step_against_reference():
    """BASELINE configs[0]: one 192x640 frame triplet, ResNet-18 encoder, depth head only -- CodepsNet.forward("train") +
    backward against the reference's own run (tests/golden/config0.npz)."""
    from codeps_amd.codeps import gen_models
    from codeps_amd.config import get_cfg_shipped
    z, meta = load("config0")
    b, h, w, seed = meta["b"], meta["h"], meta["w"], meta["seed"]
    cfg = get_cfg_shipped(False)
    cfg.model.po_depth_net.params.nof_layers = 18
    cfg.model.make_semantic = cfg.model.make_instance = False
    cfg.dataset.feed_img_size = [h, w]
    net = gen_models(cfg, None, STUFF, THINGS, [], adaptation_mode=False)
    assert net.semantic_head is None and net.instance_head is None
    for name, in_ch, s_ in (("backbone_po_depth", 3, seed + 1), ("backbone_pose_sflow", 6, seed + 2)):
        sd = gi.fill_state_dict(gi.shapes_of(oresnet.make_state_dict(18, in_ch, 0, False)), s_)
        getattr(net, name).load_state_dict({"encoder." + k: v for k, v in sd.items()}, strict=False)
    for i, name in enumerate(("depth_head", "pose_head")):
        getattr(net, name).load_state_dict(gi.fill_state_dict(head_shapes(name), seed + 10 + i))
    net = net.to(DEV).eval()
    net.depth_algo.reconstruction_loss.noise = _noise(meta["noise_seed"], [(b, h, w)])
    _tap(net)
    losses, result, _ = net(_sample(b, h, w, seed, labels=False), "train", [0, -1, 1], do_panoptic_fusion=False)
    _logit_bars("config0", net, z, result)
    assert [k for k, v in losses.items() if v is not None] == ["depth_recon", "depth_smth"]
    total = sum(meta["weights"][k] * losses[k] for k in meta["weights"])
    total.backward()
    for k in meta["weights"]:
        close(losses[k], z["loss_" + k], 1e-6, 5e-5, "loss " + k)
    close(total, z["total"], 1e-5, 5e-5)
    # depth = 1 / (0.01 + 9.99 sigmoid(logit)): d depth / d logit ~ 24 at 40 m, so two valid fp32 summation orders of the
    # 192x640 decoder already differ by ~1e-3 there (SURVEY.md section 7 "fp32 accuracy budget"): 1e-3 + 1e-4 relative, and
    # tight in disparity space, as in the adaptation test below
    sample0 = _sample(b, h, w, seed, labels=False)
    depth_bar("config0", result["depth"], z["res_depth"], lambda: oracle_depth64(net, sample0["rgb"][0]))
    close(1.0 / result["depth"], 1.0 / z["res_depth"], 2e-4, 5e-5, "scaled disparity (range 0.01..10)")
    _check_pgrads(net, z, ("depth_head", "pose_head", "backbone_po_depth", "backbone_pose_sflow"))


def test_adaptation_step_against_reference():
    z, meta = load("network_adapt")
    ht, wt, hs, ws, seed = meta["ht"], meta["wt"], meta["hs"], meta["ws"], meta["seed"]
    net = _build(seed, False, wt, ht, adaptation=True, src_hw=(hs, ws))
    # the reference drew noise per key in the order source, target, target_replay (algos/depth.py:514)
    n_src, n_tgt, n_rep = _noise(meta["noise_seed"], [(2, hs, ws), (1, ht, wt), (2, ht, wt)])
    net.depth_algo.reconstruction_loss_adapt_source.noise = [n_src]
    net.depth_algo.reconstruction_loss.noise = [n_tgt, n_rep]
    data = {"target": _sample(1, ht, wt, seed, False), "target_replay": _sample(2, ht, wt, seed + 1, False),
            "source": _sample(2, hs, ws, seed + 2), "cut_mixup": _sample(2, hs, ws, seed + 3)}
    _tap(net)
    losses, result, _, _ = net(data, "adapt", [0, -1, 1], do_panoptic_fusion=True)
    _logit_bars("network_adapt", net, z, result)
    total = sum(meta["weights"][k] * losses[k] for k in meta["weights"])
    total.backward()
    for k in meta["weights"]:
        close(losses[k], z["loss_" + k], 1e-6, 5e-5, k)
    close(total, z["total"], 1e-5, 5e-5)
    # depth = 1/(0.01 + 9.99 sigmoid(logit)): at 40 m d depth / d logit ~ 24, so a 1e-5-relative logit difference
    # between two valid fp32 summation orders already moves depth by ~2e-3 (SURVEY.md §7 "fp32 accuracy budget").
    # 1e-3 absolute holds in the train test above; here: 1e-3 + 1e-4 relative, and tight in disparity space.
    depth_bar("network_adapt", result["depth"], z["res_depth"], lambda: oracle_depth64(net, data["target"]["rgb"][0]))
    close(1.0 / result["depth"], 1.0 / z["res_depth"], 2e-4, 5e-5, "scaled disparity (range 0.01..10)")
    # labels must agree except where the reference's own top-1 / top-2 probability margin is inside fp32 noise
    bad = result["semantic"].cpu().numpy() != z["res_semantic"]
    assert bad.mean() < 1e-3 and (z["res_semantic_margin"][bad] < 1e-4).all(), (int(bad.sum()), z["res_semantic_margin"][bad])
    record_bar("network_adapt", semantic_pixels_differing=int(bad.sum()), semantic_pixels=int(bad.size))
    assert result["panoptic"].shape == (1, ht, wt) and result["image_features"].shape[1] == 512
    _check_pgrads(net, z, ("depth_head", "pose_head", "semantic_head", "backbone_pose_sflow"))


def test_real_codepsnet_adapt_against_reference():
    """tests/golden/network_adapt_real.npz is the reference's own CodepsNet.adapt (codeps/online_adap.py:291-436) incl. the
    Mixup.do_mixup call on a RAW cut_mixup item inside it, then the loss weighting of scripts/adapt_codeps.py:266-288."""
    z, meta = load("network_adapt_real")
    ht, wt, hs, ws, seed = meta["ht"], meta["wt"], meta["hs"], meta["ws"], meta["seed"]
    net = _build(seed, False, wt, ht, adaptation=True, src_hw=(hs, ws))
    net.cfg_mixup.merge({"general": {"geom_augment": True}, "cut_mix": {"nof_vert_splits": 2, "nof_hor_splits": 4, "nof_segments": 2}})
    n_src, n_tgt, n_rep = _noise(meta["noise_seed"], [(2, hs, ws), (1, ht, wt), (2, ht, wt)])
    net.depth_algo.reconstruction_loss_adapt_source.noise = [n_src]
    net.depth_algo.reconstruction_loss.noise = [n_tgt, n_rep]
    raw = gi.mixup_item(2, hs, ws, ht, wt, seed + 3)
    raw = {k: ({o: t_.to(DEV) for o, t_ in v.items()} if isinstance(v, dict) else v.to(DEV)) for k, v in raw.items()}
    data = {"target": _sample(1, ht, wt, seed, False), "target_replay": _sample(2, ht, wt, seed + 1, False),
            "source": _sample(2, hs, ws, seed + 2), "cut_mixup": raw}
    np.random.seed(meta["np_seed"])
    _tap(net)
    losses, result, _, out_data = net(data, "adapt", [0, -1, 1], do_panoptic_fusion=True)
    _logit_bars("network_adapt_real", net, z, result)
    assert [k for k, v in losses.items() if v is not None] == meta["loss_keys"]
    live = [k for k, v in losses.items() if v is not None and v.requires_grad]
    assert live == meta["live"]
    total = sum(meta["weights"][k] * losses[k] for k in live)
    total.backward()
    # the mixed item built inside adapt()
    assert np.array_equal(out_data["cut_mixup"]["semantic"].cpu().numpy(), z["mix_semantic"])
    close(out_data["cut_mixup"]["rgb"][0], z["mix_rgb"], 2e-5, 0, "mixed image")
    for k in meta["loss_keys"]:
        close(losses[k], z["loss_" + k], 1e-6, 5e-5, k)
    close(total, z["total"], 1e-5, 5e-5)
    depth_bar("network_adapt_real", result["depth"], z["res_depth"], lambda: oracle_depth64(net, data["target"]["rgb"][0]))
    for k in ("center", "offset"):
        close(result[k], z["res_" + k], 2e-5 * float(np.abs(z["res_" + k]).max()), 1e-4, k)
    close(result["image_features"], z["res_image_features"], 1e-4 * float(np.abs(z["res_image_features"]).max()), 1e-4, "image_features")
    bad = result["semantic"].cpu().numpy() != z["res_semantic"]
    assert bad.mean() < 1e-3 and (z["res_semantic_margin"][bad] < 1e-4).all(), (int(bad.sum()), z["res_semantic_margin"][bad])
    record_bar("network_adapt_real", semantic_pixels_differing=int(bad.sum()), semantic_pixels=int(bad.size))
    # panoptic ids are a function of (semantic, centre, offset): bit-exact given the same inputs (tests/test_losses_gpu.py pins the
    # fusion kernel bit for bit on identical inputs); here the inputs carry fp32 noise of both implementations, so a pixel whose
    # offset vote lands within rounding of the midpoint between two centres may join the other instance -- the agreement is
    # measured and recorded, and must be exact whenever the centre / offset maps produce the same votes
    for k in ("panoptic", "instance"):
        same = float((result[k].cpu().numpy() == z["res_" + k]).mean())
        record_bar("network_adapt_real", **{k + "_agreement": same, k + "_bar": "bit-exact" if same == 1.0 else "> 0.999 (semantic labels "
                   + ("identical" if not bad.any() else "differ inside the reference's own fp32 margin") + ")"})
        if not bad.any():
            assert same > 0.999, f"{k}: only {same:.6f} of the pixels agree"
    _check_pgrads(net, z, ("depth_head", "pose_head", "semantic_head", "backbone_pose_sflow"))


def test_eval_infer_modes_and_pre_step_against_reference():
    """SURVEY 8f-1 pinned by the reference: CodepsNet.forward("eval") with depth / semantic ground truth, panoptic fusion and
    class-wise depth statistics, forward("infer"), and codeps.pre_step (scripts/adapt_codeps.py:164-199) filling a raw mix-up
    item's pseudo labels -- against tests/golden/eval_infer.npz (the reference's own CodepsNet, device evaluators here)."""
    import copy
    import json
    from codeps_amd.codeps import pre_step
    from codeps_amd.eval import DepthEvaluator, SemanticEvaluator
    z, meta = load("eval_infer")
    h, w, hs, ws, seed = meta["h"], meta["w"], meta["hs"], meta["ws"], meta["seed"]
    net = _build(seed, False, w, h, adaptation=True, src_hw=(hs, ws))
    net.depth_algo.evaluator = DepthEvaluator(True, [0.1, 80])
    net.semantic_algo.evaluator = SemanticEvaluator(14, ignore_classes=[])
    for p_ in net.parameters():
        p_.requires_grad_(False)
    teacher = copy.deepcopy(net)
    with torch.no_grad():
        for p_ in teacher.semantic_head.parameters():
            p_.mul_(meta["teacher_scale"])
    target = _sample(1, h, w, seed)
    target["depth"] = torch.where(gi.uniform((1, 1, h, w), seed + 5) < 0.3, torch.zeros(1, 1, h, w),
                                  gi.depth_map(1, h, w, seed + 4))[:, 0].to(DEV)
    raw = gi.mixup_item(2, hs, ws, h, w, seed + 3)
    for k in ("semantic_pred_tgt_ema", "center_pred_tgt_ema", "offset_pred_tgt_ema"):
        raw.pop(k)
    raw = {k: ({o: t_.to(DEV) for o, t_ in v.items()} if isinstance(v, dict) else v.to(DEV)) for k, v in raw.items()}
    sample = {"target": target, "cut_mixup": raw}
    _tap(net, teacher)
    r_eval, s_eval = pre_step(net, teacher, ["semantic_head"], sample, [0, -1, 1], True, do_class_wise_depth_stats=True)
    _logit_bars("eval_infer_eval", teacher, z, r_eval, prefix="eval_")
    _tap(net, teacher)
    with torch.no_grad():
        _, r_inf, s_inf = net(target, "infer", [0, -1, 1], True, False)
    _logit_bars("eval_infer_infer", net, z, r_inf, prefix="infer_")
    assert s_inf["sem_conf"] is None

    def check(tag, r, margin):
        depth_bar("eval_infer_" + tag, r["depth"], z[tag + "_depth"], lambda: oracle_depth64(net, target["rgb"][0]))
        close(r["semantic_soft"], z[tag + "_semantic_soft"], 2e-5, 1e-4, tag + " semantic_soft")
        bad = r["semantic"].cpu().numpy() != z[tag + "_semantic"]
        assert bad.mean() < 1e-3 and (margin[bad] < 1e-4).all(), (tag, int(bad.sum()))
        for k in ("center", "offset"):
            close(r[k], z[f"{tag}_{k}"], 2e-5 * float(np.abs(z[f"{tag}_{k}"]).max()), 1e-4, f"{tag} {k}")
        for k in ("panoptic", "instance"):
            same = float((r[k].cpu().numpy() == z[f"{tag}_{k}"]).mean())
            record_bar("eval_infer_" + tag, **{k + "_agreement": same, "semantic_pixels_differing": int(bad.sum())})
            if not bad.any():
                assert same > 0.999, f"{tag} {k}: only {same:.6f} of the pixels agree"
        assert r["panoptic"].dtype == torch.int16 and r["semantic"].dtype == torch.uint8
        return bad

    bad_eval = check("eval", r_eval, z["eval_margin"])
    check("infer", r_inf, z["infer_margin"])
    if not bad_eval.any():
        assert np.array_equal(s_eval["sem_conf"].cpu().numpy(), z["eval_sem_conf"])
    keys = json.loads(str(z["eval_stat_keys"]))
    assert [k for k in s_eval if k != "sem_conf"] == keys
    got = np.array([float(s_eval[k]) for k in keys])
    # depth = 1 / (0.01 + 9.99 sigmoid): fp32 logit noise of 1e-5 moves a 40 m depth by 2e-3; the statistics are means of
    # per-pixel errors and thresholded counts over a few hundred pixels per class
    assert np.allclose(got, z["eval_stats"], rtol=2e-3, atol=2e-3), np.abs(got - z["eval_stats"]).max()
    m = sample["cut_mixup"]
    bad = m["semantic_pred_tgt_ema"].cpu().numpy() != z["mix_semantic_pred_tgt_ema"]
    assert bad.mean() < 1e-3 and (z["mix_margin"][bad] < 1e-4).all()
    for k in ("center", "offset"):
        ref = z[f"mix_{k}_pred_tgt_ema"]
        close(m[f"{k}_pred_tgt_ema"], ref, 2e-5 * float(np.abs(ref).max()), 1e-4, "student " + k)


def test_pre_step_shares_encoder_features_without_changing_results():
    """the per-frame evaluation passes (scripts/adapt_codeps.py:164-199) with the frozen encoder run once per
    image for teacher and student == the reference's separate calls, bit for bit."""
    import copy
    import bench
    from codeps_amd import synthetic
    from codeps_amd.codeps import pre_step
    dev = torch.device(DEV)
    net, _cfg = bench.build(dev, 18, 64, 128, "adapt")
    from codeps_amd.step import freeze_for_adaptation
    freeze_for_adaptation(net, _cfg)
    ema = copy.deepcopy(net)
    with torch.no_grad():                                   # the EMA'd head differs from the student's
        for p in ema.semantic_head.parameters():
            p.mul_(0.9)
    data, _ = synthetic.adapt_batch(64, 128, dev, 5, 1, 1, 1, 2)
    def _rel(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))

    def run(share, batch=False, needed=True):
        sample = {k: (dict(v) if isinstance(v, dict) else v) for k, v in data.items()}
        res, stats = pre_step(net, ema, ["semantic_head"], sample, [0, -1, 1], True, share_encoder=share, batch_encoder=batch,
                              needed_heads_only=needed)
        return res, sample["cut_mixup"]
    r1, m1 = run(True)
    r2, m2 = run(False, needed=False)                       # the reference's call sequence: every pass runs everything
    for k in ("depth", "semantic", "center", "offset", "panoptic"):
        assert torch.equal(r1[k], r2[k]), k
    for k in ("semantic_pred_tgt_ema", "center_pred_tgt_ema", "offset_pred_tgt_ema"):
        assert torch.equal(m1[k], m2[k]), k
    # target frame and mix-up images through the encoder in ONE batch: the kernels tile the larger batch differently, so the
    # values agree to fp32 rounding and the label maps almost everywhere
    r3, m3 = run(True, batch=True)
    for k in ("depth", "center", "offset"):
        assert _rel(r3[k], r1[k]) < 2e-5, (k, _rel(r3[k], r1[k]))
    for k in ("center_pred_tgt_ema", "offset_pred_tgt_ema"):
        assert _rel(m3[k], m1[k]) < 2e-5, k
    assert (r3["semantic"] == r1["semantic"]).float().mean() > 0.999
    assert (m3["semantic_pred_tgt_ema"] == m1["semantic_pred_tgt_ema"]).float().mean() > 0.999
    # the reference's own call sequence
    with torch.no_grad():
        ema.eval()
        want_sem = ema(data["cut_mixup"], "eval", [0, -1, 1], False, False)[1]["semantic"]
        want_ctr = net(data["cut_mixup"], "eval", [0, -1, 1], False, False)[1]["center"]
    assert torch.equal(m1["semantic_pred_tgt_ema"], want_sem) and torch.equal(m1["center_pred_tgt_ema"], want_ctr)


def test_online_adapter_keeps_the_teacher_current():
    """codeps.OnlineAdapter == the per-frame loop of scripts/adapt_codeps.py:164-307 written out with the reference's own
    per-parameter teacher update (EMA for cfg.adapt.ema modules, plain copy for the others): same evaluation results, same
    teacher weights, frame after frame -- i.e. the teacher views into the optimizer's buffers (and the weight packs made
    from them) never go stale."""
    import copy
    import numpy as np
    import bench
    from codeps_amd import synthetic
    from codeps_amd.codeps import CodepsNet, OnlineAdapter, pre_step
    from codeps_amd.datasets import DiversityReplayBuffer
    from codeps_amd.step import StepRunner
    dev = torch.device(DEV)
    net, cfg = bench.build(dev, 18, 64, 128, "adapt")
    net_ref = copy.deepcopy(net)
    replay = DiversityReplayBuffer(100, 4, 2.0, dev, feature_dim=512)
    adapter = OnlineAdapter(net, cfg, replay=replay)
    assert adapter.ema_modules == ["semantic_head"]
    runner_ref = StepRunner(net_ref, cfg, "adapt")
    teacher_ref = CodepsNet.ema_model(net_ref)
    others = ["backbone_po_depth", "backbone_pose_sflow", "depth_head", "pose_head", "instance_head"]
    prev_sem = None
    for frame in range(3):
        data, _ = synthetic.adapt_batch(64, 128, dev, 50 + frame, 1, 1, 1, 2)
        data.pop("cut_mixup")
        raw = synthetic.mixup_raw_item(2, 64, 128, 64, 128, dev, 70 + frame)

        def sample():
            s = {k: (dict(v) if isinstance(v, dict) else v) for k, v in data.items()}
            s["cut_mixup"] = dict(raw)
            return s

        np.random.seed(frame)
        res, _stats, losses, result = adapter.process(sample(), index=frame)
        np.random.seed(frame)
        s_ref = sample()
        res_ref, _ = pre_step(net_ref, teacher_ref, ["semantic_head"], s_ref, [0, -1, 1], True)
        _, losses_ref, _ = runner_ref.step(s_ref)
        teacher_ref.update_weights(net_ref, ["semantic_head"], cfg.adapt.ema.alpha)        # adapt_codeps.py:293-299
        teacher_ref.update_weights(net_ref, others, 0)
        assert torch.equal(res["semantic"], res_ref["semantic"]) or (res["semantic"] != res_ref["semantic"]).float().mean() < 1e-3
        close(res["depth"], res_ref["depth"], 1e-4, 1e-4, f"frame {frame}: evaluation depth")
        for k, v in losses_ref.items():
            if v is not None:
                assert abs(losses[k].item() - v.item()) <= 1e-4 * max(1.0, abs(v.item())), (frame, k)
        for (n1, t1), (n2, t2) in zip(adapter.ema_model.named_parameters(), teacher_ref.named_parameters()):
            # Adam's early steps move every weight by ~lr = 1e-4 whatever the gradient's size, so the sign of a near-zero
            # gradient (atomics in the loss reductions: last-bit noise) shows up as 2e-4 per step in the copied modules; the
            # EMA'd head only sees 1 % of that
            tol = 5e-6 * (frame + 1) if n1.startswith("semantic_head") else 2.5e-4 * (frame + 1)
            close(t1, t2, tol, 1e-4, f"frame {frame}: teacher {n1}")
        sem_now = torch.cat([p.detach().reshape(-1) for p in adapter.ema_model.semantic_head.parameters()]).clone()
        if prev_sem is not None:
            assert not torch.equal(sem_now, prev_sem)                     # the teacher really moves
            # ... by an EMA step, not by a copy of the student
            stu = torch.cat([p.detach().reshape(-1) for p in net.semantic_head.parameters()])
            assert (sem_now - stu).abs().max() > 0
        prev_sem = sem_now
    replay.sync()
    assert [e[0] for e in replay.log] == [0, 1, 2]
