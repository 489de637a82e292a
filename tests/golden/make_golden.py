"""Generate golden vectors by running the REAL CoDEPS reference on CPU (development container only).

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz

The reference is imported from /root/reference through tests/golden/_ref_import.py (placeholders for
its missing third-party deps).  Inputs come from tests/golden/inputs.py (seeded numpy), so the
fixtures hold only seeds/shapes + the reference's outputs.  Nothing of the reference's source is
stored.  The GPU box never runs this file.
"""
import json
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _ref_import  # noqa: E402

_ref_import.install()
import inputs as gi  # noqa: E402

torch.set_num_threads(1)

from algos import (BinaryMaskLoss, CenterLoss, DepthAlgo, EdgeAwareSmoothnessLoss,  # noqa: E402
                   InstanceSegAlgo, OffsetLoss, ReconstructionLoss, SemanticLoss, SemanticSegAlgo,
                   SSIMLoss)
from codeps.online_adap import CodepsNet  # noqa: E402
from eval import DepthEvaluator, PanopticEvaluator, SemanticEvaluator  # noqa: E402
from misc import CameraModel, ImageWarper  # noqa: E402
from misc.image_warper import CoordinateWarper  # noqa: E402
from models import DepthHead, InstanceHead, PoseHead, SemanticHead  # noqa: E402

CPU = torch.device("cpu")
R18 = np.array([64, 64, 128, 256, 512])
R50 = np.array([64, 256, 512, 1024, 2048])
THINGS = [9, 10, 11, 12, 13]
DEPTH_KEYS = ("d_a1", "d_a2", "d_a3", "d_rmse", "d_rmse_log", "d_abs_rel", "d_sq_rel")
STUFF = list(range(9))


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def grad_stats(module):
    """[n_params, 2] float64 (random projection, l2) of every parameter gradient, in named_parameters order.
    The projection vector is seeded per row (tests/golden/inputs.py::normal(shape, 4242 + row)); unlike a plain
    sum it is not hypersensitive to sign-coherent rounding noise."""
    rows = []
    for i, (_n, p) in enumerate(module.named_parameters()):
        g = p.grad
        rows.append([0.0, 0.0] if g is None else
                    [(g.double() * gi.normal(tuple(g.shape), 4242 + i).double()).sum().item(), g.double().norm().item()])
    return np.array(rows)


def cams(intr, w, h):
    return [CameraModel.from_tensor(w, h, intr[i]) for i in range(intr.shape[0])]


# ------------------------------------------------------------------------------------------ heads
def golden_heads(tag, num_ch_enc, b, h, w, seed, full=True):
    feats = gi.feature_pyramid(num_ch_enc, b, h, w, seed)
    res = {"meta": json.dumps({"num_ch_enc": [int(c) for c in num_ch_enc], "b": b, "h": h, "w": w,
                               "seed": seed})}

    def run(name, module, fwd, pseed, train=False):
        sd = gi.fill_state_dict(gi.shapes_of(module.state_dict()), pseed)
        module.load_state_dict(sd)
        module.train(train)
        res[name + "_keys"] = json.dumps([[k, list(v)] for k, v in gi.shapes_of(module.state_dict()).items()])
        fs = [f.clone().requires_grad_(True) for f in feats]
        outs = fwd(module, fs)
        loss = 0
        for i, o in enumerate(outs):
            res[f"{name}_out{i}"] = o
            loss = loss + (o * gi.normal(tuple(o.shape), pseed * 7 + i)).sum()
        loss.backward()
        res[name + "_pgrad_stats"] = grad_stats(module)
        for i, f in enumerate(fs):
            if f.grad is not None:
                res[f"{name}_dfeat{i}"] = f.grad
        if train:   # BN running stats after one train-mode forward
            res[name + "_bn_stats"] = np.concatenate(
                [v.numpy().ravel() for k, v in module.state_dict().items()
                 if k.endswith("running_mean") or k.endswith("running_var")])
        return module

    m = run("depth", DepthHead(num_ch_enc, True), lambda m_, f: m_(f, return_disparity=True), seed + 1)
    res["depth_dw_disp"] = m.dispconvs["0"][1].weight.grad
    res["depth_dw_up10"] = m.upconvs_1["0"][0].weight.grad
    res["depth_db_up04"] = m.upconvs_0["4"][0].bias.grad
    m = run("sem", SemanticHead(num_ch_enc, 14, True, True), lambda m_, f: [m_(f)], seed + 2)
    res["sem_dw_guda"] = m.semconv_guda.weight.grad
    if not full:
        save(f"heads_{tag}", **res)
        return
    run("semng", SemanticHead(num_ch_enc, 14, True, False), lambda m_, f: [m_(f)], seed + 3)
    run("inst", InstanceHead(num_ch_enc, True), lambda m_, f: list(m_(f)), seed + 4)
    run("insttrain", InstanceHead(num_ch_enc, True), lambda m_, f: list(m_(f)), seed + 4, train=True)
    run("instnomask", InstanceHead(num_ch_enc, False), lambda m_, f: list(m_(f))[:2], seed + 5)
    run("pose", PoseHead(R18, 1, 2), lambda m_, f: [m_(f, False)], seed + 6)
    run("poseinv", PoseHead(R18, 1, 2), lambda m_, f: [m_(f, True)], seed + 6)
    save(f"heads_{tag}", **res)


# ------------------------------------------------------------------------------------------ geometry
def golden_geometry(seed=11):
    b, h, w = 2, 24, 40
    intr = gi.intrinsics(b, h, w, seed)
    cm = cams(intr, w, h)
    res = {"meta": json.dumps({"b": b, "h": h, "w": w, "seed": seed})}
    # camera model round trips (misc/camera_model.py)
    sc = cm[0].get_scaled_model_image_size(w // 2, h // 2)
    res["scaled_intr"] = np.array(list(sc.intrinsics.values()), np.float64)
    u = torch.arange(w).float().expand(h, w).contiguous()
    v = torch.arange(h).float().view(h, 1).expand(h, w).contiguous()
    rx, ry, rz = cm[1].get_viewing_ray(u.clone(), v.clone())
    res["ray"] = torch.stack([rx, ry, rz])
    pu, pv = cm[1].get_image_point(rx * 5, ry * 5, rz * 5)
    res["reproj"] = torch.stack([pu, pv])
    # warp
    depth = gi.depth_map(b, h, w, seed)
    T = gi.pose(b, seed)
    src = gi.smooth_image(b, h, w, seed + 1)
    res["grid"] = CoordinateWarper(w, h, CPU)(cm, depth, T)
    res["warped"] = ImageWarper(w, h, CPU)(cm, src, depth, T)
    res["warped_identity"] = ImageWarper(w, h, CPU)(cm, src, depth, torch.eye(4).repeat(b, 1, 1))
    # SSIM map
    a, c = gi.smooth_image(b, h, w, seed + 2), gi.smooth_image(b, h, w, seed + 3)
    res["ssim"] = SSIMLoss()(a, c)
    res["ssim_self"] = SSIMLoss()(a, a)
    save("geometry", **res)


def golden_geometry_grad(seed=12):
    """Gradients of the reference's public ImageWarper / CoordinateWarper / SSIMLoss (misc/image_warper.py:92-184,
    algos/depth.py:128-155) under fixed upstream gradients, incl. nearest mode and an object-motion map."""
    b, h, w = 2, 24, 40
    intr = gi.intrinsics(b, h, w, seed)
    cm = cams(intr, w, h)
    res = {"meta": json.dumps({"b": b, "h": h, "w": w, "seed": seed})}

    def leaves():
        return (gi.smooth_image(b, h, w, seed + 1).requires_grad_(True), gi.depth_map(b, h, w, seed).requires_grad_(True),
                gi.pose(b, seed, rot=0.05, trans=0.6).requires_grad_(True))

    src, depth, T = leaves()
    out = ImageWarper(w, h, CPU)(cm, src, depth, T)
    (out * gi.normal(tuple(out.shape), seed + 5)).sum().backward()
    res.update(warped=out, warped_dsrc=src.grad, warped_ddepth=depth.grad, warped_dT=T.grad)
    src, depth, T = leaves()
    grid = CoordinateWarper(w, h, CPU)(cm, depth, T)
    (grid * gi.normal(tuple(grid.shape), seed + 6)).sum().backward()
    res.update(grid=grid, grid_ddepth=depth.grad, grid_dT=T.grad)
    src, depth, T = leaves()
    out = ImageWarper(w, h, CPU)(cm, src, depth, T, interp_mode="nearest")
    (out * gi.normal(tuple(out.shape), seed + 5)).sum().backward()
    res.update(nearest=out, nearest_dsrc=src.grad)
    src, depth, T = leaves()
    motion = gi.normal((b, 3, h, w), seed + 7, 0.05).requires_grad_(True)
    out = ImageWarper(w, h, CPU)(cm, src, depth, T, object_motion_map=motion)
    (out * gi.normal(tuple(out.shape), seed + 5)).sum().backward()
    res.update(motion_warped=out, motion_dmotion=motion.grad, motion_ddepth=depth.grad, motion_dT=T.grad)
    a = gi.smooth_image(b, h, w, seed + 2).requires_grad_(True)
    c = gi.smooth_image(b, h, w, seed + 3).requires_grad_(True)
    sm = SSIMLoss()(a, c)
    (sm * gi.normal(tuple(sm.shape), seed + 8)).sum().backward()
    res.update(ssim=sm, ssim_da=a.grad, ssim_dc=c.grad)
    save("geometry_grad", **res)


# ------------------------------------------------------------------------------------------ losses
def golden_recon(tag, b, h, w, seed, num_scales=5):
    intr = gi.intrinsics(b, h, w, seed)
    imgs = gi.triplet(b, h, w, seed)
    depth = gi.depth_map(b, h, w, seed).requires_grad_(True)
    poses = [gi.pose(b, seed + 1).requires_grad_(True), gi.pose(b, seed + 2).requires_grad_(True)]
    rec = ReconstructionLoss(w, h, SSIMLoss(), num_scales, CPU)
    torch.manual_seed(1000 + seed)
    loss = rec(cams(intr, w, h), imgs, depth, poses)
    loss.backward()
    # the noise the reference drew (same generator state, same order)
    torch.manual_seed(1000 + seed)
    noise = {f"noise{s}": torch.randn(b, 2, h // 2 ** s, w // 2 ** s) * .00001 for s in range(num_scales)}
    disp = gi.uniform((b, 1, h, w), seed + 4, 0.01, 1.0).requires_grad_(True)
    sm = EdgeAwareSmoothnessLoss()(imgs[0], disp)
    sm.backward()
    save(f"recon_{tag}", meta=json.dumps({"b": b, "h": h, "w": w, "seed": seed, "num_scales": num_scales}),
         loss=loss, ddepth=depth.grad, dT0=poses[0].grad, dT1=poses[1].grad,
         smooth=sm, ddisp=disp.grad, **noise)


def golden_sem_inst_losses(seed=31):
    b, h, w, c = 2, 24, 40, 14
    logits = gi.normal((b, c, h, w), seed, 2.0)
    probs = torch.softmax(logits, 1).requires_grad_(True)
    target = gi.semantic_labels(b, h, w, seed, c)
    pw = gi.pixel_weights(b, h, w, seed)
    res = {"meta": json.dumps({"b": b, "h": h, "w": w, "c": c, "seed": seed})}
    for name, topk in (("topk", 0.2), ("full", 1.0)):
        sl = SemanticLoss(device=CPU, class_weights=[1.0] * c, top_k_percent_pixels=topk, ignore_labels=[])
        loss = sl(probs, target.clone(), pw)
        probs.grad = None
        loss.backward()
        res[f"sem_{name}"] = loss
        res[f"sem_{name}_dprobs"] = probs.grad.clone()
    cw = [0.5 + 0.1 * i for i in range(c)]
    sl = SemanticLoss(device=CPU, class_weights=cw, top_k_percent_pixels=0.2, ignore_labels=[])
    probs.grad = None
    loss = sl(probs, target.clone(), pw)
    loss.backward()
    res["sem_cw"] = loss
    res["sem_cw_dprobs"] = probs.grad.clone()
    res["class_weights"] = np.array(cw, np.float32)
    # ignore_labels (algos/semantic_seg.py:50-55): labels remapped to 255 in place, channels sliced away; the targets must
    # stay below the number of preserved channels or the reference's CrossEntropyLoss raises -> labels drawn from 0..10
    ign = [3, 7, 12]
    tgt_small = gi.semantic_labels(b, h, w, seed + 9, 11)
    for name, topk in (("ign_topk", 0.2), ("ign_full", 1.0)):
        sl = SemanticLoss(device=CPU, class_weights=cw, top_k_percent_pixels=topk, ignore_labels=ign)
        probs.grad = None
        tgt = tgt_small.clone()
        loss = sl(probs, tgt, pw)
        loss.backward()
        res[f"sem_{name}"] = loss
        res[f"sem_{name}_dprobs"] = probs.grad.clone()
        res[f"sem_{name}_target_after"] = tgt
    res["ignore_labels"] = np.array(ign, np.int32)
    # instance losses
    cp = gi.normal((b, 1, h, w), seed + 1).requires_grad_(True)
    ct = gi.center_heatmap(b, h, w, seed)
    op = gi.normal((b, 2, h, w), seed + 2, 10.0).requires_grad_(True)
    ot = gi.offsets(b, h, w, seed)
    tp = torch.softmax(gi.normal((b, 2, h, w), seed + 3), 1).requires_grad_(True)
    tm = (gi.semantic_labels(b, h, w, seed + 4, 2)).clone()
    lc, lo, lm = CenterLoss()(cp, ct, None), OffsetLoss()(op, ot, None), BinaryMaskLoss()(tp, tm)
    (lc + lo + lm).backward()
    res.update(center=lc, offset=lo, mask=lm, dcenter=cp.grad, doffset=op.grad, dmask=tp.grad)
    save("sem_inst_losses", **res)


# ------------------------------------------------------------------------------------------ panoptic
def golden_panoptic():
    algo = InstanceSegAlgo(CenterLoss(), OffsetLoss(),
                           PanopticEvaluator(stuff_list=STUFF, thing_list=THINGS, label_divisor=1000,
                                             void_label=-1), BinaryMaskLoss())
    res = {}
    cases = []
    h, w = 48, 96
    # (a) random smooth case, uint8 semantics
    sem = gi.semantic_labels(2, h, w, 41, 14, block=8)
    cases.append(("rand", sem, gi.center_heatmap(2, h, w, 41, 12, 3.0), gi.offsets(2, h, w, 41, 20.0)))
    # (b) empty: no centre above threshold
    cases.append(("empty", sem, torch.full((2, 1, h, w), 0.05), gi.offsets(2, h, w, 42, 5.0)))
    # (c) plateau ties: equal-valued neighbouring maxima (NMS keeps all equal maxima)
    ctr = torch.zeros(1, 1, h, w)
    ctr[0, 0, 10, 10] = 0.9
    ctr[0, 0, 10, 11] = 0.9
    ctr[0, 0, 30, 50] = 0.5
    ctr[0, 0, 33, 52] = 0.5
    ctr[0, 0, 0, 0] = 0.3
    ctr[0, 0, h - 1, w - 1] = 0.3
    cases.append(("ties", sem[:1], ctr, torch.zeros(1, 2, h, w)))
    # (d) more than 200 centres: isolated peaks on a stride-8 lattice (12x24=288 > 200) with many
    #     equal scores so that the strict '>' rule on the 200-th score matters
    ctr = torch.zeros(1, 1, 96, 192)
    g = np.random.default_rng(43)
    vals = g.choice(np.array([0.2, 0.4, 0.6, 0.8], np.float32), (12, 24))
    ctr[0, 0, 4::8, 4::8] = torch.from_numpy(vals)
    sem_big = gi.semantic_labels(1, 96, 192, 44, 14, block=16, void_frac=0.1)
    cases.append(("many", sem_big, ctr, gi.offsets(1, 96, 192, 45, 6.0)))
    # (e) int64 semantics (adapt path passes argmax int64, algos/semantic_seg.py:213)
    cases.append(("int64", sem.long(), gi.center_heatmap(2, h, w, 46, 6, 4.0), gi.offsets(2, h, w, 46, 30.0)))
    # (f) all-thing image with a single centre and all-stuff image
    sem_t = torch.full((2, h, w), 11, dtype=torch.uint8)
    sem_t[1] = 3
    cases.append(("uniform", sem_t, gi.center_heatmap(2, h, w, 47, 2, 5.0), gi.offsets(2, h, w, 47, 3.0)))
    names = []
    for name, s, c, o in cases:
        pan, inst, pts = algo.panoptic_fusion(s.clone(), c.clone(), o.clone(), return_center=True)
        names.append(name)
        res[f"{name}_sem"], res[f"{name}_ctr"], res[f"{name}_off"] = s, c, o
        res[f"{name}_pan"], res[f"{name}_inst"] = pan, inst
        for i, p in enumerate(pts):
            res[f"{name}_pts{i}"] = p.reshape(-1, 2) if p.numel() else np.zeros((0, 2), np.int64)
    res["meta"] = json.dumps({"cases": names, "thing_list": THINGS})
    save("panoptic", **res)


# ------------------------------------------------------------------------------------------ network-level
class _StandInEncoder(torch.nn.Module):
    """R18-shaped trunk built from the oracle's torchvision restatement -- a stand-in, since
    torchvision is absent; network-level goldens pin everything *around* the trunk."""

    def __init__(self, in_ch, seed):
        super().__init__()
        from oracle import resnet
        self._resnet = resnet
        sd = gi.fill_state_dict(gi.shapes_of(resnet.make_state_dict(18, in_ch, 0, with_fc=False)), seed)
        self.sd = {k: (torch.nn.Parameter(v) if v.is_floating_point() and "running" not in k else v)
                   for k, v in sd.items()}
        for k, v in self.sd.items():
            if isinstance(v, torch.nn.Parameter):
                self.register_parameter(k.replace(".", "_"), v)
        self.num_ch_enc = R18

    def forward(self, x):
        return self._resnet.features(self.sd, x, 18, training=False)


def _build_net(seed, use_thing_mask, w, h, adaptation=False, src_wh=None, depth_ranges=None):
    enc = _StandInEncoder(3, seed + 1)
    penc = _StandInEncoder(6, seed + 2)
    heads = OrderedDict(depth_head=DepthHead(R18, True), pose_head=PoseHead(R18, 1, 2),
                        semantic_head=SemanticHead(R18, 14, True, True),
                        instance_head=InstanceHead(R18, use_thing_mask))
    for i, m in enumerate(heads.values()):
        m.load_state_dict(gi.fill_state_dict(gi.shapes_of(m.state_dict()), seed + 10 + i))
    ssim = SSIMLoss()
    rec = ReconstructionLoss(w, h, ssim, 5, CPU)
    rec_src = ReconstructionLoss(src_wh[0], src_wh[1], ssim, 5, CPU) if adaptation else None
    dalgo = DepthAlgo(rec, EdgeAwareSmoothnessLoss(), DepthEvaluator(True, depth_ranges), None, None, rec_src, None)
    salgo = SemanticSegAlgo(SemanticLoss(device=CPU, class_weights=[1.0] * 14, top_k_percent_pixels=0.2,
                                         ignore_labels=[]), SemanticEvaluator(num_classes=14, ignore_classes=[]))
    ialgo = InstanceSegAlgo(CenterLoss(), OffsetLoss(),
                            PanopticEvaluator(stuff_list=STUFF, thing_list=THINGS, label_divisor=1000,
                                              void_label=-1), BinaryMaskLoss())
    net = CodepsNet(None, enc, penc, heads["depth_head"], heads["pose_head"], None,
                    heads["semantic_head"], heads["instance_head"], dalgo, salgo, ialgo)
    net.eval()
    return net


class _LogitTap:
    """north_star states its tolerance on LOGITS: forward hooks on the reference's own modules record the pre-sigmoid disparity
    (models/depth_head.py:71-74: `self.dispconvs["0"](x)`) and the pre-softmax semantic features (models/semantic_head.py:72-74:
    `sem_feat`) of every head call; `depth_logit` / `semantic_logit` pick the call that produced a given result tensor."""

    def __init__(self, net):
        self.disp, self.sem, self._h = [], [], []
        self._h.append(net.depth_head.dispconvs["0"].register_forward_hook(lambda m, i, o: self.disp.append(o.detach().clone())))
        sh = getattr(net, "semantic_head", None)
        if sh is not None:
            conv = sh.semconv_guda if sh.use_guda_fusion else sh.semconv
            self._h.append(conv.register_forward_hook(lambda m, i, o: self.sem.append(o.detach().clone())))

    def close(self):
        for h in self._h:
            h.remove()

    def depth_logit(self, depth):
        for lg in self.disp:
            if lg.shape == depth.shape and torch.equal(DepthHead.disp_to_depth(torch.sigmoid(lg)), depth.detach()):
                return lg
        for lg in self.disp:                                  # the result may be one item of a batched call
            if lg.shape[1:] == depth.shape[1:]:
                d = DepthHead.disp_to_depth(torch.sigmoid(lg))
                for b0 in range(0, lg.shape[0] - depth.shape[0] + 1):
                    if torch.equal(d[b0:b0 + depth.shape[0]], depth.detach()):
                        return lg[b0:b0 + depth.shape[0]]
        raise AssertionError("no recorded disparity logit reproduces the result's depth map")

    def semantic_logit(self, semantic):
        want = semantic.detach().long()
        for lg in self.sem:
            if lg.shape[0] == want.shape[0] and lg.shape[-2:] == want.shape[-2:] and torch.equal(torch.softmax(lg, 1).argmax(1), want.reshape(lg.shape[0], *lg.shape[-2:])):
                return lg
        raise AssertionError("no recorded semantic logit reproduces the result's label map")


def _sample(b, h, w, seed, labels=True):
    f0, fm, fp = gi.triplet(b, h, w, seed)
    d = {"rgb": {0: f0, -1: fm, 1: fp}, "camera_model": gi.intrinsics(b, h, w, seed)}
    if labels:
        d.update(semantic=gi.semantic_labels(b, h, w, seed), semantic_weights=gi.pixel_weights(b, h, w, seed),
                 center=gi.center_heatmap(b, h, w, seed), offset=gi.offsets(b, h, w, seed),
                 center_weights=torch.ones(b, 1, h, w, dtype=torch.uint8),
                 offset_weights=torch.ones(b, 1, h, w, dtype=torch.uint8),
                 thing_mask=(gi.semantic_labels(b, h, w, seed + 1, 2, void_frac=0.02)).unsqueeze(1))
    return d


def golden_network(seed=51):
    b, h, w = 2, 64, 128
    net = _build_net(seed, True, w, h)
    sample = _sample(b, h, w, seed)
    torch.manual_seed(2000 + seed)
    tap = _LogitTap(net)
    losses, result, _stats = net(sample, "train", [0, -1, 1], do_panoptic_fusion=True)
    tap.close()
    wts = dict(depth_recon=10.0, depth_smth=0.001, semantic=1.0, center=20.0, offset=0.1, thing_mask=0.5)
    total = sum(wts[k] * v for k, v in losses.items() if v is not None and k in wts)
    total.backward()
    res = {"meta": json.dumps({"b": b, "h": h, "w": w, "seed": seed, "weights": wts,
                               "noise_seed": 2000 + seed})}
    for k, v in losses.items():
        if v is not None:
            res["loss_" + k] = v
    res["total"] = total
    for k in ("depth", "semantic", "center", "offset", "panoptic", "instance", "thing_mask"):
        res["res_" + k] = result[k]
    res["res_disp_logit"], res["res_sem_logit"] = tap.depth_logit(result["depth"]), tap.semantic_logit(result["semantic"])
    for name in ("depth_head", "pose_head", "semantic_head", "instance_head", "backbone_po_depth",
                 "backbone_pose_sflow"):
        res["pgrad_" + name] = grad_stats(getattr(net, name))
    save("network_train", **res)


def golden_config0(seed=71):
    """BASELINE configs[0]: one 192x640 frame triplet, ResNet-18 encoder, depth head only (no semantic / instance head),
    CodepsNet.forward("train") + backward of the reference on the CPU."""
    b, h, w = 1, 192, 640
    enc, penc = _StandInEncoder(3, seed + 1), _StandInEncoder(6, seed + 2)
    dh, ph = DepthHead(R18, True), PoseHead(R18, 1, 2)
    for i, m in enumerate((dh, ph)):
        m.load_state_dict(gi.fill_state_dict(gi.shapes_of(m.state_dict()), seed + 10 + i))
    rec = ReconstructionLoss(w, h, SSIMLoss(), 5, CPU)
    dalgo = DepthAlgo(rec, EdgeAwareSmoothnessLoss(), DepthEvaluator(True, None), None, None, None, None)
    net = CodepsNet(None, enc, penc, dh, ph, None, None, None, dalgo, None, None)
    net.eval()
    sample = _sample(b, h, w, seed, labels=False)
    torch.manual_seed(3000 + seed)
    tap = _LogitTap(net)
    losses, result, _stats = net(sample, "train", [0, -1, 1], do_panoptic_fusion=False)
    tap.close()
    wts = dict(depth_recon=10.0, depth_smth=0.001)
    total = sum(wts[k] * v for k, v in losses.items() if v is not None and k in wts)
    total.backward()
    res = {"meta": json.dumps({"b": b, "h": h, "w": w, "seed": seed, "weights": wts, "noise_seed": 3000 + seed})}
    for k, v in losses.items():
        if v is not None:
            res["loss_" + k] = v
    res["total"] = total
    res["res_depth"] = result["depth"]
    res["res_disp_logit"] = tap.depth_logit(result["depth"])
    for name in ("depth_head", "pose_head", "backbone_po_depth", "backbone_pose_sflow"):
        res["pgrad_" + name] = grad_stats(getattr(net, name))
    save("config0", **res)


def golden_adapt(seed=61):
    """DepthAlgo.adaptation weighted mean over keys with two resolutions + semantic adaptation losses
    (algos/depth.py:507-595, algos/semantic_seg.py:200-249), mixup items given pre-mixed."""
    ht, wt, hs, ws = 64, 160, 64, 128
    net = _build_net(seed, False, wt, ht, adaptation=True, src_wh=(ws, hs))
    data = {"target": _sample(1, ht, wt, seed, labels=False),
            "target_replay": _sample(2, ht, wt, seed + 1, labels=False),
            "source": _sample(2, hs, ws, seed + 2)}
    mix = _sample(2, hs, ws, seed + 3)
    torch.manual_seed(3000 + seed)
    # the reference's adapt() runs Mixup.do_mixup (datasets/mixup.py, out of scope §8f-2) on '*mixup'
    # keys; we pin the rest by calling the three algos exactly as CodepsNet.adapt does.
    images = {k: [v["rgb"][o] for o in (0, -1, 1)] for k, v in data.items()}
    feats = {k: [net.backbone_po_depth(v["rgb"][o]) for o in (0, -1, 1)] for k, v in data.items()}
    feats["cut_mixup"] = [net.backbone_po_depth(mix["rgb"][0])]
    cm = {k: cams(v["camera_model"], v["rgb"][0].shape[3], v["rgb"][0].shape[2]) for k, v in data.items()}
    sem_gt = {"target": None, "target_replay": None, "source": data["source"]["semantic"],
              "cut_mixup": mix["semantic"]}
    sem_w = {"source": data["source"]["semantic_weights"]}
    tap = _LogitTap(net)
    sem_losses, _cm, sem_pred = net.semantic_algo.adaptation(feats, net.semantic_head, sem_gt, sem_w)
    dl, _fl, depth_pred, transf, _om, _ = net.depth_algo.adaptation(
        images, feats, cm, net.depth_head, net.backbone_pose_sflow, net.pose_head, None)
    tap.close()
    wts = dict(depth_recon=10.0, depth_smth=0.001, semantic_source=1.0, semantic_cut_mixup=1.0)
    total = 10.0 * dl["recon"] + 0.001 * dl["smth"] + sem_losses["source"] + sem_losses["cut_mixup"]
    total.backward()
    with torch.no_grad():           # top-1 / top-2 probability margin of the target prediction: argmax ties are fp32 noise
        top2 = net.semantic_head(feats["target"][0]).topk(2, dim=1).values
    res = {"meta": json.dumps({"ht": ht, "wt": wt, "hs": hs, "ws": ws, "seed": seed, "weights": wts,
                               "noise_seed": 3000 + seed}),
           "loss_depth_recon": dl["recon"], "loss_depth_smth": dl["smth"],
           "loss_semantic_source": sem_losses["source"], "loss_semantic_cut_mixup": sem_losses["cut_mixup"],
           "total": total, "res_depth": depth_pred, "res_semantic": sem_pred,
           "res_disp_logit": tap.depth_logit(depth_pred), "res_sem_logit": tap.semantic_logit(sem_pred),
           "res_semantic_margin": top2[:, 0] - top2[:, 1], "res_T0": transf[0], "res_T1": transf[1]}
    for name in ("depth_head", "pose_head", "semantic_head", "backbone_pose_sflow"):
        res["pgrad_" + name] = grad_stats(getattr(net, name))
    save("network_adapt", **res)


def golden_adapt_real(seed=63):
    """The REAL CodepsNet.adapt (codeps/online_adap.py:291-436) on target / target_replay / source items at two resolutions
    and a RAW cut_mixup item (Mixup.do_mixup runs inside adapt, online_adap.py:331-335), followed by the step's loss
    weighting and backward (scripts/adapt_codeps.py:266-288)."""
    ht, wt, hs, ws = 64, 160, 64, 128
    net = _build_net(seed, False, wt, ht, adaptation=True, src_wh=(ws, hs))
    net.cfg_mixup = _cfg_mixup(True)
    data = {"target": _sample(1, ht, wt, seed, labels=False),
            "target_replay": _sample(2, ht, wt, seed + 1, labels=False),
            "source": _sample(2, hs, ws, seed + 2),
            "cut_mixup": gi.mixup_item(2, hs, ws, ht, wt, seed + 3)}
    torch.manual_seed(3000 + seed)
    np.random.seed(seed)
    tap = _LogitTap(net)
    losses, result, _stats, out_data = net(data, "adapt", [0, -1, 1], do_panoptic_fusion=True)
    tap.close()
    wts = dict(depth_recon=10.0, depth_smth=0.001, semantic_source=1.0, semantic_cut_mixup=1.0, center_source=20.0, offset_source=0.1)
    live = OrderedDict((k, v.mean()) for k, v in losses.items() if v is not None and v.requires_grad)
    total = sum(wts[k] * v for k, v in live.items())
    total.backward()
    with torch.no_grad():
        top2 = net.semantic_head(net.backbone_po_depth(data["target"]["rgb"][0])).topk(2, dim=1).values
    res = {"meta": json.dumps({"ht": ht, "wt": wt, "hs": hs, "ws": ws, "seed": seed, "weights": wts, "noise_seed": 3000 + seed,
                               "np_seed": seed, "live": list(live), "loss_keys": [k for k, v in losses.items() if v is not None]}),
           "total": total, "res_semantic_margin": top2[:, 0] - top2[:, 1]}
    for k, v in losses.items():
        if v is not None:
            res["loss_" + k] = v
    for k in ("depth", "semantic", "center", "offset", "panoptic", "instance", "image_features"):
        res["res_" + k] = result[k]
    res["res_disp_logit"], res["res_sem_logit"] = tap.depth_logit(result["depth"]), tap.semantic_logit(result["semantic"])
    res["mix_rgb"], res["mix_semantic"] = out_data["cut_mixup"]["rgb"][0], out_data["cut_mixup"]["semantic"]
    for name in ("depth_head", "pose_head", "semantic_head", "backbone_pose_sflow"):
        res["pgrad_" + name] = grad_stats(getattr(net, name))
    save("network_adapt_real", **res)


def golden_eval_infer(seed=67):
    """the reference's own CodepsNet.forward in modes "eval" (with depth / semantic ground truth, panoptic
    fusion, class-wise depth statistics) and "infer", and the teacher / student sequence of scripts/adapt_codeps.py:164-199
    that fills the mix-up item's pseudo labels -- run line by line with the reference's model objects (the script's
    `adapt()` is not importable without its data loaders; the calls below are its lines 166-199 with `ema_modules =
    ["semantic_head"]`, the shipped configuration)."""
    import copy
    h, w, hs, ws = 64, 160, 64, 128
    net = _build_net(seed, False, w, h, adaptation=True, src_wh=(ws, hs), depth_ranges=[0.1, 80])
    teacher = _build_net(seed, False, w, h, adaptation=True, src_wh=(ws, hs), depth_ranges=[0.1, 80])     # = CodepsNet.ema_model(net)
    with torch.no_grad():                                   # an EMA'd semantic head differs from the student's
        for p_ in teacher.semantic_head.parameters():
            p_.mul_(0.97)
    target = _sample(1, h, w, seed)                          # labels present: semantic GT, instance GT
    target["depth"] = torch.where(gi.uniform((1, 1, h, w), seed + 5) < 0.3, torch.zeros(1, 1, h, w), gi.depth_map(1, h, w, seed + 4))[:, 0]   # [B,H,W]
    res = {"meta": json.dumps({"h": h, "w": w, "hs": hs, "ws": ws, "seed": seed, "teacher_scale": 0.97})}
    teacher.eval()
    with torch.no_grad():
        tap_t, tap_s = _LogitTap(teacher), _LogitTap(net)
        _, r_eval, s_eval = teacher(target, "eval", [0, -1, 1], True, True)
        _, r_inf, s_inf = net.eval()(target, "infer", [0, -1, 1], True, False)
        tap_t.close(), tap_s.close()
        res["eval_disp_logit"], res["eval_sem_logit"] = tap_t.depth_logit(r_eval["depth"]), tap_t.semantic_logit(r_eval["semantic"])
        res["infer_disp_logit"], res["infer_sem_logit"] = tap_s.depth_logit(r_inf["depth"]), tap_s.semantic_logit(r_inf["semantic"])
        assert s_inf["sem_conf"] is None
        mix = gi.mixup_item(2, hs, ws, h, w, seed + 3)
        for k in ("semantic_pred_tgt_ema", "center_pred_tgt_ema", "offset_pred_tgt_ema"):
            mix.pop(k)
        _, r_pl, _ = teacher(mix, "eval", [0, -1, 1], False, False)         # adapt_codeps.py:173-177 (key_rgb = "rgb_tgt")
        mix["semantic_pred_tgt_ema"] = r_pl["semantic"].detach().clone()
    _, r_st, _ = net(mix, "eval", [0, -1, 1], False, False)                # adapt_codeps.py:193-199 (student, grad enabled)
    for k in ("depth", "semantic", "semantic_soft", "center", "offset", "panoptic", "instance"):
        res["eval_" + k] = r_eval[k]
        res["infer_" + k] = r_inf[k]
    res["eval_sem_conf"] = s_eval["sem_conf"]
    stat_keys = [k for k in s_eval if k != "sem_conf"]
    res["eval_stat_keys"] = json.dumps(stat_keys)
    res["eval_stats"] = np.array([float(s_eval[k]) for k in stat_keys], np.float64)
    with torch.no_grad():
        top2 = teacher.semantic_head(teacher.backbone_po_depth(target["rgb"][0])).topk(2, dim=1).values
        res["eval_margin"] = top2[:, 0] - top2[:, 1]
        top2 = net.semantic_head(net.backbone_po_depth(target["rgb"][0])).topk(2, dim=1).values
        res["infer_margin"] = top2[:, 0] - top2[:, 1]
        top2 = teacher.semantic_head(teacher.backbone_po_depth(mix["rgb_tgt"][0])).topk(2, dim=1).values
        res["mix_margin"] = top2[:, 0] - top2[:, 1]
    res["mix_semantic_pred_tgt_ema"] = mix["semantic_pred_tgt_ema"]
    res["mix_center_pred_tgt_ema"], res["mix_offset_pred_tgt_ema"] = r_st["center"].detach(), r_st["offset"].detach()
    save("eval_infer", **res)
    del copy


def golden_preprocessing(seed=91):
    """what the reference's data loader does to a decoded image, produced by the libraries it calls -- Pillow for
    `transforms.Resize(size, LANCZOS)` (datasets/dataset.py:66-67) and, behind torchvision 0.13.1's F.adjust_* wrappers,
    for the colour jitter (datasets/preprocessing.py:78-105); ToTensor + Normalize written out (preprocessing.py:12-18).
    torchvision itself is not installed (its wrappers are one Pillow call each and are written out below)."""
    from PIL import Image, ImageEnhance
    h, w, size = 96, 160, (48, 96)
    r = np.random.default_rng(seed)
    a = np.kron(r.random((h // 8 + 2, w // 8 + 2, 3)), np.ones((8, 8, 1)))[:h, :w]
    a = (a * 255 + r.normal(0, 8, (h, w, 3))).clip(0, 255).astype(np.uint8)
    img = Image.fromarray(a).resize((size[1], size[0]), Image.LANCZOS)
    resized = np.array(img)
    ops = [("contrast", 1.13), ("hue", -0.06), ("brightness", 0.85), ("saturation", 1.17)]
    for name, f in ops:
        if name == "brightness":
            img = ImageEnhance.Brightness(img).enhance(f)
        elif name == "contrast":
            img = ImageEnhance.Contrast(img).enhance(f)
        elif name == "saturation":
            img = ImageEnhance.Color(img).enhance(f)
        else:                                               # F.adjust_hue (functional_pil.py of torchvision 0.13.1)
            hh, ss, vv = img.convert("HSV").split()
            np_h = np.array(hh, dtype=np.uint8)
            np_h = (np_h.astype(np.int32) + (int(f * 255) & 0xFF)).astype(np.uint8)      # np_h += np.uint8(hue_factor * 255)
            img = Image.merge("HSV", (Image.fromarray(np_h, "L"), ss, vv)).convert("RGB")
    jittered = np.array(img)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    ten = torch.from_numpy(jittered).permute(2, 0, 1).float().div(255)                 # ToTensor
    ten = (ten - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)   # Normalize
    save("preprocessing", meta=json.dumps({"h": h, "w": w, "seed": seed, "size": list(size), "ops": [list(o) for o in ops],
                                           "mean": list(mean), "std": list(std)}),
         resized=resized, jittered=jittered, tensor=ten)


# ------------------------------------------------------------------------------------------ mix-up
def _cfg_mixup(geom):
    from yacs.config import CfgNode as CN
    return CN(general=CN(geom_augment=geom), cut_mix=CN(nof_vert_splits=2, nof_hor_splits=4, nof_segments=2),
              class_mix=CN(conf_thresh=0.98), conf_instance_mix=CN(conf_thresh=0.98, min_inst_size=100))


def golden_mixup():
    from datasets.mixup import Mixup
    algo = InstanceSegAlgo(CenterLoss(), OffsetLoss(),
                           PanopticEvaluator(stuff_list=STUFF, thing_list=THINGS, label_divisor=1000,
                                             void_label=-1), BinaryMaskLoss())
    res, meta = {}, {"cases": {}, "thing_list": THINGS}
    for name, (b, hs, ws, ht, wt, geom, seed) in {"geom": (2, 48, 96, 40, 112, True, 71), "plain": (2, 48, 96, 48, 96, False, 72),
                                                  "geom_same": (1, 64, 128, 64, 128, True, 73)}.items():
        item = gi.mixup_item(b, hs, ws, ht, wt, seed)
        np.random.seed(seed)
        out = Mixup.do_mixup("cut_mixup", {k: (dict(v) if isinstance(v, dict) else v) for k, v in item.items()}, algo,
                             _cfg_mixup(geom))
        meta["cases"][name] = dict(b=b, hs=hs, ws=ws, ht=ht, wt=wt, geom=geom, seed=seed)
        res[f"{name}_rgb"], res[f"{name}_semantic"], res[f"{name}_instance"] = out["rgb"][0], out["semantic"], out["instance"]
        res[f"{name}_center"], res[f"{name}_offset"] = out["center"], out["offset"]
        if geom:                                   # the float64 warps on their own (Mixup.warp_c2c, three mode combinations)
            cs = [CameraModel.from_tensor(ws, hs, t) for t in item["camera_model_src"]]
            ct = [CameraModel.from_tensor(wt, ht, t) for t in item["camera_model_tgt"]]
            _, inst_t = algo.panoptic_fusion(item["semantic_pred_tgt_ema"].clone(), item["center_pred_tgt_ema"].clone(),
                                             item["offset_pred_tgt_ema"].clone())
            res[f"{name}_inst_tgt"] = inst_t
            res[f"{name}_warp_rgb"] = Mixup.warp_c2c(ct, cs, item["rgb_tgt"][0], item["rgb_src"][0], interp_mode="bilinear",
                                                     padding_mode="zeros")
            res[f"{name}_warp_rgb_border"] = Mixup.warp_c2c(ct, cs, item["rgb_tgt"][0], item["rgb_src"][0])
            res[f"{name}_warp_inst"] = Mixup.warp_c2c(ct, cs, inst_t, item["instance_src"], interp_mode="nearest",
                                                      padding_mode="zeros")
            res[f"{name}_warp_sem"] = Mixup.warp_c2c(ct, cs, item["semantic_pred_tgt_ema"], item["semantic_src"],
                                                     interp_mode="nearest", padding_mode="border")
    # class_mixup (mixup.py:261-311) on the "geom" and "plain" inputs
    for name, (b, hs, ws, ht, wt, geom, seed) in {"cls_geom": (2, 48, 96, 40, 112, True, 76), "cls_plain": (2, 48, 96, 40, 112, False, 77)}.items():
        item = gi.mixup_item(b, hs, ws, ht, wt, seed)
        np.random.seed(seed)
        out = Mixup.do_mixup("class_mixup", {k: (dict(v) if isinstance(v, dict) else v) for k, v in item.items()}, algo, _cfg_mixup(geom))
        meta["cases"][name] = dict(b=b, hs=hs, ws=ws, ht=ht, wt=wt, geom=geom, seed=seed, strategy="class_mixup")
        res[f"{name}_rgb"], res[f"{name}_semantic"], res[f"{name}_instance"] = out["rgb"][0], out["semantic"], out["instance"]
        res[f"{name}_center"], res[f"{name}_offset"] = out["center"], out["offset"]
    # Mixup.get_off_cen on its own: ids at the image border (clipped Gaussians), one-pixel instances, an id split in two parts
    inst = torch.zeros(2, 1, 72, 120, dtype=torch.int32)
    inst[0, 0, 0:5, 0:9] = 26001
    inst[0, 0, 60:72, 100:120] = 26002
    inst[0, 0, 30:41, 50:53] = 24001
    inst[0, 0, 10, 60] = 7
    inst[0, 0, 20:24, 10:30] = 25001
    inst[0, 0, 50:54, 90:110] = 25001
    inst[1, 0, :, :] = gi.instance_labels(gi.semantic_labels(1, 72, 120, 74), 75)[0, 0]
    ctr, off = Mixup.get_off_cen(inst)
    res["offcen_inst"], res["offcen_center"], res["offcen_offset"] = inst, ctr, off
    res["meta"] = json.dumps(meta)
    save("mixup", **res)


def golden_conf_mixup():
    """Mixup.do_mixup("conf_instance_mixup") of the real reference (datasets/mixup.py:347-404, 542-660) with the oracle's
    distance transform standing in for kornia's (see _ref_import.py).  B = 1: the reference reads batch item 0's instance map
    for every item's centre (mixup.py:581)."""
    from datasets.mixup import Mixup
    from yacs.config import CfgNode as CN
    algo = InstanceSegAlgo(CenterLoss(), OffsetLoss(),
                           PanopticEvaluator(stuff_list=STUFF, thing_list=THINGS, label_divisor=1000,
                                             void_label=-1), BinaryMaskLoss())
    res, meta = {}, {"cases": {}, "thing_list": THINGS}
    for name, (hs, ws, ht, wt, geom, seed, thr, min_size) in {"geom": (64, 128, 56, 120, True, 171, 0.3, 12),
                                                              "plain": (64, 128, 64, 128, False, 172, 0.3, 12),
                                                              "big": (96, 160, 96, 160, False, 173, 0.1, 40)}.items():
        item = gi.mixup_item(1, hs, ws, ht, wt, seed)
        cfg = CN(general=CN(geom_augment=geom), conf_instance_mix=CN(conf_thresh=thr, min_inst_size=min_size))
        out = Mixup.do_mixup("conf_instance_mixup", {k: (dict(v) if isinstance(v, dict) else v) for k, v in item.items()}, algo, cfg)
        meta["cases"][name] = dict(hs=hs, ws=ws, ht=ht, wt=wt, geom=geom, seed=seed, conf_thresh=thr, min_inst_size=min_size)
        res[f"{name}_rgb"], res[f"{name}_semantic"], res[f"{name}_instance"] = out["rgb"][0], out["semantic"], out["instance"]
        res[f"{name}_center"], res[f"{name}_offset"] = out["center"], out["offset"]
        pasted = int((out["instance"] != item["instance_src"].squeeze(1)).sum())
        print(name, "pixels whose instance id changed:", pasted, "ids:", out["instance"].unique().tolist()[:12])
    res["meta"] = json.dumps(meta)
    save("conf_mixup", **res)


# ------------------------------------------------------------------------------------------ metrics
def golden_metrics():
    res, meta = {}, {"depth": {}, "thing_list": THINGS, "stuff_list": STUFF}
    b, h, w = 2, 48, 96
    gt = gi.depth_map(b, h, w, 81)
    gt = torch.where(gi.uniform((b, 1, h, w), 82) < 0.3, torch.zeros_like(gt), gt)          # 30 % void
    pred = (gi.depth_map(b, h, w, 81) * (0.4 + 0.2 * gi.uniform((b, 1, h, w), 83)) + 0.05).contiguous()
    sem = gi.semantic_labels(b, h, w, 84)
    for name, (scale, garg) in {"scaled": (True, False), "garg": (True, True), "raw": (False, False)}.items():
        ev = DepthEvaluator(scale, [0.1, 80], garg)
        out = ev.compute_depth_metrics(gt.clone(), pred.clone())
        meta["depth"][name] = dict(use_gt_scale=scale, garg=garg)
        res[f"depth_{name}"] = np.array([float(out[k]) for k in DEPTH_KEYS], np.float64)
    ev = DepthEvaluator(True, [0.1, 80])
    pc = ev.compute_depth_metrics_per_class(gt[:, 0].clone(), pred.clone(), sem)
    meta["per_class_keys"] = sorted(pc)
    res["depth_per_class"] = np.array([float(pc[k]) for k in sorted(pc)], np.float64)
    # confusion matrix
    sev = SemanticEvaluator(14, ignore_classes=[3])
    spred = gi.semantic_labels(b, h, w, 85, void_frac=0.0)
    res["conf"] = sev.compute_confusion_matrix(spred, sem)
    res["conf_noignore"] = SemanticEvaluator(14).compute_confusion_matrix(spred, sem)
    res["miou"] = sev.compute_sem_miou(res["conf"].clone())
    # panoptic quality
    pgt, ppred = gi.panoptic_maps(2, 64, 128, 86, shift=1)
    pev = PanopticEvaluator(stuff_list=STUFF, thing_list=THINGS, label_divisor=1000, void_label=-1)
    pev.update(pgt.clone(), ppred.clone())
    pev.update(ppred.clone(), pgt.clone())
    cats = sorted(pev.pq_stats.pq_per_cat)
    res["pq_cats"] = np.array(cats)
    res["pq_iou"] = np.array([pev.pq_stats[c].iou for c in cats], np.float64)
    res["pq_counts"] = np.array([[pev.pq_stats[c].tp, pev.pq_stats[c].fp, pev.pq_stats[c].fn] for c in cats])
    ev = pev.evaluate()
    res["pq_all"] = np.array([ev["All"]["pq"], ev["All"]["sq"], ev["All"]["rq"], ev["All"]["n"]], np.float64)
    res["pq_things"] = np.array([ev["Things"]["pq"], ev["Things"]["sq"], ev["Things"]["rq"], ev["Things"]["n"]], np.float64)
    res["meta"] = json.dumps(meta)
    save("metrics", **res)


# ------------------------------------------------------------------------------------------ replay buffer
REPLAY_CASES = {   # tag: (capacity, channels, h, w, threshold, frames, seed)
    "small": (6, 64, 4, 8, 0.97, 60, 101),
    "full": (16, 2048, 16, 32, 0.95, 40, 102),          # the shipped feature size: ResNet-101 layer4 of a 512x1024 frame
    "keepall": (5, 48, 3, 5, 2.0, 25, 103),             # threshold never rejects: an eviction per frame once full
}


def _reference_replay_buffer(capacity, threshold, n_target):
    """The reference's ReplayBuffer with the fields its `add` / `load_state` / `save_state` read, without the data-set half of its
    constructor (files, samplers, multiprocessing arrays)."""
    from datasets.replay_buffer import ReplayBuffer
    rb = object.__new__(ReplayBuffer)
    rb.maximize_diversity, rb.similarity_threshold, rb.target_buffer_size = True, threshold, capacity
    rb.faiss_index = rb.distance_matrix = rb.distance_matrix_indices = None
    rb.buffer_indices = {"target": torch.zeros(n_target, dtype=torch.int32)}
    rb.target_data = None
    return rb


def _replay_run(rb, frames, first_index=0):
    """ReplayBuffer.add frame by frame -> per-frame (stored, evicted id, top-1 similarity, eviction margin)."""
    import contextlib
    import io
    stored, evicted, sims, margins = [], [], [], []
    for i, f in enumerate(frames):
        idx = first_index + i
        idx_obj = rb.faiss_index
        before_ids = [] if idx_obj is None else idx_obj._ids.tolist()
        before_x = None if idx_obj is None else idx_obj.index._x.copy()
        nlog = 0 if idx_obj is None else len(idx_obj.top1_log)
        with contextlib.redirect_stdout(io.StringIO()):
            rb.add(idx, f)
        after_ids = rb.faiss_index._ids.tolist()
        was_stored = bool(rb.buffer_indices["target"][idx] == 1) or (idx in after_ids)
        gone = [j for j in before_ids + ([idx] if was_stored else []) if j not in after_ids]
        assert len(gone) <= 1
        # stored-and-immediately-evicted frames: the reference clears buffer_indices["target"][idx] again
        if not was_stored and idx_obj is not None and len(rb.faiss_index.top1_log) > nlog and rb.faiss_index.top1_log[-1] < rb.similarity_threshold:
            was_stored, gone = True, [idx]
        stored.append(was_stored)
        evicted.append(gone[0] if gone else -1)
        sims.append(0.0 if (idx_obj is None or len(before_ids) == 0) else rb.faiss_index.top1_log[-1])
        margin = -1.0
        if gone:                                                # gap between the two largest eviction scores, in float64
            q = f.mean(-1).mean(-1).numpy().astype(np.float64)
            q /= np.linalg.norm(q)
            x = np.vstack([before_x.astype(np.float64), q])
            m = x @ x.T
            sc = np.sort(m.sum(0) - m.diagonal())
            margin = float(sc[-1] - sc[-2])
        margins.append(margin)
    return dict(stored=np.array(stored, np.int8), evicted=np.array(evicted, np.int64), sim=np.array(sims, np.float32),
                margin=np.array(margins, np.float64))


def golden_replay():
    """the reference's OWN ReplayBuffer.add / save_state / load_state (datasets/replay_buffer.py:387-499) over frame
    sequences with rejections and evictions, on the brute-force faiss stand-in of _ref_import.py.  Inputs: inputs.replay_features
    (seeds in the meta record); outputs: the decision per frame, the final index, the saved state, and a second-domain run that
    starts from the loaded state (ids 10000 + idx, "hardcoded continual learning")."""
    import tempfile
    res, meta = {}, {"cases": {k: dict(zip(("capacity", "c", "h", "w", "threshold", "n", "seed"), v)) for k, v in REPLAY_CASES.items()}}
    for tag, (cap, c, h, w, thr, n, seed) in REPLAY_CASES.items():
        rb = _reference_replay_buffer(cap, thr, 1000)
        out = _replay_run(rb, gi.replay_features(n, c, h, w, seed))
        for k, v in out.items():
            res[f"{tag}_{k}"] = v
        res[f"{tag}_final_ids"] = rb.faiss_index._ids.copy()
        res[f"{tag}_final_vecs"] = rb.faiss_index.index._x.copy()
        res[f"{tag}_final_target"] = rb.buffer_indices["target"].numpy().copy()
        print(f"replay {tag}: stored {int(out['stored'].sum())} / {n}, evictions {int((out['evicted'] >= 0).sum())}, "
              f"smallest eviction margin {min([m for m in out['margin'] if m >= 0], default=-1):.2e}")
    # second domain: save_state of the "small" run -> load_state into a fresh buffer -> 20 more frames
    cap, c, h, w, thr, n, seed = REPLAY_CASES["small"]
    rb = _reference_replay_buffer(cap, thr, 1000)
    _replay_run(rb, gi.replay_features(n, c, h, w, seed))
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            rb.save_state()
            rb2 = _reference_replay_buffer(cap, thr, 1000)
            rb2.load_state()
        finally:
            os.chdir(cwd)
    res["domain2_loaded_ids"] = rb2.faiss_index._ids.copy()
    res["domain2_prev_target"] = np.array(rb2.buffer_indices["prev_target"], np.int64)
    meta["domain2"] = dict(from_case="small", n=20, seed=104)
    out = _replay_run(rb2, gi.replay_features(20, c, h, w, 104))
    for k, v in out.items():
        res[f"domain2_{k}"] = v
    res["domain2_final_ids"] = rb2.faiss_index._ids.copy()
    res["domain2_final_prev_target"] = np.array(rb2.buffer_indices["prev_target"], np.int64)
    res["domain2_final_target"] = rb2.buffer_indices["target"].numpy().copy()
    print(f"replay domain2: stored {int(out['stored'].sum())} / 20, evictions {int((out['evicted'] >= 0).sum())}, of previous-domain "
          f"samples {int((out['evicted'] >= 10000).sum())}")
    res["meta"] = json.dumps(meta)
    save("replay", **res)


if __name__ == "__main__":
    jobs = {"replay": golden_replay, "conf_mixup": golden_conf_mixup, "config0": golden_config0, "geometry": golden_geometry, "geometry_grad": golden_geometry_grad, "network_adapt_real": golden_adapt_real, "eval_infer": golden_eval_infer, "preprocessing": golden_preprocessing,
            "heads_r18": lambda: golden_heads("r18", R18, 2, 32, 64, 3),
            "heads_r50": lambda: golden_heads("r50", R50, 1, 32, 64, 5, full=False),
            "recon_small": lambda: golden_recon("small", 2, 48, 80, 21),
            "recon_odd": lambda: golden_recon("odd", 1, 40, 72, 22, num_scales=3),
            "sem_inst_losses": golden_sem_inst_losses, "panoptic": golden_panoptic,
            "network_train": golden_network, "network_adapt": golden_adapt, "mixup": golden_mixup, "metrics": golden_metrics}
    for name in (sys.argv[1:] or jobs):          # no arguments = regenerate everything
        jobs[name]()
