"""Product nn.Modules (HIP path) against the reference's golden vectors and the CPU oracle."""
import numpy as np
import pytest
import torch

import inputs as gi
from helpers import close, grad_stats, keys_of, load

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(z, name, module, fwd, pseed, feats, out_tol=(2e-5, 2e-5), train=False):
    sd = gi.fill_state_dict(keys_of(z, name), pseed)
    module.load_state_dict(sd)
    module.to(DEV).train(train)
    fs = [f.clone().to(DEV).requires_grad_(True) for f in feats]
    outs = fwd(module, fs)
    loss = 0
    for i, o in enumerate(outs):
        close(o, z[f"{name}_out{i}"], *out_tol, f"{name}_out{i}")
        loss = loss + (o * gi.normal(tuple(o.shape), pseed * 7 + i).to(DEV)).sum()
    loss.backward()
    st = grad_stats([p.grad for _n, p in module.named_parameters()])
    ref = z[name + "_pgrad_stats"]
    err = np.abs(st - ref) / (np.abs(ref[:, 1:2]) + 1e-6)      # (projection, l2) relative to l2
    assert err.max() < 2e-3, f"{name}: param-grad stats differ (max {err.max():.2e} at row {err.argmax() // 2})"
    for i, f in enumerate(fs):
        if f"{name}_dfeat{i}" in z.files:
            close(f.grad, z[f"{name}_dfeat{i}"], 2e-4, 2e-4, f"{name}_dfeat{i}")
    return module


@pytest.mark.parametrize("tag", ["r18", "r50"])
def test_heads_against_reference_golden(tag):
    from codeps_amd.models import DepthHead, InstanceHead, PoseHead, SemanticHead
    z, meta = load("heads_" + tag)
    seed, ch = meta["seed"], np.array(meta["num_ch_enc"])
    feats = gi.feature_pyramid(ch, meta["b"], meta["h"], meta["w"], seed)
    m = _run(z, "depth", DepthHead(ch, True), lambda m_, f: m_(f, return_disparity=True), seed + 1, feats, (1e-3, 1e-5))
    close(m.dispconvs["0"][1].weight.grad, z["depth_dw_disp"], 2e-4, 2e-4)
    close(m.upconvs_1["0"][0].weight.grad, z["depth_dw_up10"], 2e-4, 2e-4)
    close(m.upconvs_0["4"][0].bias.grad, z["depth_db_up04"], 2e-4, 2e-4)
    m = _run(z, "sem", SemanticHead(ch, 14, True, True), lambda m_, f: [m_(f)], seed + 2, feats)
    close(m.semconv_guda.weight.grad, z["sem_dw_guda"], 2e-4, 2e-4)
    if tag == "r50":
        return
    _run(z, "semng", SemanticHead(ch, 14, True, False), lambda m_, f: [m_(f)], seed + 3, feats)
    _run(z, "inst", InstanceHead(ch, True), lambda m_, f: list(m_(f)), seed + 4, feats, (1e-4, 1e-4))
    m = _run(z, "insttrain", InstanceHead(ch, True), lambda m_, f: list(m_(f)), seed + 4, feats, (2e-4, 2e-4), train=True)
    bn = np.concatenate([v.detach().cpu().numpy().ravel() for k, v in m.state_dict().items()
                         if k.endswith("running_mean") or k.endswith("running_var")])
    close(bn, z["insttrain_bn_stats"], 1e-5, 1e-4, "BN running stats")
    _run(z, "instnomask", InstanceHead(ch, False), lambda m_, f: list(m_(f))[:2], seed + 5, feats, (1e-4, 1e-4))
    _run(z, "pose", PoseHead(ch, 1, 2), lambda m_, f: [m_(f, False)], seed + 6, feats, (1e-6, 1e-5))
    _run(z, "poseinv", PoseHead(ch, 1, 2), lambda m_, f: [m_(f, True)], seed + 6, feats, (1e-6, 1e-5))


@pytest.mark.parametrize("layers,in_ch,train", [(18, 3, False), (18, 6, True), (50, 3, False), (50, 3, True), (34, 3, False),
                                                (101, 3, False)])      # 101 eval = the frozen encoder of the benchmarked adapt step
def test_encoder_against_oracle(layers, in_ch, train):
    """Trunk vs the oracle's torchvision restatement (fwd, BN running stats, grads)."""
    from codeps_amd.models import ResnetEncoder
    from oracle import resnet as oresnet
    torch.set_num_threads(8)
    sd = gi.fill_state_dict(gi.shapes_of(oresnet.make_state_dict(layers, in_ch, 0)), 77 + layers)
    enc = ResnetEncoder(layers, False, in_ch // 3, 3)
    enc.load_state_dict({"encoder." + k: v for k, v in sd.items()})
    enc.to(DEV).train(train)
    h, w = (128, 192) if (train and layers >= 50) else (64, 96)     # deep train-mode BN needs enough samples per channel
    x = gi.smooth_image(2, h, w, 5, 3)
    if in_ch == 6:
        x = torch.cat([x, gi.smooth_image(2, h, w, 6, 3)], 1)
    def oracle_run(dt):
        osd = {k: ((v.clone().to(dt).requires_grad_(True) if "running" not in k else v.clone().to(dt))
                   if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        of = oresnet.features(osd, x.to(dt), layers, training=train)
        sum((f * gi.normal(tuple(f.shape), 900 + i).to(dt)).sum() for i, f in enumerate(of)).backward()
        return osd, of

    # float64 oracle = the reference value; the float32 oracle run measures how ill-conditioned this case is
    # (train-mode BN over a handful of samples amplifies rounding): relative-L2 tolerance = max(1e-3, 8 x fp32-oracle noise)
    ref_sd, rf = oracle_run(torch.float64)
    f32_sd, _ = oracle_run(torch.float32)
    feats = enc(x.to(DEV))
    sum((f * gi.normal(tuple(f.shape), 900 + i).to(DEV)).sum() for i, f in enumerate(feats)).backward()
    for i, (a, b) in enumerate(zip(feats, rf)):
        assert a.shape == b.shape
        close(a, b.detach(), 2e-4 * float(b.detach().abs().max()), 0, f"feat{i}")
    def rel(g, rg):
        return float((g.double().cpu() - rg).norm()) / (float(rg.norm()) + 1e-30)
    # how far the fp32 CPU oracle itself lands from fp64 over all parameters: the conditioning of this case
    noise_max = max(rel(f32_sd[k].grad, ref_sd[k].grad) for k, _ in enc.encoder.named_parameters() if not k.startswith("fc"))
    for k, p in enc.encoder.named_parameters():
        if k.startswith("fc"):
            assert p.grad is None
            continue
        rg = ref_sd[k].grad
        nrm = float(rg.norm()) + 1e-30
        noise = float((f32_sd[k].grad.double() - rg).norm()) / nrm          # fp32 CPU oracle vs fp64
        err = float((p.grad.double().cpu() - rg).norm()) / nrm
        # a single ReLU whose pre-activation is ~1e-7 can land on the other side of 0 under a different
        # (MFMA k-ordered) summation order; one flipped mask moves every upstream gradient by ~2e-3 rel-L2
        # (measured: scratch analysis in DESIGN.md "numerics"), hence 1e-2 here and exactness checks elsewhere
        # (tests/test_ops_gpu.py pins every op to 1e-5 in isolation).
        assert err < max(1e-2 if train else 1e-3, 8 * noise, 2 * noise_max), \
            f"grad {k}: rel-L2 err {err:.2e} (fp32 oracle noise {noise:.2e}, worst parameter {noise_max:.2e})"
    if train:
        for k, v in enc.encoder.state_dict().items():
            if "running" in k:
                close(v, ref_sd[k], 1e-5, 1e-4, k)
    else:       # the batched form CodepsNet.adapt uses for a frozen encoder (one forward over all keys' centre frames)
        with torch.no_grad():           # (both without the autograd tape: eval BatchNorm folded into the conv epilogues)
            xb = x.to(DEV)
            fb, fp = enc.forward_batch([xb[:1], xb[1:]]), enc(xb)
        for i, (a, b, c_) in enumerate(zip(fb, fp, feats)):
            assert torch.equal(a, b), f"forward_batch feat{i} differs from the plain forward"
            close(a, c_.detach(), 2e-5 * float(c_.detach().abs().max()), 0, f"folded vs separate BatchNorm, feat{i}")
