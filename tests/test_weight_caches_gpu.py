"""Caches of DERIVED weights must follow the fused optimizer, which rewrites `p.data` through raw pointers (torch's tensor version
does not move): validation of a TRAINABLE encoder (`net.eval()` under `no_grad` between optimizer steps,
scripts/train_codeps.py:290-330) reaches the frozen-encoder fast paths -- the K = 224 stem (F.stem3_conv2d), the chained 1x1 pairs
of layer3 (F.chain_expand_reduce) and the folded BatchNorms (models/_blocks.folded_bn) -- and must see the weights and the running
statistics of the LAST step, not the ones packed at first use."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def _features(net, rgb, fast):
    from codeps_amd import functional as F
    old = F._CHAIN, F._STEM3
    F._CHAIN, F._STEM3 = fast, fast
    try:
        net.eval()
        with torch.no_grad():
            return [f.clone() for f in net.backbone_po_depth(rgb)]
    finally:
        F._CHAIN, F._STEM3 = old
        net.train()


def _rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("layers", [50, 101])
def test_validation_between_optimizer_steps_sees_the_current_weights(layers):
    sys.path.insert(0, ROOT)
    import bench
    from codeps_amd import functional as F
    from codeps_amd.step import StepRunner
    dev = torch.device(DEV)
    h, w, b = 128, 256, 2                                   # layer3: 2 x 8 x 16 = 256 pixels (the chain kernel takes multiples of 128)
    net, cfg = bench.build(dev, layers, h, w, "train")
    net.train()
    data, _ = bench.make_data("train", h, w, dev, 0, b, False, 1)
    rgb = data["rgb"][0]
    runner = StepRunner(net, cfg, "train", 1)
    runner.lr = 3e-3                                         # a visible update in two steps
    # first use: every cache of derived weights is filled here
    fast0, slow0 = _features(net, rgb, True), _features(net, rgb, False)
    for a, r in zip(fast0, slow0):
        assert _rel(a, r) < 1e-5
    if F._CHAIN and F._STEM3:
        assert F._chain_cache and F._stem3_cache, "the fast paths did not run: this test would not see a stale cache"
    for _ in range(2):
        runner.step(data)
    fast1, slow1 = _features(net, rgb, True), _features(net, rgb, False)
    torch.cuda.synchronize()
    for lvl, (a, r, a0) in enumerate(zip(fast1, slow1, fast0)):
        moved = _rel(r, slow0[lvl])
        assert moved > 1e-3, f"level {lvl}: the layer-by-layer path did not move ({moved:.2e}); the test is vacuous"
        assert _rel(a, r) < 1e-5 + 1e-3 * moved, (f"level {lvl}: fast paths {_rel(a, r):.2e} away from the layer-by-layer path after two "
                                                   f"optimizer steps that moved the features by {moved:.2e} (stale packs?)")
        assert _rel(a, a0) > 0.5 * moved


def test_folded_batchnorm_follows_train_mode_running_statistics():
    """A BatchNorm that ran in train mode (running statistics written by the statistics kernel through raw pointers) and is then
    evaluated folded: the fold must be recomputed."""
    from codeps_amd.models._blocks import conv_bn, folded_bn
    from torch import nn
    dev = torch.device(DEV)
    torch.manual_seed(3)
    conv, bn = nn.Conv2d(16, 32, 3, padding=1, bias=False).to(dev), nn.BatchNorm2d(32).to(dev)
    for p in list(conv.parameters()) + list(bn.parameters()):
        p.requires_grad_(False)
    x = torch.randn(2, 8, 8, 16, device=dev) * 3 + 1
    bn.eval()
    s0, t0 = [v.clone() for v in folded_bn(bn)]
    bn.train()
    with torch.no_grad():
        conv_bn(x, conv, bn, "relu")
    bn.eval()
    s1, t1 = folded_bn(bn)
    torch.cuda.synchronize()
    want_s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    want_t = bn.bias - bn.running_mean * want_s
    assert not torch.allclose(s0, s1) and not torch.allclose(t0, t1), "the fold did not follow the running statistics"
    assert torch.allclose(s1, want_s, rtol=1e-6, atol=1e-7) and torch.allclose(t1, want_t, rtol=1e-5, atol=1e-6)
