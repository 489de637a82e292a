"""Which train-mode BatchNorm layers of one pre-training step (configs[1]) still run the separate statistics / reduce passes."""
import collections
import sys
sys.path.insert(0, ".")
import torch
import bench
from codeps_amd import functional as F
from codeps_amd.step import StepRunner

dev = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "train"
net, cfg = bench.build(dev, 50 if mode == "train" else 101, 512, 1024, mode, None)
data, _ = bench.make_data(mode, 512, 1024, dev, 0, 4 if mode == "train" else 8, False, 1)
runner = StepRunner(net, cfg, mode, 1)
runner.step(data)
fw, bw = collections.Counter(), collections.Counter()
of, ob = F._BatchNormTrain.forward, F._BatchNormTrain.backward


def fwd(ctx, x, gamma, beta, residual, rm, rv, mom, eps, act, reduce_fn, groups, opts=None):
    y = of(ctx, x, gamma, beta, residual, rm, rv, mom, eps, act, reduce_fn, groups, opts)
    fw[(tuple(x.shape), act, residual is not None, "tiles" if (opts is not None and opts.stats is not None) else "PASS")] += 1
    ctx._dbg = (tuple(x.shape), act, residual is not None)
    return y


def bwd(ctx, dy):
    gate = ctx.out_gate
    how = "no-gate" if gate is None else ("gated" if gate.handoff is not None else "GATE-UNUSED")
    bw[ctx._dbg + (how,)] += 1
    return ob(ctx, dy)


F._BatchNormTrain.forward = staticmethod(fwd)
F._BatchNormTrain.backward = staticmethod(bwd)
runner.step(data)
torch.cuda.synchronize()
print("forward statistics:")
for k, v in sorted(fw.items(), key=lambda kv: str(kv[0])):
    print("  ", k, v)
print("backward:")
for k, v in sorted(bw.items(), key=lambda kv: str(kv[0])):
    print("  ", k, v)
