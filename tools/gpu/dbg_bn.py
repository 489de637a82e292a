"""debug: BN test case under the epilogue switches; prints which launches gated and per-gradient errors vs the separate path"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from codeps_amd import functional as F, _lib
import test_round5_epilogue_gpu as T

log = []

def run_case(case):
    b, h, w, c_in, c, n_blocks, stride, ngroups = case
    per = b // ngroups
    groups = [(i * per, (i + 1) * per) for i in range(ngroups)]
    g = torch.Generator().manual_seed(c + h)
    x = T._nhwc(g, b, h, w, c_in)
    orig_h = F._handoff
    def spy(gate, e, part, dx):
        log.append((gate.kind, e.applied, _lib.lib().codeps_conv2d_last_kernel().decode(), tuple(dx.shape)))
        return orig_h(gate, e, part, dx)
    F._handoff = spy
    res = {}
    for mode in ("separate", "fused", "fused_nobn", "fused_nostats"):
        m = T._bn_modules(c_in, c, n_blocks, stride, seed=7)
        F._EPI = mode != "separate"
        F._EPI_PARTS["bn"] = mode != "fused_nobn"
        F._EPI_PARTS["stats"] = mode != "fused_nostats"
        log.clear()
        xi = x.clone().requires_grad_()
        y = T._run_blocks(F, m, xi, groups if ngroups > 1 else None, True)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)).to("cuda:0")
        grads = torch.autograd.grad(y, [xi] + list(m.parameters()), gy)
        torch.cuda.synchronize()
        res[mode] = [y.detach()] + [t.detach() for t in grads]
        print(mode, "handoffs:", log)
    for mode in ("fused", "fused_nobn", "fused_nostats"):
        errs = [float((a - r).abs().max() / (r.abs().max() + 1e-30)) for a, r in zip(res[mode], res["separate"])]
        print(mode, " ".join(f"{e:.1e}" for e in errs))

    F._handoff = orig_h

for a in sys.argv[1:]:
    print('=== case', a)
    run_case(tuple(int(v) for v in a.split(',')))
