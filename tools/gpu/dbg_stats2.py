import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from codeps_amd import functional as F, _lib
dev = "cuda:0"
torch.manual_seed(0)
for (b, h, w, ci, co, k, st, off) in [(8, 16, 32, 64, 128, 1, 2, 0.0), (8, 16, 32, 64, 128, 1, 2, 3.0), (16, 16, 32, 64, 128, 1, 2, 3.0), (12, 64, 128, 64, 64, 3, 1, 0.0), (12, 64, 128, 64, 64, 3, 1, 3.0),
                                 (12, 128, 256, 64, 64, 3, 1, 2.0)]:
    x = torch.randn(b, h, w, ci, device=dev) + off
    wt = (torch.randn(co, ci, k, k, device=dev) / (ci * k * k) ** 0.5).abs().requires_grad_()      # positive weights + offset input: |mean| >> std
    gamma, beta = torch.ones(co, device=dev).requires_grad_(), torch.zeros(co, device=dev).requires_grad_()
    out = {}
    for fused in (False, True):
        F._EPI = fused
        rm, rv = torch.zeros(co, device=dev), torch.ones(co, device=dev)
        y = F.conv2d(x, wt, None, stride=st, pad=k // 2, bn_stats=True)
        z = F.batch_norm_train(y, gamma, beta, rm, rv, momentum=1.0)
        torch.cuda.synchronize()
        out[fused] = (rm.double(), rv.double(), getattr(y, "_codeps_stats", None) is not None)
    F._EPI = True
    yy = y.detach().double().reshape(-1, co)
    mu, var = yy.mean(0), yy.var(0, unbiased=True)
    f = lambda a, r: float(((a - r).abs() / r.abs().clamp_min(1e-30)).max())
    print((b, h, w, ci, co, k, st, off), "fused ran:", out[True][2], f"|mean|/std {float((mu.abs() / var.sqrt()).max()):.1f}",
          f"rel mean err sep {f(out[False][0], mu):.2e} fused {f(out[True][0], mu):.2e} | rel var err sep {f(out[False][1], var):.2e} fused {f(out[True][1], var):.2e}")
