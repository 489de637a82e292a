import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from codeps_amd import functional as F, _lib
dev = "cuda:0"
torch.manual_seed(0)
for (b, h, w, ci, co, k, st, groups) in [
        (8, 16, 32, 64, 128, 1, 2, [(0, 3), (3, 6), (6, 7), (7, 8)]),
        (8, 16, 32, 64, 128, 1, 2, [(0, 2), (2, 4), (4, 5), (5, 6), (6, 7), (7, 8)]),
        (8, 16, 32, 64, 128, 1, 2, [(0, 1), (1, 2), (2, 4), (4, 6), (6, 7), (7, 8)]),
        (16, 16, 32, 64, 128, 1, 2, [(0, 2), (2, 4), (4, 8), (8, 12), (12, 14), (14, 16)]),
        (12, 64, 128, 64, 64, 3, 1, [(0, 2), (2, 4), (4, 6), (6, 8), (8, 10), (10, 12)])]:
    x = torch.randn(b, h, w, ci, device=dev)
    wt = (torch.randn(co, ci, k, k, device=dev) / (ci * k * k) ** 0.5).requires_grad_()
    gamma, beta = (torch.rand(co, device=dev) + 0.5).requires_grad_(), torch.randn(co, device=dev).requires_grad_()
    res = {}
    for fused in (False, True):
        F._EPI = fused
        rm, rv = torch.zeros(co, device=dev), torch.ones(co, device=dev)
        y = F.conv2d(x, wt, None, stride=st, pad=k // 2, bn_stats=True)
        st_rec = getattr(y, "_codeps_stats", None)
        z = F.batch_norm_train(y, gamma, beta, rm, rv, groups=groups, act="relu")
        torch.cuda.synchronize()
        res[fused] = (y.detach().clone(), z.detach().clone(), rm.clone(), rv.clone(), None if st_rec is None else {k2: v for k2, v in st_rec.items() if k2 != "part"},
                      _lib.lib().codeps_conv2d_last_kernel().decode())
    F._EPI = True
    a, c = res[False], res[True]
    print((b, h, w, ci, co, k, st), "groups", groups, "stats rec", c[4], c[5])
    print("   conv equal", torch.equal(a[0], c[0]), " bn out max diff", float((a[1] - c[1]).abs().max()), " rm diff", float((a[2] - c[2]).abs().max()),
          " rv diff", float((a[3] - c[3]).abs().max()))
    # per group reference in float64
    yy = a[0].double()
    for gi, (b0, b1) in enumerate(groups):
        mu = yy[b0:b1].mean((0, 1, 2)); var = yy[b0:b1].reshape(-1, co).var(0, unbiased=False)
        ref = torch.relu((yy[b0:b1] - mu) / torch.sqrt(var + 1e-5) * gamma.detach().double() + beta.detach().double())
        print(f"   group {gi}: separate err {float((a[1][b0:b1].double() - ref).abs().max()):.2e}  fused err {float((c[1][b0:b1].double() - ref).abs().max()):.2e}")
