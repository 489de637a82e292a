#!/usr/bin/env python3
"""HBM rate of the two BatchNorm passes over a whole tensor that remain in a train-mode step (csrc/nn_ops.hip): the forward apply
(bn_apply_mg_kernel: y = relu((x - mean) invstd gamma + beta [+ residual])) and the backward apply (bn_bwd_apply_mg_kernel:
dx = gamma invstd (g - sum_g / n - xhat sum_gx / n)), beside a plain triad a = b + 1.5 c over tensors of the same size (codeps_diag_stream_triad).

    python tools/bn_apply_rate.py [--json profiles/r06_bn_apply_rate.json]

Shapes: the large BatchNorm layers of the pre-training step (ResNet-50 + heads, B = 4 @ 512x1024) and of the adaptation step's pose encoder
(B = 12 frame pairs in 6 statistic groups)."""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

DEV = "cuda:0"
#          name                 B   H    W    C    residual groups
SHAPES = [("r50_l1_bn3", 4, 128, 256, 256, True, 1),
          ("r50_l1_bn1", 4, 128, 256, 64, False, 1),
          ("r50_l2_bn3", 4, 64, 128, 512, True, 1),
          ("r50_l3_bn3", 4, 32, 64, 1024, True, 1),
          ("r50_l4_bn3", 4, 16, 32, 2048, True, 1),
          ("inst_head_dw", 4, 256, 512, 128, False, 1),
          ("inst_dec_144", 4, 128, 256, 144, False, 1),
          ("pose_l1", 12, 128, 256, 64, True, 6),
          ("pose_l2", 12, 64, 128, 128, True, 6)]


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    from codeps_amd import _lib
    lib, diag = _lib.lib(), _lib.diag_lib()
    ptr, st = _lib.ptr, _lib.stream_ptr
    res = {}
    for name, b, h, w, c, has_res, ng in SHAPES:
        g = torch.Generator(device=DEV).manual_seed(1)
        rows = b * h * w
        x = torch.randn(rows, c, device=DEV, generator=g)
        r = torch.randn(rows, c, device=DEV, generator=g) if has_res else None
        dy = torch.randn(rows, c, device=DEV, generator=g)
        y, dx = torch.empty_like(x), torch.empty_like(x)
        mean, var = torch.zeros(ng, c, device=DEV), torch.ones(ng, c, device=DEV)
        invstd = torch.ones(ng, c, device=DEV)
        gamma, beta = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
        sums = torch.zeros(ng, 2, c, device=DEV)
        dgam, dbet = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
        per = rows // ng
        arr = ctypes.c_longlong * ng
        row0, nrows = arr(*[i * per for i in range(ng)]), arr(*[per] * ng)
        inv_count = (ctypes.c_float * ng)(*[1.0 / per] * ng)
        ws = torch.empty(lib.codeps_bn_mg_workspace_bytes(ng, c), dtype=torch.uint8, device=DEV)
        iarr = ctypes.c_int * ng
        gb0, gb1 = iarr(*[i * (b // ng) for i in range(ng)]), iarr(*[(i + 1) * (b // ng) for i in range(ng)])
        one = ctypes.c_int * 1

        def fwd():
            _lib.check(lib.codeps_bn_fwd_mg(ptr(x), ng, row0, nrows, c, c, ptr(gamma), ptr(beta), ptr(r), 1, 1e-5, 0.1, None, None, ptr(mean), ptr(var),
                                            ptr(invstd), ptr(y), 0, 1, ptr(ws), ws.numel(), st()), "bn_fwd_mg")

        def bwd():
            _lib.check(lib.codeps_bn_bwd_tiles(None, 1, one(0), one(1), one(1), ng, row0, nrows, gb0, gb1, ptr(dy), ptr(x), c, c, ptr(mean), ptr(invstd),
                                               ptr(gamma), inv_count, ptr(sums), ptr(sums), ptr(dgam), ptr(dbet), 0, ptr(dx), 2, st()), "bn_bwd_tiles")
        ta, tb_, tc_ = torch.empty(rows * c, device=DEV), torch.ones(rows * c, device=DEV), torch.ones(rows * c, device=DEV)
        ms = ctypes.c_float()            # the same three streams (two tensors in, one out) as a plain a = b + 1.5 c, timed inside the library
        _lib.check_diag(diag.codeps_diag_stream_triad(ptr(ta), ptr(tb_), ptr(tc_), rows * c, 20, ctypes.byref(ms), st()), "stream_triad")
        tc = ms.value * 1e3
        tf, tb = timed(fwd, args.iters), timed(bwd, args.iters)
        by_f = rows * c * 4.0 * (3 if has_res else 2)
        by_b = rows * c * 4.0 * 3
        res[name] = {"shape": f"B{b} {h}x{w} C{c}" + (" +res" if has_res else "") + (f" {ng} groups" if ng > 1 else ""),
                     "fwd_apply_us": round(tf, 1), "fwd_TBps": round(by_f / tf / 1e6, 2), "bwd_apply_us": round(tb, 1), "bwd_TBps": round(by_b / tb / 1e6, 2),
                     "triad_same_size_us": round(tc, 1), "triad_TBps": round(by_b / tc / 1e6, 2)}
        print(name, json.dumps(res[name]), flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
