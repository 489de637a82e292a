#!/usr/bin/env python3
"""The GUDA tap-plane gather of the semantic head (csrc/nn_ops.hip: guda_gather_fwd_tiled_kernel, guda_gather_bwd_{x,y}_kernel;
models/semantic_head.py:64-72) against the bytes it has to move: base + out at full resolution (16 channels) and the 144-channel tap planes of
the three coarse branches (1/8, 1/4, 1/2).

    python tools/guda_rate.py [--json profiles/r06_guda_rate.json] [--batch 3]"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

DEV = "cuda:0"


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
    ap.add_argument("--batch", type=int, default=3)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    from codeps_amd import _lib
    lib = _lib.lib()
    ptr, st = _lib.ptr, _lib.stream_ptr
    b, h, w = args.batch, 512, 1024
    g = torch.Generator(device=DEV).manual_seed(1)
    base = torch.randn(b, h, w, 16, device=DEV, generator=g)
    geoms = [(64, 128), (128, 256), (256, 512)]
    ys = [torch.randn(b, bh, bw, 144, device=DEV, generator=g) for bh, bw in geoms]
    out = torch.empty_like(base)
    n = len(ys)
    yp = (ctypes.c_void_p * n)(*[y.data_ptr() for y in ys])
    bhs, bws = (ctypes.c_int * n)(*[x[0] for x in geoms]), (ctypes.c_int * n)(*[x[1] for x in geoms])

    def fwd():
        _lib.check(lib.codeps_guda_gather_fwd(ptr(base), b, h, w, n, yp, bhs, bws, ptr(out), st()), "guda_gather_fwd")
    res = {}
    tf = timed(fwd, args.iters)
    by = (2 * base.numel() + sum(y.numel() for y in ys)) * 4.0
    res["fwd"] = {"shape": f"B{b} {h}x{w}x16 + 144-channel planes at 1/8, 1/4, 1/2", "MB": round(by / 1e6, 1), "us": round(tf, 1), "TBps": round(by / tf / 1e6, 2)}
    print("fwd", json.dumps(res["fwd"]), flush=True)
    dout = torch.randn_like(base)
    for (bh, bw), y in zip(geoms, ys):
        dy = torch.empty_like(y)
        ws = torch.empty(lib.codeps_guda_gather_bwd_workspace_bytes(b, h, bw), dtype=torch.uint8, device=DEV)

        def bwd():
            _lib.check(lib.codeps_guda_gather_bwd(ptr(dout), b, h, w, bh, bw, ptr(dy), ptr(ws), ws.numel(), st()), "guda_gather_bwd")
        tb = timed(bwd, args.iters)
        byb = (dout.numel() + dy.numel()) * 4.0
        res[f"bwd_{bh}x{bw}"] = {"MB_in_out": round(byb / 1e6, 1), "workspace_MB": round(ws.numel() / 1e6, 1), "us": round(tb, 1), "TBps": round(byb / tb / 1e6, 2)}
        print(f"bwd_{bh}x{bw}", json.dumps(res[f"bwd_{bh}x{bw}"]), flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
