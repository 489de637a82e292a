#!/usr/bin/env python3
"""The depthwise 5x5 kernels of the instance head (csrc/nn_ops.hip: dwconv5_kernel forward / data gradient, dwconv5_wgrad_kernel) against the
bytes they have to move: forward and data gradient read one tensor and write one, the weight gradient reads two.

    python tools/dwconv_rate.py [--json profiles/r06_dwconv_rate.json]

Shapes: models/instance_head.py:38-46 at B = 4 @ 512x1024 (the pre-training configurations) and B = 1 (the adaptation step's inference pass)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

DEV = "cuda:0"
SHAPES = [("dec_320", 4, 32, 64, 320), ("dec_288", 4, 64, 128, 288), ("dec_144", 4, 128, 256, 144), ("head_128", 4, 256, 512, 128),
          ("head_128_b1", 1, 256, 512, 128), ("dec_144_b1", 1, 128, 256, 144)]


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
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    from codeps_amd import _lib
    from codeps_amd import functional as F
    lib = _lib.lib()
    ptr, st = _lib.ptr, _lib.stream_ptr
    res = {}
    for name, b, h, w, c in SHAPES:
        g = torch.Generator(device=DEV).manual_seed(1)
        x = torch.randn(b, h, w, c, device=DEV, generator=g)
        dz = torch.randn(b, h, w, c, device=DEV, generator=g)
        wt = torch.randn(c, 1, 5, 5, device=DEV, generator=g)
        w25 = F._dw_pack(wt, c)
        y = torch.empty_like(x)
        d25 = torch.empty(25, c, device=DEV)
        ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)

        def fwd():
            _lib.check(lib.codeps_dwconv5x5(ptr(x), ptr(w25), None, None, 0, b, h, w, c, 0, ptr(y), st()), "dwconv5x5")

        def dgrad():
            _lib.check(lib.codeps_dwconv5x5(ptr(dz), ptr(w25), None, None, 0, b, h, w, c, 1, ptr(y), st()), "dwconv5x5")

        def wgrad():
            _lib.check(lib.codeps_dwconv5x5_wgrad(ptr(x), ptr(dz), b, h, w, c, ptr(d25), ptr(ws), ws.numel(), st()), "dwconv5x5_wgrad")
        tf, td, tw = timed(fwd, args.iters), timed(dgrad, args.iters), timed(wgrad, args.iters)
        by = x.numel() * 4.0 * 2
        res[name] = {"shape": f"B{b} {h}x{w} C{c}", "MB_per_pass": round(by / 1e6, 1), "fwd_us": round(tf, 1), "fwd_TBps": round(by / tf / 1e6, 2),
                     "dgrad_us": round(td, 1), "dgrad_TBps": round(by / td / 1e6, 2), "wgrad_us": round(tw, 1), "wgrad_TBps": round(by / tw / 1e6, 2)}
        print(name, json.dumps(res[name]), flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
