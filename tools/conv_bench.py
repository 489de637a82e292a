#!/usr/bin/env python3
"""Micro-benchmark of single conv layers (forward and optionally backward) through the C ABI: HIP-event time per launch,
algorithmic TFLOP/s and the kernel the dispatcher picked.  Used to A/B kernel variants (CODEPS_* switches, or another build of
the library through CODEPS_HIP_LIB) on the shapes that dominate the adapt step.

    python tools/conv_bench.py [--shapes l3_3x3,l3_reduce,...] [--iters 30] [--bwd]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

SHAPES = {               # name: (B, H, W, Cin, Cout, k, stride)
    "l3_3x3": (8, 32, 64, 256, 256, 3, 1),
    "l3_reduce": (8, 32, 64, 1024, 256, 1, 1),
    "l3_expand": (8, 32, 64, 256, 1024, 1, 1),
    "l2_3x3": (8, 64, 128, 128, 128, 3, 1),
    "l2_expand": (8, 64, 128, 128, 512, 1, 1),
    "l1_3x3": (8, 128, 256, 64, 64, 3, 1),
    "l1_expand": (8, 128, 256, 64, 256, 1, 1),
    "l4_3x3": (8, 16, 32, 512, 512, 3, 1),
    "pose_l1": (12, 128, 256, 64, 64, 3, 1),
    "pose_l4": (12, 16, 32, 512, 512, 3, 1),
    "dec_2048": (6, 16, 32, 2048, 256, 3, 1),
}
# K sweeps of the 1x1 expand shape (fixed M x N): time = intercept (prologue, epilogue, tail) + slope x K
SHAPES.update({f"exp_k{k}": (8, 32, 64, k, 1024, 1, 1) for k in (64, 128, 256, 512, 1024, 2048)})
SHAPES.update({f"red_k{k}": (8, 32, 64, k, 256, 1, 1) for k in (64, 128, 256, 512, 1024, 2048)})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="l3_3x3,l3_reduce,l3_expand,l2_3x3,l1_3x3,l4_3x3")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--bwd", action="store_true")
    args = ap.parse_args()
    from codeps_amd import functional as F, _lib
    lib = _lib.lib()
    lib.codeps_conv2d_last_kernel.restype = ctypes.c_char_p
    dev = torch.device("cuda", 0)
    for name in args.shapes.split(","):
        b, h, w, ci, co, k, st = SHAPES[name]
        g = torch.Generator(device="cpu").manual_seed(1)
        x = torch.randn(b, h, w, ci, generator=g).to(dev)
        wt = (torch.randn(co, ci, k, k, generator=g) * 0.05).to(dev)
        if args.bwd:
            x.requires_grad_(True)
            wt.requires_grad_(True)
        y = F.conv2d(x, wt, stride=st, pad=k // 2)
        kname = lib.codeps_conv2d_last_kernel().decode()
        gy = torch.randn_like(y) if args.bwd else None
        if args.bwd:
            y.backward(gy)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            y = F.conv2d(x, wt, stride=st, pad=k // 2)
            if args.bwd:
                y.backward(gy)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        flops = 2.0 * b * (h // st) * (w // st) * ci * co * k * k * (3 if args.bwd else 1)
        print(f"{name:10s} B{b} {h}x{w} {ci}->{co} k{k} s{st}  {kname:32s} {us:8.1f} us  {flops / us / 1e6:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
