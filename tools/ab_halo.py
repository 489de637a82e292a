#!/usr/bin/env python3
"""A/B of the two forms of the halo-tile ring kernel in ONE process (codeps_conv2d_set_halo_ws), alternating, on the step's layers:
    python tools/ab_halo.py [--rounds 6] [--iters 100]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

SHAPES = {"l3_3x3": (8, 32, 64, 256, 256), "l2_3x3": (8, 64, 128, 128, 128), "l4_3x3": (8, 16, 32, 512, 512), "pose_l4": (12, 16, 32, 512, 512),
          "pose_l3": (12, 32, 64, 256, 256), "pose_l2": (12, 64, 128, 128, 128), "dec_2048": (6, 16, 32, 2048, 256)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--bwd", action="store_true")
    a = ap.parse_args()
    from codeps_amd import functional as F, _lib
    lib = _lib.lib()
    lib.codeps_conv2d_last_kernel.restype = ctypes.c_char_p
    dev = torch.device("cuda:0")
    for name, (b, h, w, ci, co) in SHAPES.items():
        g = torch.Generator().manual_seed(1)
        x = torch.randn(b, h, w, ci, generator=g).to(dev)
        wt = (torch.randn(co, ci, 3, 3, generator=g) * 0.05).to(dev)
        if a.bwd:
            x.requires_grad_(True)
        gy = torch.randn(b, h, w, co, generator=g).to(dev)
        t = {0: [], 1: []}
        kn = {}
        for r in range(a.rounds):
            for ws in (1, 0):
                lib.codeps_conv2d_set_halo_ws(ws)
                for _ in range(20):
                    y = F.conv2d(x, wt, pad=1)
                    if a.bwd:
                        y.backward(gy)
                kn[ws] = lib.codeps_conv2d_last_kernel().decode()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    y = F.conv2d(x, wt, pad=1)
                    if a.bwd:
                        y.backward(gy)
                e1.record()
                torch.cuda.synchronize()
                t[ws].append(e0.elapsed_time(e1) * 1e3 / a.iters)
        m1, m0 = sorted(t[1])[len(t[1]) // 2], sorted(t[0])[len(t[0]) // 2]
        print(f"{name:9s} B{b} {h}x{w} {ci}->{co}  ws=1 {kn[1]:28s} {m1:7.1f} us   ws=0 {kn[0]:28s} {m0:7.1f} us   ratio {m1 / m0:.3f}", flush=True)
    lib.codeps_conv2d_set_halo_ws(1)


if __name__ == "__main__":
    main()
