#!/usr/bin/env python3
"""Time the chained 1x1 pair (csrc/chain1x1.hip) against the two layers as separate conv launches on ResNet-101 layer3's shape
(B 8, 32 x 64: 16384 pixels, 256 -> 1024 -> 256).  python tools/chain_bench.py [--reps 200]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    from codeps_amd import functional as F
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    out = []
    for b, h, w, n2 in ((8, 32, 64, 1024), (4, 32, 64, 1024), (8, 64, 128, 512)):
        y1 = torch.relu(torch.randn(b, h, w, 256, generator=g, device=dev))
        res = torch.relu(torch.randn(b, h, w, n2, generator=g, device=dev))
        w2 = torch.randn(n2, 256, 1, 1, generator=g, device=dev) * 0.09
        w3 = torch.randn(256, n2, 1, 1, generator=g, device=dev) * 0.04
        s2, t2, s3, t3 = (torch.rand(n, generator=g, device=dev) + 0.5 for n in (n2, n2, 256, 256))

        def chained():
            return F.chain_expand_reduce(y1, res, w2, s2, t2, w3, s3, t3)

        def separate():
            o = F.conv2d(y1, w2, scale=s2, shift=t2, residual=res, act="relu")
            return o, F.conv2d(o, w3, scale=s3, shift=t3, act="relu")

        row = {"shape": f"B{b} {h}x{w} 256->{n2}->256"}
        with torch.no_grad():
            for name, fn in (("separate_us", separate), ("chained_us", chained), ("separate_us_2", separate), ("chained_us_2", chained)):
                for _ in range(10):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                row[name] = round(e0.elapsed_time(e1) * 1e3 / a.reps, 2)
        flops = 2.0 * b * h * w * (n2 * 256 * 2)
        row["chained_tflops"] = round(flops / (min(row["chained_us"], row["chained_us_2"]) * 1e-6) / 1e12, 1)
        row["separate_tflops"] = round(flops / (min(row["separate_us"], row["separate_us_2"]) * 1e-6) / 1e12, 1)
        out.append(row)
        print(row, flush=True)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
