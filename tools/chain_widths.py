#!/usr/bin/env python3
"""The chained 1x1 pair against the same two layers as separate launches, at the three bottleneck stages of the frozen ResNet-101 trunk
(the chain at layer1 / layer2 widths).

    python tools/chain_widths.py [--json profiles/r06_chain_widths.json]

Per shape: HIP-event time per pair, chained and separate, alternating on one box; the bytes a pair moves either way (fp32 maps in HBM:
chained reads y1 + residual and writes out + y0'; separate writes `out` and reads it back), and the rate those bytes move at."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

DEV = "cuda:0"
SHAPES = {"layer1": (8, 128, 256, 64, 256), "layer2": (8, 64, 128, 128, 512), "layer3": (8, 32, 64, 256, 1024)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    from codeps_amd import functional as F
    res = {}
    for name, (b, h, w, k, n2) in SHAPES.items():
        g = torch.Generator().manual_seed(1)
        y1 = torch.randn(b, h, w, k, generator=g).relu().to(DEV)
        idn = torch.randn(b, h, w, n2, generator=g).relu().to(DEV)
        w2 = (torch.randn(n2, k, 1, 1, generator=g) * (2.0 / k) ** 0.5).to(DEV)
        w3 = (torch.randn(k, n2, 1, 1, generator=g) * (2.0 / n2) ** 0.5).to(DEV)
        s2, t2 = torch.ones(n2, device=DEV), torch.zeros(n2, device=DEV)
        s3, t3 = torch.ones(k, device=DEV), torch.zeros(k, device=DEV)

        def chained():
            return F.chain_expand_reduce(y1, idn, w2, s2, t2, w3, s3, t3)

        def separate():
            o = F.conv2d(y1, w2, scale=s2, shift=t2, residual=idn, act="relu")
            return o, F.conv2d(o, w3, scale=s3, shift=t3, act="relu")
        times = {"chained": [], "separate": []}
        with torch.no_grad():
            for _rep in range(3):
                for label, fn in (("chained", chained), ("separate", separate)):
                    fn()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.iters):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    times[label].append(e0.elapsed_time(e1) * 1e3 / args.iters)
        rows = b * h * w
        by_c = rows * 4.0 * (k + n2 + n2 + k)
        by_s = rows * 4.0 * (k + n2 + n2 + n2 + k)
        flops = 4.0 * rows * k * n2
        tc, ts = min(times["chained"]), min(times["separate"])
        res[name] = {"shape": f"B{b} {h}x{w} {k}->{n2}->{k}", "chained_us": round(tc, 1), "separate_us": round(ts, 1), "speedup": round(ts / tc, 3),
                     "chained_MB": round(by_c / 1e6, 1), "separate_MB": round(by_s / 1e6, 1),
                     "chained_TBps": round(by_c / tc / 1e6, 2), "separate_TBps": round(by_s / ts / 1e6, 2),
                     "chained_TFLOPs": round(flops / tc / 1e6, 1)}
        print(name, json.dumps(res[name]), flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
