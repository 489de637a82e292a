#!/usr/bin/env python3
"""Per-layer table of the conv engine on the adapt step: every distinct (kind, shape) with its launch count per step, the
kernel that ran it, event-timed average duration, algorithmic TFLOP/s and the minimum HBM bytes (operands read once, result
written once) as GB/s -- the list the tile / kernel choices in conv.hip are tuned against.

    python tools/conv_layers.py [--steps 2] [--json out.json]

Branches are serialised (no side streams), as in bench.py's roofline leg."""
import argparse
import ctypes
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--mode", default="adapt")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    os.environ.setdefault("CODEPS_STREAMS", "0")
    import bench
    from codeps_amd import functional as F, _lib
    from codeps_amd.step import StepRunner
    device = torch.device("cuda", 0)
    enc = 101 if args.mode == "adapt" else 50
    net, cfg = bench.build(device, enc, 512, 1024, args.mode, None)
    data, _ = bench.make_data(args.mode, 512, 1024, device, 0, 8 if args.mode == "adapt" else 4, False, 1)
    runner = StepRunner(net, cfg, args.mode, 1)
    for _ in range(2):
        runner.step(data)
    torch.cuda.synchronize()

    # kernel names: ask the library right after each launch
    lib = _lib.lib()
    lib.codeps_conv2d_last_kernel.restype = ctypes.c_char_p
    names = []
    orig_end = F._prof_end

    def end(start, kind, flops, desc=None, source=0, tag=None, kernel=None):
        n0 = len(F._profile["events"])
        orig_end(start, kind, flops, desc, source, tag, kernel)
        if len(F._profile["events"]) > n0:
            k = lib.codeps_conv2d_last_kernel()
            names.append(kernel or (k.decode() if k else "?"))
    F._prof_end = end
    F._profile["events"].clear()
    F._profile["on"] = True
    for _ in range(args.steps):
        runner.step(data)
    torch.cuda.synchronize()
    F._profile["on"] = False
    rows = {}
    for (kind, flops, a, b, tag), kname in zip(F._profile["events"], names):
        if not kind.startswith("conv_"):
            continue
        r = rows.setdefault((kind, tag, kname), [0, 0.0, 0.0])
        r[0] += 1
        r[1] += flops
        r[2] += a.elapsed_time(b) * 1e-3
    out = []
    for (kind, tag, kname), (n, fl, s) in rows.items():
        m = re.match(r"B(\d+) (\d+)x(\d+) (\d+)\+(\d+)->(\d+) k(\d+) s(\d+)( up)?", tag or "")
        mb = None
        if m:
            b_, h, w, c0, c1, co, k, st, up = m.groups()
            b_, h, w, c0, c1, co, k, st = map(int, (b_, h, w, c0, c1, co, k, st))
            ho, wo = (h * (2 if up else 1)) // st, (w * (2 if up else 1)) // st
            mb = 4e-6 * (b_ * h * w * c0 + (b_ * ho * st * wo * st * c1) + b_ * ho * wo * co + (c0 + c1) * co * k * k)
        out.append({"kind": kind, "shape": tag, "kernel": kname, "per_step": n // args.steps, "avg_us": s / n * 1e6,
                    "ms_per_step": s / args.steps * 1e3, "tflops": fl / s / 1e12, "min_mb": mb,
                    "min_gbs": (mb * 1e-3 / (s / n)) if mb else None})
    out.sort(key=lambda r: -r["ms_per_step"])
    tot = sum(r["ms_per_step"] for r in out)
    print(f"{'kind':17s} {'shape':34s} {'kernel':32s} {'n':>3s} {'avg us':>8s} {'ms/step':>8s} {'TF/s':>7s} {'GB/s':>7s}")
    for r in out:
        print(f"{r['kind']:17s} {r['shape'] or '':34s} {r['kernel'][:32]:32s} {r['per_step']:3d} {r['avg_us']:8.1f} {r['ms_per_step']:8.3f} "
              f"{r['tflops']:7.1f} {(r['min_gbs'] or 0):7.0f}")
    print(f"total {tot:.3f} ms/step")
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
