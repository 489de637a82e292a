#!/usr/bin/env python3
"""Where a replayed step is NOT filling the chip: from a rocprofv3 kernel trace of bench.py, per steady step, the wall time during which
exactly one kernel runs, split by that kernel's grid size (small grids = latency-bound launches nothing overlaps), and the idle gaps.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary
    python tools/critical_path.py gpurun_out/trace"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+)", name)
    return m.group(1) if m else name


def main():
    rows = []
    for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    steps, cur, adam = [], [], 0
    for r in rows:
        cur.append(r)
        if "adam_ema_kernel" in r["Kernel_Name"]:
            adam += 1
            if adam % 2 == 0:
                steps.append(cur)
                cur = []
    st = steps[len(steps) // 2]
    t0 = int(st[0]["Start_Timestamp"])
    ev = []
    for i, r in enumerate(st):
        ev.append((int(r["Start_Timestamp"]) - t0, 1, i))
        ev.append((int(r["End_Timestamp"]) - t0, -1, i))
    ev.sort()
    active = set()
    last = 0
    idle = 0
    alone = defaultdict(float)
    overlap = 0
    for t, d, i in ev:
        dt = t - last
        if dt > 0:
            if not active:
                idle += dt
            elif len(active) == 1:
                k = st[next(iter(active))]
                blocks = int(k["Grid_Size_X"]) * int(k["Grid_Size_Y"]) * int(k["Grid_Size_Z"]) // max(1, int(k["Workgroup_Size_X"]) * int(k["Workgroup_Size_Y"]) * int(k["Workgroup_Size_Z"]))
                alone[(short(k["Kernel_Name"]), "small" if blocks < 256 else "full")] += dt
            else:
                overlap += dt
        last = t
        if d == 1:
            active.add(i)
        else:
            active.discard(i)
    span = last
    small = sum(v for (k, s), v in alone.items() if s == "small")
    full = sum(v for (k, s), v in alone.items() if s == "full")
    out = {"span_ms": span / 1e6, "idle_ms": idle / 1e6, "two_or_more_kernels_ms": overlap / 1e6, "one_kernel_small_grid_ms": small / 1e6,
           "one_kernel_full_grid_ms": full / 1e6,
           "alone_small_grid_top": {k: round(v / 1e3, 1) for (k, s), v in sorted(alone.items(), key=lambda kv: -kv[1]) if s == "small"},
           "alone_full_grid_top": dict(list({k: round(v / 1e3, 1) for (k, s), v in sorted(alone.items(), key=lambda kv: -kv[1]) if s == "full"}.items())[:25])}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
