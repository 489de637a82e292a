#!/usr/bin/env python3
"""Concurrency of a step from a rocprofv3 kernel trace: wall time of a step, sum of kernel times, time with 0 / 1 / >= 2 kernels in
flight, and the kernels that run ALONE for the longest total time (the serial part of the step).
    python tools/overlap.py gpurun_out/trace_dir [groups=2]"""
import csv
import glob
import re
import sys
from collections import Counter


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z_0-9:]+(?:<[^(]*>)?)", n)
    return m.group(1) if m else n


def main():
    d = sys.argv[1]
    groups = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "adam_ema_kernel" in r["Kernel_Name"]]
    lo, hi = idx[-2 * groups - 1] + 1, idx[-groups - 1] + 1            # the second-to-last full step
    st = rows[lo:hi]
    ev = []
    for r in st:
        ev.append((int(r["Start_Timestamp"]), 1, r))
        ev.append((int(r["End_Timestamp"]), -1, r))
    ev.sort(key=lambda e: (e[0], e[1]))
    t0, t1 = ev[0][0], ev[-1][0]
    active, last = {}, t0
    hist = Counter()
    alone = Counter()
    for t, kind, r in ev:
        n = len(active)
        hist[min(n, 3)] += t - last
        if n == 1:
            alone[short(next(iter(active.values()))["Kernel_Name"])] += t - last
        last = t
        if kind == 1:
            active[id(r)] = r
        else:
            active.pop(id(r), None)
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in st)
    print(f"step wall {(t1 - t0) / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms, launches {len(st)}")
    for k in sorted(hist):
        print(f"  {k}{'+' if k == 3 else ''} kernels in flight: {hist[k] / 1e6:.3f} ms")
    print("running alone (ms):")
    for k, v in alone.most_common(25):
        print(f"  {v / 1e6:7.3f}  {k}")


if __name__ == "__main__":
    main()
