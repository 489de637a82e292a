#!/usr/bin/env python3
"""Kernel time per HIP queue (= stream) and step from a rocprofv3 --kernel-trace CSV: which stream carries how much of a step.
    python tools/queue_times.py <trace dir> [groups=2]"""
import csv
import glob
import re
import sys
from collections import Counter, defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z_0-9:]+)", n)
    return m.group(1) if m else n


def main():
    d = sys.argv[1]
    groups = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    steps, cur, adam = [], [], 0
    for r in rows:
        cur.append(r)
        if "adam_ema_kernel" in r["Kernel_Name"]:
            adam += 1
            if adam % groups == 0:
                steps.append(cur)
                cur = []
    st = steps[len(steps) // 2]
    per_q = defaultdict(float)
    names = defaultdict(Counter)
    for r in st:
        q = r.get("Queue_Id", "?")
        dt = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        per_q[q] += dt
        names[q][short(r["Kernel_Name"])] += dt
    print("step of", len(st), "launches, kernel time", round(sum(per_q.values()), 2), "ms")
    for q, t in sorted(per_q.items(), key=lambda kv: -kv[1]):
        top = ", ".join(f"{k} {v:.2f}" for k, v in names[q].most_common(6))
        print(f"  queue {q}: {t:6.2f} ms   {top}")


if __name__ == "__main__":
    main()
