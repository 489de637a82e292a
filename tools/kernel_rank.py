#!/usr/bin/env python3
"""Per-step ranking of a rocprofv3 --kernel-trace --stats CSV: ms per step, launches per step, average us.
    python tools/kernel_rank.py <dir with *kernel_stats.csv> <steps covered> [top]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps = int(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
rows = list(csv.DictReader(open(f)))
print("total kernel ms/step", round(sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6, 3))
for r in rows[:top]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"\(.*", "", n)
    print("%6.3f ms %4d %8.1f us  %s" % (float(r["TotalDurationNs"]) / steps / 1e6, int(r["Calls"]) // steps, float(r["AverageNs"]) / 1e3, n[:90]))
