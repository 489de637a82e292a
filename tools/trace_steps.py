#!/usr/bin/env python3
"""Per-step view of a rocprofv3 kernel trace of bench.py: the trace is cut at every `adam_ema_kernel` pair (one optimizer step =
one launch per flat parameter group) and each step reports its kernel time, launch count, the weight-repack launches
(`pack_weights_kernel`, `split_pack_kernel`) and the library-internal launches that are not this repo's kernels.  A HIP-graph
step that is a REAL adaptation step shows the same repack count as an eagerly launched one.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline
    python tools/trace_steps.py gpurun_out/trace > profiles/rNN_step_trace.json
"""
import csv
import glob
import json
import re
import sys
from collections import Counter



def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+)", name)
    return m.group(1) if m else name


def main():
    d = sys.argv[1]
    groups = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 2
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
    out = []
    for i, st in enumerate(steps):
        names = Counter(short(r["Kernel_Name"]) for r in st)
        ours = lambda n: re.sub(r"^void ", "", n).startswith("(anonymous namespace)::")      # noqa: E731  this library's kernels
        foreign_rows = [r for r in st if not ours(r["Kernel_Name"])]
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in st)
        span = int(st[-1]["End_Timestamp"]) - int(st[0]["Start_Timestamp"])
        foreign = Counter(short(r["Kernel_Name"]) for r in foreign_rows)
        foreign_ms = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in foreign_rows) / 1e6
        out.append({"step": i, "launches": len(st), "kernel_busy_ms": round(busy / 1e6, 3), "span_ms": round(span / 1e6, 3),
                    "pack_weights_kernel": names.get("pack_weights_kernel", 0), "split_pack_kernel": names.get("split_pack_kernel", 0),
                    "library_launches": sum(foreign.values()), "library_busy_ms": round(foreign_ms, 3), "library_kernels": dict(sorted(foreign.items(), key=lambda kv: -kv[1])[:12])})
    res = {"steps": out}
    if "--hist" in sys.argv and len(steps) > 2:
        # every kernel of ONE steady step (the middle one): launches and busy time per kernel name
        st = steps[len(steps) // 2]
        hist = {}
        for r in st:
            e = hist.setdefault(short(r["Kernel_Name"]), [0, 0])
            e[0] += 1
            e[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        res["kernels_of_step"] = {"step": len(steps) // 2, "by_kernel": {k: {"launches": v[0], "busy_us": round(v[1] / 1e3, 1)}
                                                                          for k, v in sorted(hist.items(), key=lambda kv: -kv[1][1])}}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
