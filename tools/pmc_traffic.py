#!/usr/bin/env python3
"""Summarise HBM traffic per kernel family from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --graph 0
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ... (same)
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write --steps 3 > profiles/rNN_pmc_traffic.json

Corrections (MI355X_MICROARCH.md, "HBM"): the counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of
wide coalesced reads, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  The two passes run the same
deterministic program, so dispatches are matched by (Dispatch_Id, Kernel_Name).
"""
import argparse
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def family(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(?:<[^(]*>)?)", name)
    return m.group(1) if m else name


def load(directory, counter):
    out = {}
    for f in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                key = (int(r["Dispatch_Id"]), r["Kernel_Name"])
                out[key] = out.get(key, 0.0) + float(r["Counter_Value"])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--steps", type=int, required=True, help="bench steps + warmup in the profiled run")
    a = ap.parse_args()
    fetch, write = load(a.fetch_dir, "FETCH_SIZE"), load(a.write_dir, "WRITE_SIZE")
    fam = defaultdict(lambda: [0, 0.0, 0.0])
    unmatched = 0
    for key, kib in fetch.items():
        if key not in write:
            unmatched += 1
            continue
        e = fam[family(key[1])]
        e[0] += 1
        e[1] += 2.0 * kib * 1024.0
        e[2] += write[key] * 1024.0
    rows = []
    for k, (n, rd, wr) in sorted(fam.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        rows.append({"kernel": k, "launches_per_step": round(n / a.steps, 2), "read_mb_per_launch": round(rd / n / 1e6, 3),
                     "write_mb_per_launch": round(wr / n / 1e6, 3), "hbm_gb_per_step": round((rd + wr) / a.steps / 1e9, 4)})
    conv = [r for r in rows if r["kernel"].startswith(("conv_igemm", "conv_halo3", "conv_wgrad", "wgrad_small", "wgrad_halo64", "wgrad_stem74"))]
    n_conv = sum(r["launches_per_step"] for r in conv)
    gb_conv = sum(r["hbm_gb_per_step"] for r in conv)
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); FETCH_SIZE x2 (gfx950), KiB -> bytes",
               "steps_profiled": a.steps, "unmatched_dispatches": unmatched,
               "conv": {"launches_per_step": round(n_conv, 1), "hbm_gb_per_step": round(gb_conv, 3),
                        "hbm_mb_per_launch": round(gb_conv * 1e3 / max(n_conv, 1), 2)},
               "total_hbm_gb_per_step": round(sum(r["hbm_gb_per_step"] for r in rows), 3), "kernels": rows}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
