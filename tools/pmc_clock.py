#!/usr/bin/env python3
"""Effective clock and matrix-pipe share per kernel from ONE rocprofv3 PMC pass joined with its kernel trace
(MI355X_MICROARCH.md "DVFS give-back": effective clock ~ GRBM_GUI_ACTIVE / 8 / kernel wall time; reads HIGH on dispatches shorter
than ~0.3 ms, so tools/clock_cycles.py's in-kernel stamps are the reference and this is the cross-check on the UNstamped product build).

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv \
        -d gpurun_out/pmc_clock -o r -- python3 tools/conv_bench.py --shapes l3_3x3,red_k1024,l2_expand,l1_expand,pose_l1 --bwd
    python tools/pmc_clock.py gpurun_out/pmc_clock > profiles/rNN_pmc_clock.json"""
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


def main():
    d = sys.argv[1]
    cnt = defaultdict(dict)
    names = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = int(r["Dispatch_Id"])
            names[k] = r["Kernel_Name"]
            cnt[k][r["Counter_Name"]] = cnt[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                cnt[k]["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = int(r["Dispatch_Id"])
            if k in cnt:
                cnt[k]["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    fam = defaultdict(list)
    for k, c in cnt.items():
        if "_ns" in c and "GRBM_GUI_ACTIVE" in c and c["_ns"] > 0:
            fam[family(names[k])].append(c)
    rows = []
    for name, lst in fam.items():
        n = len(lst)
        ns = sum(c["_ns"] for c in lst) / n
        gui = sum(c["GRBM_GUI_ACTIVE"] for c in lst) / n / 8.0
        mfma = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for c in lst) / n
        wave = sum(c.get("SQ_WAVE_CYCLES", 0.0) for c in lst) / n
        rows.append({"kernel": name, "launches": n, "avg_us": round(ns / 1e3, 1), "gpu_active_cycles": round(gui),
                     "effective_ghz_gui_active_over_wall": round(gui / ns, 3), "mfma_busy": round(mfma / (gui * 1024), 4) if gui else None,
                     "sq_wave_quad_cycles": round(wave), "sq_busy_cycles": round(sum(c.get("SQ_BUSY_CYCLES", 0.0) for c in lst) / n)})
    rows.sort(key=lambda r: -r["avg_us"] * r["launches"])
    json.dump({"source": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace (one pass, product build)",
               "caveat": "GRBM_GUI_ACTIVE / 8 / wall reads high on dispatches shorter than ~0.3 ms (MI355X_MICROARCH.md); in-kernel stamps: rNN_clock_cycles.json",
               "kernels": rows}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
