#!/usr/bin/env python3
"""Matrix-pipe utilisation per kernel family from ONE rocprofv3 PMC pass (north_star: "MFMA utilisation against gfx950 peak").

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
        -d gpurun_out/pmc_mfma -o r -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --graph 0
    python tools/mfma_busy.py gpurun_out/pmc_mfma --steps 3 > profiles/rNN_mfma_busy.json

Units (MI355X_MICROARCH.md, cycle constants): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles in which a SIMD's matrix pipe is
busy, summed over SIMDs (32 per v_mfma_f32_32x32x16_bf16, 16 per 16x16x32 on the issuing SIMD); GRBM_GUI_ACTIVE counts
cycles the GPU is active, summed over the 8 XCDs.  mfma_busy = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs): the share
of (SIMD x cycle) slots of the dispatch in which a matrix pipe worked -- 1.0 would be the MFMA roofline at the clock the chip
held.  SQ_BUSY_CYCLES (per-SE busy cycles) is reported as delivered for cross-checks."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pmc_dir")
    ap.add_argument("--steps", type=int, required=True)
    ap.add_argument("--cus", type=int, default=256)
    a = ap.parse_args()
    disp = defaultdict(dict)
    for f in glob.glob(a.pmc_dir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            key = (int(r["Dispatch_Id"]), r["Kernel_Name"])
            disp[key][r["Counter_Name"]] = disp[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    fam = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for (_id, name), c in disp.items():
        if "GRBM_GUI_ACTIVE" not in c:
            continue
        e = fam[family(name)]
        e[0] += 1
        e[1] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        e[2] += c["GRBM_GUI_ACTIVE"]
        e[3] += c.get("SQ_BUSY_CYCLES", 0.0)
    rows = []
    for k, (n, mfma, gui, sq) in fam.items():
        slots = gui / 8.0 * a.cus * 4
        rows.append({"kernel": k, "launches_per_step": round(n / a.steps, 2), "mfma_busy": round(mfma / slots, 4) if slots else None,
                     "mfma_busy_cycles_per_launch": round(mfma / n), "gpu_active_cycles_per_launch": round(gui / 8.0 / n),
                     "sq_busy_cycles_per_launch": round(sq / n), "share_of_gpu_active_cycles": 0.0})
    tot = sum(r["gpu_active_cycles_per_launch"] * r["launches_per_step"] for r in rows) or 1.0
    for r in rows:
        r["share_of_gpu_active_cycles"] = round(r["gpu_active_cycles_per_launch"] * r["launches_per_step"] / tot, 4)
    rows.sort(key=lambda r: -r["share_of_gpu_active_cycles"])
    mf = [r for r in rows if r["mfma_busy"] and r["mfma_busy"] > 0.01]
    w = sum(r["share_of_gpu_active_cycles"] for r in mf) or 1.0
    json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace (one pass, eager launches)",
               "definition": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)",
               "steps_profiled": a.steps,
               "matrix_kernels_time_weighted_mfma_busy": round(sum(r["mfma_busy"] * r["share_of_gpu_active_cycles"] for r in mf) / w, 4),
               "kernels": rows}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
