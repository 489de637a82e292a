#!/usr/bin/env python3
"""Per-kernel averages of ONE rocprofv3 --pmc pass joined with its kernel trace, for the kernels whose name contains one of the given
substrings: where the wave cycles go (issuing / stalled / parked), vector-ALU share, LDS bank conflicts, occupancy.

    rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT \
              SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_k -o r -- \
              python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --graph 0
    python tools/pmc_kernels.py gpurun_out/pmc_k photo_fwd photo_bwd wgrad_halo64s > profiles/rNN_pmc_kernels.json

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_BUSY_CYCLES per SE;
GRBM_GUI_ACTIVE cycles summed over the 8 XCDs.  WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY
~ WAVE_CYCLES.  occupancy = wave-cycles / (4 x GPU-active cycles x 1024 SIMDs) = average waves resident per SIMD."""
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
    d, want = sys.argv[1], sys.argv[2:]
    cnt, names = defaultdict(dict), {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = int(r["Dispatch_Id"])
            names[k] = r["Kernel_Name"]
            cnt[k][r["Counter_Name"]] = cnt[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            for col in ("LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Workgroup_Size", "Grid_Size"):
                if col in r and r[col] != "":
                    cnt[k]["_" + col] = float(r[col])
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = int(r["Dispatch_Id"])
            if k in cnt:
                cnt[k]["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    fam = defaultdict(list)
    for k, c in cnt.items():
        n = family(names[k])
        if "_ns" in c and (not want or any(w in n for w in want)):
            fam[n].append(c)
    out = []
    for n, lst in sorted(fam.items()):
        avg = lambda key: sum(c.get(key, 0.0) for c in lst) / len(lst)           # noqa: E731
        wave, gui = avg("SQ_WAVE_CYCLES"), avg("GRBM_GUI_ACTIVE") / 8.0
        row = {"kernel": n, "launches": len(lst), "avg_us": round(avg("_ns") / 1e3, 1), "waves": round(avg("SQ_WAVES")),
               "workgroup_size": round(avg("_Workgroup_Size")), "lds_bytes_per_block": round(avg("_LDS_Block_Size")),
               "vgprs": round(avg("_VGPR_Count")), "agprs": round(avg("_Accum_VGPR_Count")),
               "gpu_active_cycles": round(gui), "effective_ghz": round(gui / max(avg("_ns"), 1.0), 3),
               "avg_waves_per_simd": round(wave * 4.0 / max(gui * 1024.0, 1.0), 2)}
        if wave:
            row.update(share_of_wave_cycles={"parked_at_waitcnt_or_barrier": round(avg("SQ_WAIT_ANY") / wave, 3),
                                             "issue_stalled": round(avg("SQ_WAIT_INST_ANY") / wave, 3),
                                             "issuing_vector_alu": round(avg("SQ_ACTIVE_INST_VALU") / wave, 3),
                                             "issuing_lds": round(avg("SQ_ACTIVE_INST_LDS") / wave, 3)})
        if "SQ_LDS_BANK_CONFLICT" in lst[0]:
            row["lds_bank_conflict_cycles"] = round(avg("SQ_LDS_BANK_CONFLICT"))
            row["lds_bank_conflict_cycles_over_gpu_active_x_cus"] = round(avg("SQ_LDS_BANK_CONFLICT") / max(gui * 256.0, 1.0), 4)
        out.append(row)
    json.dump({"source": "one rocprofv3 --pmc pass (SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                         "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE) + kernel trace, eager launches of the adapt step",
               "kernels": out}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
