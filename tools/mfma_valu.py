#!/usr/bin/env python3
"""Vector-ALU work next to a matrix-core stream on one SIMD (libcodeps_diag.so: codeps_diag_mfma_valu): time per MFMA of a
register-resident v_mfma_f32_32x32x16_bf16 loop (changing operands) with V v_fma_f32 per MFMA issued by the same wave or by a
second wave of the same SIMD.  32 cycles per MFMA = the nominal rate.
    python tools/mfma_valu.py [--json out.json]"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--clock-ghz", type=float, default=2.3, help="clock to express the times in cycles (rocm-smi reads 2.29 GHz under this load)")
    args = ap.parse_args()
    from codeps_amd import _lib
    lib = _lib.diag_lib()
    dev = torch.device("cuda:0")
    blocks, iters, reps = 256, 4000, 5
    scratch = torch.empty(blocks * 512, device=dev)
    ms = ctypes.c_float()
    out = {"blocks": blocks, "iters": iters, "clock_ghz_assumed": args.clock_ghz, "rows": []}
    for other in (0, 1):
        for v in (0, 2, 4, 8, 16):
            _lib.check_diag(lib.codeps_diag_mfma_valu(v, other, blocks, iters, reps, ctypes.c_void_p(scratch.data_ptr()), ctypes.byref(ms),
                                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "codeps_diag_mfma_valu")
            ns = ms.value * 1e6 / (iters * 16)
            row = {"valu_per_mfma": v, "issued_by": "second wave of the SIMD" if other else "the MFMA wave", "ns_per_mfma": round(ns, 2),
                   "cycles_per_mfma": round(ns * args.clock_ghz, 1), "tflops_bf16": round(2.0 * 32 * 32 * 16 * 16 * iters * 4 * blocks / (ms.value * 1e-3) / 1e12, 1)}
            out["rows"].append(row)
            print(row)
    if args.json:
        json.dump(out, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
