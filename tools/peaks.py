#!/usr/bin/env python3
"""Measured rooflines of this box: the MFMA micro-kernel and the stream triad of csrc/diag.hip.

    python tools/peaks.py [--json profiles/rNN_peaks.json]

Prints, for each MFMA flavour the conv engine uses, the register-resident issue rate on zero operands (what vendor peaks are
quoted on) and on operand fragments that change between consecutive MFMAs (what a contraction of real data sees under the
chip's power limit), and the HBM triad rate."""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def measure(sustain_ms=200.0):
    from codeps_amd import _lib
    lib = _lib.diag_lib()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    blocks = cus * 4                                      # 4 blocks x 4 waves per CU: 4 waves per SIMD
    scratch = torch.empty(blocks * 256, device=dev)
    out = {"cus": cus}
    names = {0: "mfma_f32_32x32x16_bf16", 1: "mfma_f32_32x32x2_f32", 2: "mfma_f32_16x16x32_bf16"}
    for mode, name in names.items():
        for rnd in (0, 1):
            flops, ms = ctypes.c_double(), ctypes.c_float()
            iters = 2000 if mode != 1 else 1000
            # short probe, then enough launches to hold the load for `sustain_ms` (clocks settle under the power limit)
            _lib.check_diag(lib.codeps_diag_mfma_rate(mode, rnd, blocks, iters, 2, _lib.ptr(scratch), ctypes.byref(flops), ctypes.byref(ms),
                                                 _lib.stream_ptr()), "codeps_diag_mfma_rate")
            reps = max(3, int(sustain_ms / max(ms.value, 1e-3)))
            _lib.check_diag(lib.codeps_diag_mfma_rate(mode, rnd, blocks, iters, reps, _lib.ptr(scratch), ctypes.byref(flops), ctypes.byref(ms),
                                                 _lib.stream_ptr()), "codeps_diag_mfma_rate")
            out[f"{name}_{'random' if rnd else 'zeros'}_tflops"] = round(flops.value / (ms.value * 1e-3) / 1e12, 1)
    n = 1 << 28                                           # 3 x 1 GiB: beyond the 256 MiB Infinity Cache
    a, b, c = (torch.empty(n, device=dev) for _ in range(3))
    b.fill_(1.0)
    c.fill_(2.0)
    ms = ctypes.c_float()
    _lib.check_diag(lib.codeps_diag_stream_triad(_lib.ptr(a), _lib.ptr(b), _lib.ptr(c), n, 20, ctypes.byref(ms), _lib.stream_ptr()),
               "codeps_diag_stream_triad")
    assert float(a[12345]) == 4.0
    out["hbm_triad_gbs"] = round(12.0 * n / (ms.value * 1e-3) / 1e9, 1)
    _lib.check_diag(lib.codeps_diag_stream_copy(_lib.ptr(a), _lib.ptr(b), n, 20, ctypes.byref(ms), _lib.stream_ptr()), "codeps_diag_stream_copy")
    assert float(a[12345]) == 1.0
    out["hbm_copy_gbs"] = round(8.0 * n / (ms.value * 1e-3) / 1e9, 1)       # the roof MI355X_MICROARCH.md measures (float4 copy: 6.29 TB/s)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    res = measure()
    for k, v in res.items():
        print(f"{k:44s} {v}")
    if args.json:
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
