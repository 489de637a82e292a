#!/usr/bin/env python3
"""Bytes per clock a CU takes in next to a full-rate MFMA stream (csrc/diag.hip, codeps_diag_feed_rate): per iteration each of
the block's four waves issues 96 v_mfma_f32_16x16x32_bf16 (one 128 x 128 x 32 six-product K-step) and the block brings kb KiB
from an L2-sized window into LDS -- through VGPRs + ds_write (how the conv kernels stage) or by LDS DMA.
    python tools/feed_rate.py [--window-mb 2] [--kb 0,16,32,48,64]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--window-mb", type=float, default=2.0)
    ap.add_argument("--kb", default="0,16,24,32,40,48,64")
    ap.add_argument("--iters", type=int, default=2000)
    args = ap.parse_args()
    from codeps_amd import _lib
    lib = _lib.diag_lib()
    dev = torch.device("cuda", 0)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    window = torch.randn(int(args.window_mb * (1 << 20)) // 4, device=dev)
    scratch = torch.empty(cus * 256, device=dev)
    ms = ctypes.c_float()
    print(f"{cus} CUs, window {args.window_mb} MiB, {args.iters} iterations of 96 MFMAs per wave")
    print("  kb  mode            us/iter   TFLOP/s (bf16)   GB/s per CU   TB/s chip")
    for kb in [int(k) for k in args.kb.split(",")]:
        for mode, name in ((0, "no loads"), (1, "VGPR + ds_write"), (2, "LDS DMA"), (5, "VGPR + frags"), (6, "LDS DMA + frags")):
            if kb == 0 and mode != 0:
                continue
            if kb != 0 and mode == 0:
                continue
            if (mode & 4) and kb < 8:
                continue
            _lib.check_diag(lib.codeps_diag_feed_rate(mode, kb, cus, args.iters, 3, _lib.ptr(window), window.numel() * 4, _lib.ptr(scratch),
                                                 ctypes.byref(ms), _lib.stream_ptr()), "codeps_diag_feed_rate")
            us = ms.value * 1e3 / args.iters
            flops = 96 * 4 * 2 * 16 * 16 * 32 * cus / (us * 1e-6)
            gbs = kb * 1024 / (us * 1e-6) / 1e9
            print(f"{kb:4d}  {name:15s} {us:8.3f}   {flops / 1e12:10.1f}   {gbs:10.1f}   {gbs * cus / 1e3:8.2f}", flush=True)


if __name__ == "__main__":
    main()
