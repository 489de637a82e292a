#!/usr/bin/env python3
"""Cycles and clocks measured IN the kernels (MI355X_MICROARCH.md "DVFS give-back" item 6): s_memtime counts shader cycles and
s_memrealtime a constant 100 MHz, so a pair of stamps gives the cycles a loop took AND the clock the chip held while it ran -- no
frequency is assumed anywhere (tools/mfma_valu.py of round 3 converted wall time with rocm-smi's 2.3 GHz).

  python tools/clock_cycles.py --bare    [--json out.json]     libcodeps_diag.so: bare MFMA loops (zeros / random operands) and the
                                                               V x v_fma_f32-per-MFMA table, both issue modes
  CODEPS_HIP_LIB=codeps_amd/variants/libcodeps_hip_stamps.so \\
  python tools/clock_cycles.py --kernels [--json out.json]     the diagnostic build of the product library (tools/build_variant.sh stamps
                                                               -DCODEPS_CLOCK_STAMPS): conv_halo3r, conv_igemm<split3>, wgrad_halo64s,
                                                               chain_expand_reduce on the layers of the adapt step

Every measurement runs >= `--warm` seconds of back-to-back launches on random data first; the stamps read are those of the last launch.
The stamped build's run time is not a benchmark (its fences forbid overlaps the product kernels have): read its clocks and shares."""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FLOP_PER_CYCLE_SIMD = 1024.0       # bf16 MFMA: 2 * 32 * 32 * 16 / 32 cycles = 2 * 16 * 16 * 32 / 16 cycles (MI355X_MICROARCH.md cycle constants)
N_SIMD = 1024


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def bare(warm_s):
    from codeps_amd import _lib
    lib = _lib.diag_lib()
    dev = torch.device("cuda:0")
    blocks, iters, reps = 256, 4000, 10              # one block per CU, one wave per SIMD
    scratch = torch.empty(blocks * 512, device=dev)
    stamps = torch.zeros(blocks * 4 * 4, dtype=torch.int64, device=dev)
    ms = ctypes.c_float()
    rows = []

    def run(call, flop_per_mfma, nominal):
        t0 = time.time()
        while time.time() - t0 < warm_s:             # >= warm_s of back-to-back launches before the stamps that count
            call()
            torch.cuda.synchronize()
        stamps.zero_()
        call()
        torch.cuda.synchronize()
        s = stamps.cpu().numpy().reshape(-1, 4).astype(np.float64)
        dt, dr = s[:, 2] - s[:, 0], s[:, 3] - s[:, 1]
        ok = dr > 0
        n_mfma = 16 * iters
        cyc = float(np.median(dt[ok] / n_mfma))
        ghz = float(np.median(dt[ok] / dr[ok]) * 0.1)
        tflops = flop_per_mfma * n_mfma * 4 * blocks / (ms.value * 1e-3) / 1e12
        return {"cycles_per_mfma": round(cyc, 2), "nominal_cycles": nominal, "in_kernel_ghz": round(ghz, 3), "tflops": round(tflops, 1),
                "ghz_min_max_over_waves": [round(float((dt[ok] / dr[ok]).min() * 0.1), 3), round(float((dt[ok] / dr[ok]).max() * 0.1), 3)],
                "check_tflops_from_cycles_and_clock": round(flop_per_mfma / cyc * ghz * 1e9 * N_SIMD / 1e12, 1)}

    names = {0: ("v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16, 32), 1: ("v_mfma_f32_32x32x2_f32", 2.0 * 32 * 32 * 2, 64),
             2: ("v_mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32, 16)}
    for mode in (0, 2, 1):
        for rnd in (0, 1):
            nm, fl, nominal = names[mode]
            r = run(lambda: _lib.check_diag(lib.codeps_diag_mfma_rate_clock(mode, rnd, blocks, iters, reps, ctypes.c_void_p(scratch.data_ptr()),
                                                                            ctypes.c_void_p(stamps.data_ptr()), ctypes.byref(ms), _stream()),
                                            "codeps_diag_mfma_rate_clock"), fl, nominal)
            r.update(loop="bare " + nm, operands="random (change between consecutive MFMAs)" if rnd else "zeros")
            rows.append(r)
            print(r, flush=True)
    for other in (0, 1):
        for v in (0, 2, 4, 8, 16):
            r = run(lambda: _lib.check_diag(lib.codeps_diag_mfma_valu_clock(v, other, blocks, iters, reps, ctypes.c_void_p(scratch.data_ptr()),
                                                                            ctypes.c_void_p(stamps.data_ptr()), ctypes.byref(ms), _stream()),
                                            "codeps_diag_mfma_valu_clock"), 2.0 * 32 * 32 * 16, 32)
            r.update(loop="v_mfma_f32_32x32x16_bf16 + V v_fma_f32 per MFMA", valu_per_mfma=v,
                     issued_by="a second wave of the same SIMD" if other else "the MFMA wave, between the MFMAs", operands="random")
            rows.append(r)
            print(r, flush=True)
    return {"what": "bare MFMA loops with in-kernel clock stamps: one 256-thread block per CU (one MFMA wave per SIMD), 16 x 4000 MFMAs per wave",
            "rows": rows}


def kernels(warm_s):
    from codeps_amd import _lib
    from codeps_amd import functional as F
    lib = _lib.lib()
    if not hasattr(lib, "codeps_stamp_buffer_conv"):
        raise SystemExit("--kernels needs the stamped diagnostic build: CODEPS_HIP_LIB=codeps_amd/variants/libcodeps_hip_stamps.so")
    lib.codeps_conv2d_last_kernel.restype = ctypes.c_char_p
    dev = torch.device("cuda:0")
    cap = 1 << 16
    stamps = torch.zeros(cap * 4, dtype=torch.int64, device=dev)
    for setter in (lib.codeps_stamp_buffer_conv, lib.codeps_stamp_buffer_chain):
        setter.argtypes, setter.restype = [ctypes.c_void_p, ctypes.c_uint], ctypes.c_int
        assert setter(ctypes.c_void_p(stamps.data_ptr()), cap) == 0
    g = torch.Generator(device="cpu").manual_seed(1)
    rows = []

    arith16 = _lib.lib().codeps_conv2d_set_arith(-1) == 16

    def h16(kname):
        """the launch ran the two-fp16-piece arithmetic: its name says so, or (weight gradient / chain rows, named by this tool) the engine's default does"""
        return "f16x2" in kname or "_h_kernel" in kname or (arith16 and ("wgrad_halo64s" in kname or "chain_expand" in kname))

    def measure(name, fn, stamped_part, flops, note=""):
        """fn(): enqueue the launches; stamped_part(): the sub-sequence whose stamps are read (run once after the warm-up)."""
        fn()
        torch.cuda.synchronize()
        t0 = time.time()
        while time.time() - t0 < warm_s:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        kname = stamped_part(stamps)
        torch.cuda.synchronize()
        s = stamps.cpu().numpy().reshape(-1, 4).astype(np.float64)
        s = s[s[:, 1] > 0]
        if len(s) == 0:
            print({"layer": name, "kernel": kname, "skipped": "the dispatcher picked a kernel that carries no stamps"}, flush=True)
            return
        ghz = float(np.median(s[:, 0] / s[:, 1]) * 0.1)
        life = float(np.median(s[:, 0]))
        row = {"layer": name, "kernel": kname, "blocks_stamped": int(len(s)), "us_per_launch_stamped_build": round(us, 1), "in_kernel_ghz": round(ghz, 3),
               "ghz_p10_p90": [round(float(np.percentile(s[:, 0] / s[:, 1], 10) * 0.1), 3), round(float(np.percentile(s[:, 0] / s[:, 1], 90) * 0.1), 3)],
               "block_lifetime_cycles_median": round(life), "algorithmic_tflops_stamped_build": round(flops / us / 1e6, 1),
               # fraction of the matrix pipe's cycles that the piece products per fp32 multiply need (three on the fp16 form -- the kernel's
               # name says so --, six on the bf16 one), AT THE CLOCK THE KERNEL RAN AT
               "products_per_multiply": 3 if h16(kname) else 6,
               "mfma_cycles_needed_over_cycles_elapsed": round((3.0 if h16(kname) else 6.0) * flops / PEAK_FLOP_PER_CYCLE_SIMD / N_SIMD / (us * 1e-6 * ghz * 1e9), 3)}
        if s[:, 2].sum() > 0:
            row["phase_mfma_share_of_lifetime"] = round(float(np.median(s[:, 2] / s[:, 0])), 3)
            row["phase_staging_share_of_lifetime"] = round(float(np.median(s[:, 3] / s[:, 0])), 3)
        if note:
            row["note"] = note
        rows.append(row)
        print(row, flush=True)

    def conv_case(name, b, h, w, ci, co, k, wgrad=False):
        x = torch.randn(b, h, w, ci, generator=g).to(dev)
        wt = (torch.randn(co, ci, k, k, generator=g) * 0.05).to(dev)
        flops = 2.0 * b * h * w * ci * co * k * k
        if not wgrad:
            def fn():
                F.conv2d(x, wt, pad=k // 2)

            def part(st):
                st.zero_()
                F.conv2d(x, wt, pad=k // 2)
                return lib.codeps_conv2d_last_kernel().decode()
            measure(name, fn, part, flops)
        else:
            wt.requires_grad_(True)
            y = F.conv2d(x, wt, pad=k // 2)
            gy = torch.randn_like(y)

            def fn():
                wt.grad = None
                F.conv2d(x, wt, pad=k // 2).backward(gy)

            def part(st):
                wt.grad = None
                yy = F.conv2d(x, wt, pad=k // 2)
                torch.cuda.synchronize()
                st.zero_()
                yy.backward(gy)                        # weight gradient only (x needs no gradient): the stamps are the wgrad kernel's
                return "wgrad_halo64s_kernel (weight gradient; forward launch not stamped here)"
            # timed unit = forward + weight gradient + slab reduction; only clock and phase shares are read from it
            measure(name, fn, part, 2.0 * flops, note="us / TFLOP/s cover forward + weight gradient + slab reduction; clock and shares are the weight-gradient kernel's")

    conv_case("ResNet-101 layer3 3x3 (B8 32x64 256->256)", 8, 32, 64, 256, 256, 3)
    conv_case("ResNet-101 layer2 3x3 (B8 64x128 128->128)", 8, 64, 128, 128, 128, 3)
    conv_case("1x1 reduce, deep K (B8 32x64 1024->512)", 8, 32, 64, 1024, 512, 1)
    conv_case("1x1 layer2 reduce (B8 64x128 512->128)", 8, 64, 128, 512, 128, 1)
    conv_case("1x1 layer2 expand (B8 64x128 128->512)", 8, 64, 128, 128, 512, 1)
    conv_case("1x1 layer1 expand, HBM-bound (B8 128x256 64->256)", 8, 128, 256, 64, 256, 1)
    conv_case("pose encoder layer1 wgrad (B12 128x256 64->64 k3)", 12, 128, 256, 64, 64, 3, wgrad=True)
    conv_case("pose encoder layer3 wgrad (B12 32x64 256->256 k3)", 12, 32, 64, 256, 256, 3, wgrad=True)
    # chained 1x1 pair of layer3
    b, h, w, n2 = 8, 32, 64, 1024
    y1 = torch.randn(b, h, w, 256, generator=g).to(dev)
    res = torch.randn(b, h, w, n2, generator=g).to(dev)
    w2 = (torch.randn(n2, 256, 1, 1, generator=g) * 0.05).to(dev)
    w3 = (torch.randn(256, n2, 1, 1, generator=g) * 0.03).to(dev)
    s2, t2 = torch.rand(n2, generator=g).to(dev) + 0.5, torch.randn(n2, generator=g).to(dev) * 0.1
    s3, t3 = torch.rand(256, generator=g).to(dev) + 0.5, torch.randn(256, generator=g).to(dev) * 0.1

    def chain():
        with torch.no_grad():
            F.chain_expand_reduce(y1, res, w2, s2, t2, w3, s3, t3)

    def chain_part(st):
        st.zero_()
        chain()
        return "chain_expand_reduce_h_kernel" if arith16 else "chain_expand_reduce_kernel"
    measure("layer3 chained 1x1 pair (B8 32x64 256->1024->256)", chain, chain_part, 2.0 * b * h * w * (n2 * 256 + 256 * n2))
    return {"what": "diagnostic build with in-kernel clock stamps (-DCODEPS_CLOCK_STAMPS): wave 0 of every block stamps its lifetime; run times of "
                    "this build are NOT benchmarks", "rows": rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bare", action="store_true")
    ap.add_argument("--kernels", action="store_true")
    ap.add_argument("--warm", type=float, default=2.0)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    out = {"device": torch.cuda.get_device_name(0), "warm_seconds_of_back_to_back_launches": args.warm}
    if args.bare:
        out["bare"] = bare(args.warm)
    if args.kernels:
        out["kernels"] = kernels(args.warm)
    if args.json:
        json.dump(out, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
