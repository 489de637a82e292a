#!/usr/bin/env python3
"""What v_mfma_f32_16x16x32_f16 does with its operands (stage A (i)): ONE instruction on chosen bit patterns through
libcodeps_diag.so's codeps_diag_mfma_once, against float64.

    python tools/f16_mfma_probe.py [--json profiles/r06_f16_mfma_probe.json]

Questions and how each is asked:
  subnormal_inputs   a = 2^-20 (an fp16 subnormal) x b = 1 summed over k: 32 * 2^-20 if honoured, 0 if flushed
  subnormal_product  a = 2^-24 (the smallest subnormal) x b = 2^-24: 2^-48 per product, far below fp16 but a normal fp32
  exact_products     (1 + 2^-10)^2 = 1 + 2^-9 + 2^-20 needs 21 significand bits: exact in fp32, lost if products were rounded to fp16
  wide_sum           2^20 + 2^-3 inside ONE instruction (24 bits apart: representable in fp32 only if the internal sum keeps them)
  random             random fp16 operands, |error| of d against the float64 dot product, in units of the fp32 ulp of sum |a b|
The bf16 instruction runs the same questions (its 8-bit pieces make every product exact in fp32 by construction)."""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def run(lib, _lib, mode, a16, b16, c=None):
    dev = torch.device("cuda", 0)
    a = torch.from_numpy(a16.view(np.int16).copy()).to(dev)
    b = torch.from_numpy(b16.view(np.int16).copy()).to(dev)
    ct = torch.zeros(16, 16, device=dev) if c is None else torch.from_numpy(c.astype(np.float32)).to(dev)
    d = torch.empty(16, 16, device=dev)
    _lib.check_diag(lib.codeps_diag_mfma_once(mode, _lib.ptr(a), _lib.ptr(b), _lib.ptr(ct), _lib.ptr(d), _lib.stream_ptr()), "codeps_diag_mfma_once")
    torch.cuda.synchronize()
    return d.cpu().numpy()


def bf16_of(x):
    """float32 array -> bf16 bit patterns (round to nearest even), as uint16"""
    u = np.asarray(x, np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def bf16_to_f64(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    from codeps_amd import _lib
    lib = _lib.diag_lib()
    out = {}
    for mode, name in ((0, "f16"), (1, "bf16")):
        enc = (lambda x: np.asarray(x, np.float64).astype(np.float16).view(np.uint16)) if mode == 0 else (lambda x: bf16_of(np.asarray(x, np.float32)))
        dec = (lambda u: u.view(np.float16).astype(np.float64)) if mode == 0 else bf16_to_f64
        res = {}
        ones = enc(np.ones((16, 32)))
        if mode == 0:
            d = run(lib, _lib, mode, enc(np.full((16, 32), 2.0 ** -20)), ones)
            res["subnormal_inputs"] = {"got": float(d[0, 0]), "honoured": float(32 * 2.0 ** -20), "flushed": 0.0}
            d = run(lib, _lib, mode, enc(np.full((16, 32), 2.0 ** -24)), enc(np.full((16, 32), 2.0 ** -24)))
            res["subnormal_product"] = {"got": float(d[0, 0]), "honoured": float(32 * 2.0 ** -48)}
            a = np.zeros((16, 32)); a[:, 0] = 1 + 2.0 ** -10
            d = run(lib, _lib, mode, enc(a), enc(a))
            res["exact_products"] = {"got_minus_1": float(np.float64(d[0, 0]) - 1.0), "exact_minus_1": 2.0 ** -9 + 2.0 ** -20}
            a = np.zeros((16, 32)); b = np.zeros((16, 32))
            a[:, 0] = 1024.0; b[:, 0] = 1024.0; a[:, 17] = 2.0 ** -2; b[:, 17] = 0.5
            d = run(lib, _lib, mode, enc(a), enc(b))
            res["wide_sum"] = {"got_minus_2^20": float(np.float64(d[0, 0]) - 2.0 ** 20), "exact": 2.0 ** -3}
        else:
            d = run(lib, _lib, mode, enc(np.full((16, 32), 2.0 ** -130)), ones)     # a bf16 subnormal
            res["subnormal_inputs"] = {"got": float(d[0, 0]), "honoured": float(32 * 2.0 ** -130), "flushed": 0.0}
        rng = np.random.default_rng(0)
        worst, rms = 0.0, []
        for kind in ("normal", "positive", "wide"):
            for _ in range(8):
                a = rng.standard_normal((16, 32)); b = rng.standard_normal((16, 32))
                if kind == "positive": a, b = np.abs(a), np.abs(b)
                if kind == "wide": a *= np.exp2(rng.integers(-6, 7, a.shape)); b *= np.exp2(rng.integers(-6, 7, b.shape))
                ae, be = enc(a), enc(b)
                d = run(lib, _lib, mode, ae, be).astype(np.float64)
                af, bf = dec(ae), dec(be)
                ref = af @ bf.T
                mag = np.abs(af) @ np.abs(bf).T
                ulp = np.exp2(np.floor(np.log2(mag)) - 23)
                e = np.abs(d - ref) / ulp
                worst = max(worst, float(e.max())); rms.append(float(np.sqrt((e ** 2).mean())))
        res["random"] = {"worst_error_in_ulp_of_sum_abs": worst, "rms": float(np.mean(rms)),
                         "note": "0.5 = a correctly rounded sum of exact products; an fp32 FMA chain over 32 terms typically reaches 1-3"}
        out[name] = res
    print(json.dumps(out, indent=1))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
