#!/usr/bin/env python3
"""Accuracy of split-precision bf16 MFMA emulation of an fp32 GEMM: a = hi + lo (+ mid) with
bf16 pieces, product = sum of the piece products that matter, fp32 accumulation on the hardware (fp64 here, so the numbers
isolate the representation error).  CPU only; run: python tools/split_precision_study.py"""
import torch


def pieces(x, n):
    out, r = [], x
    for _ in range(n):
        p = r.bfloat16().float()
        out.append(p)
        r = r - p
    return out


def main():
    torch.manual_seed(0)
    for k in (256, 2304, 18432):                       # 1x1 256-ch, 3x3 256-ch, 3x3 2048-ch contractions
        a = torch.relu(torch.randn(512, k))           # post-ReLU activations
        b = torch.randn(k, 256) * 0.05
        ref = a.double() @ b.double()
        rel = lambda x: float((x - ref).norm() / ref.norm())
        a2, b2 = pieces(a, 2), pieces(b, 2)
        x3 = sum(a2[i].double() @ b2[j].double() for i in range(2) for j in range(2) if i + j <= 1)
        a3, b3 = pieces(a, 3), pieces(b, 3)
        x6 = sum(a3[i].double() @ b3[j].double() for i in range(3) for j in range(3) if i + j <= 2)
        print(f"K={k:6d}  rel-L2 error: fp32 GEMM {rel((a @ b).double()):.2e}   bf16x3 {rel(x3):.2e}   bf16x6 {rel(x6):.2e}")


if __name__ == "__main__":
    main()
