#!/usr/bin/env python3
"""Two fp16 pieces / three products against three bf16 pieces / six products, kernel by kernel.

    python tools/f16_stage_a.py [--json profiles/r06_f16_stage_a.json] [--skip-accuracy]

Timing: the layers named in the verdict (ResNet-101 layer3 3x3: B8 32x64 256->256; the K-deep decoder layer: B6 32x64 256(up)+1024->256) and a few
more of the step's 3x3 / 1x1 shapes, forward and backward, HIP-event time per launch under arithmetic 3 and 16 alternately on one box.
Accuracy: relative L2 error against float64 (torch CPU) of forward, data gradient and weight gradient on the `normal / wide / cancel` operand
kinds of tests/test_precision_gpu.py plus `grad` (1e-6-scale heavy-tailed operands, the gradient case), beside torch's own fp32 error."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.nn.functional as TF  # noqa: E402

DEV = "cuda:0"

TIMED = {               # name: (B, H, W, C0, C1, Cout, k, up0)
    "l3_3x3": (8, 32, 64, 256, 0, 256, 3, False),
    "dec_k1280": (6, 32, 64, 256, 1024, 256, 3, True),
    "l2_3x3": (8, 64, 128, 128, 0, 128, 3, False),
    "l1_3x3": (8, 128, 256, 64, 0, 64, 3, False),
    "l4_3x3": (8, 16, 32, 512, 0, 512, 3, False),
    "pose_l2": (12, 64, 128, 128, 0, 128, 3, False),
    "pose_l1": (12, 128, 256, 64, 0, 64, 3, False),
    "dec_64": (6, 128, 256, 64, 256, 64, 3, True),
    "dec_32col": (6, 256, 512, 32, 64, 32, 3, True),
    "l1_expand": (8, 128, 256, 64, 0, 256, 1, False),
    "l2_expand": (8, 64, 128, 128, 0, 512, 1, False),
    "l3_reduce": (8, 32, 64, 1024, 0, 256, 1, False),
}


def set_arith(a):
    from codeps_amd._lib import lib
    return lib().codeps_conv2d_set_arith(a)


def last():
    from codeps_amd._lib import lib
    return lib().codeps_conv2d_last_kernel().decode()


def time_layer(F, name, iters, bwd):
    b, h, w, c0, c1, co, k, up = TIMED[name]
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(b, h // 2 if up else h, w // 2 if up else w, c0, generator=g).relu().to(DEV)
    x1 = torch.randn(b, h, w, c1, generator=g).relu().to(DEV) if c1 else None
    wt = (torch.randn(co, c0 + c1, k, k, generator=g) * (2.0 / ((c0 + c1) * k * k)) ** 0.5).to(DEV)
    if bwd:
        x0.requires_grad_(True)
        wt.requires_grad_(True)
    gy = torch.randn(b, h, w, co, generator=g).to(DEV) * 1e-4
    res = {}
    for arith in (3, 16, 3, 16):
        set_arith(arith)
        y = F.conv2d(x0, wt, x1=x1, pad=k // 2, up0=up)
        kname = last()
        if bwd:
            y.backward(gy)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            y = F.conv2d(x0, wt, x1=x1, pad=k // 2, up0=up)
            if bwd:
                y.backward(gy)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        res.setdefault(arith, []).append((us, kname))
    flops = 2.0 * b * h * w * (c0 + c1) * co * k * k * (3 if bwd else 1)
    out = {"shape": f"B{b} {h}x{w} {c0}+{c1}->{co} k{k}" + (" up" if up else ""), "bwd": bwd}
    for arith, runs in res.items():
        us = min(r[0] for r in runs)
        out[f"arith{arith}"] = {"us": round(us, 1), "tflops": round(flops / us / 1e6, 1), "kernel": runs[0][1]}
    out["speedup"] = round(out["arith3"]["us"] / out["arith16"]["us"], 3)
    return out


def l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def operands(kind, shape_x, shape_w, seed):
    g = torch.Generator().manual_seed(seed)
    if kind == "normal":
        x = torch.relu(torch.randn(shape_x, generator=g))
        w = torch.randn(shape_w, generator=g) * (2.0 / (shape_w[1] * shape_w[2] * shape_w[3])) ** 0.5
    elif kind == "wide":
        x = torch.randn(shape_x, generator=g) * torch.exp2(torch.randint(-12, 13, shape_x, generator=g).float())
        w = torch.randn(shape_w, generator=g) * torch.exp2(torch.randint(-6, 7, shape_w, generator=g).float())
    elif kind == "cancel":
        x = 1.0 + 1e-3 * torch.randn(shape_x, generator=g)
        w = torch.randn(shape_w, generator=g)
        w = w - w.mean(dim=(1, 2, 3), keepdim=True)
    elif kind == "grad":                    # 1e-6-scale, heavy-tailed (student-t like): what a gradient tensor looks like
        x = 1e-6 * torch.randn(shape_x, generator=g) / (torch.rand(shape_x, generator=g) + 0.02)
        w = torch.randn(shape_w, generator=g) * (2.0 / (shape_w[1] * shape_w[2] * shape_w[3])) ** 0.5
    else:                                   # "big": magnitudes up to 1e4
        x = 1e4 * torch.randn(shape_x, generator=g)
        w = torch.randn(shape_w, generator=g)
    return x, w, g


ACC = {                 # name: (B, H, W, Cin, Cout, k)
    "halo3r_256": (2, 64, 128, 128, 256, 3),
    "halo3r_dgrad": (2, 64, 128, 256, 256, 3),
    "halo3r_splitk": (4, 32, 64, 512, 256, 3),
    "deep_k": (2, 32, 64, 2048, 256, 3),
    "halo3_64": (4, 64, 128, 64, 64, 3),
    "igemm_1x1": (8, 32, 64, 256, 1024, 1),
    "igemm_1x1_b": (8, 32, 64, 512, 2048, 1),
    "igemm_odd": (3, 40, 72, 96, 160, 3),
}


def accuracy(F, case, kind):
    b, h, w_, ci, co, k = ACC[case]
    x, wt, g = operands(kind, (b, ci, h, w_), (co, ci, k, k), 77 + len(case) + len(kind))
    gy = torch.randn(b, co, h, w_, generator=g)
    if kind in ("wide", "grad"):
        gy = gy * torch.exp2(torch.randint(-8, 9, gy.shape, generator=g).float()) * (1e-6 if kind == "grad" else 1.0)

    def ref(dtype):
        xr, wr = x.clone().to(dtype).requires_grad_(True), wt.clone().to(dtype).requires_grad_(True)
        y = TF.conv2d(xr, wr, None, 1, k // 2)
        y.backward(gy.to(dtype))
        return y.detach(), xr.grad, wr.grad
    y64, dx64, dw64 = ref(torch.float64)
    y32, dx32, dw32 = ref(torch.float32)
    out = {"torch_fp32": {"fwd": l2(y32, y64), "dgrad": l2(dx32, dx64), "wgrad": l2(dw32, dw64)}}
    for arith in (0, 3, 16):
        set_arith(arith)
        xd = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
        wd = wt.to(DEV).requires_grad_(True)
        y = F.conv2d(xd, wd, pad=k // 2)
        kf = last()
        y.backward(gy.permute(0, 2, 3, 1).contiguous().to(DEV))
        yy = y.detach()[..., :co].permute(0, 3, 1, 2).cpu()
        out[f"arith{arith}"] = {"fwd": l2(yy, y64), "dgrad": l2(xd.grad[..., :ci].permute(0, 3, 1, 2).cpu(), dx64), "wgrad": l2(wd.grad.cpu(), dw64),
                                "fwd_kernel": kf, "finite": bool(torch.isfinite(yy).all() and torch.isfinite(xd.grad).all())}
    set_arith(3)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--shapes", default=",".join(TIMED))
    ap.add_argument("--skip-accuracy", action="store_true")
    ap.add_argument("--skip-bwd", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(8)
    from codeps_amd import functional as F
    res = {"timing": {}, "accuracy": {}}
    for name in args.shapes.split(","):
        for bwd in ((False,) if args.skip_bwd else (False, True)):
            r = time_layer(F, name, args.iters, bwd)
            res["timing"][name + ("_fwd_bwd" if bwd else "_fwd")] = r
            print(name, "bwd" if bwd else "fwd", json.dumps(r), flush=True)
    if not args.skip_accuracy:
        for case in ACC:
            for kind in ("normal", "wide", "cancel", "grad", "big"):
                r = accuracy(F, case, kind)
                res["accuracy"][f"{case}/{kind}"] = r
                print(case, kind, json.dumps(r), flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
