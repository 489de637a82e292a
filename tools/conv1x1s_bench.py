#!/usr/bin/env python3
"""1x1 convolution with pre-split activations (csrc/conv1x1s.hip: both operands by LDS DMA) against the conv engine's 1x1 path on
the same tensors: agreement, HIP-event time per launch (the split of the activations is NOT in the timed region: in the intended
pipeline the producing layer's epilogue emits that form) and the cost of a stand-alone split pass.
    python tools/conv1x1s_bench.py [--shapes l3_expand,l3_reduce,...]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

SHAPES = {               # name: (B, H, W, Cin, Cout)
    "l3_expand": (8, 32, 64, 256, 1024), "l3_reduce": (8, 32, 64, 1024, 256),
    "l2_expand": (8, 64, 128, 128, 512), "l2_reduce": (8, 64, 128, 512, 128),
    "l1_expand": (8, 128, 256, 64, 256), "l1_reduce": (8, 128, 256, 256, 64),
    "l4_expand": (8, 16, 32, 512, 2048), "l4_reduce": (8, 16, 32, 2048, 512),
}


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    from codeps_amd import functional as F, _lib
    lib = _lib.diag_lib()
    lib.codeps_conv2d_last_kernel.restype = ctypes.c_char_p
    dev = torch.device("cuda", 0)
    print(f"{'shape':10s} {'engine kernel':28s} {'engine us':>9s} {'TF/s':>6s} | {'split-in us':>11s} {'TF/s':>6s} {'max|diff|':>10s} | {'split pass us':>13s}")
    for name in args.shapes.split(","):
        b, h, w, ci, co = SHAPES[name]
        g = torch.Generator().manual_seed(3)
        x = torch.randn(b, h, w, ci, generator=g).to(dev)
        wt = (torch.randn(co, ci, 1, 1, generator=g) * 0.05).to(dev)
        scale, shift = (torch.rand(co, generator=g) + 0.5).to(dev), torch.randn(co, generator=g).to(dev)
        res = torch.randn(b, h, w, co, generator=g).to(dev)
        ref = F.conv2d(x, wt, scale=scale, shift=shift, residual=res, act="relu")
        kname = lib.codeps_conv2d_last_kernel().decode()
        t_ref = timeit(lambda: F.conv2d(x, wt, scale=scale, shift=shift, residual=res, act="relu"), args.iters)
        wf, _ = F.pack_weights(wt, co, ci, False)
        w_split = wf.view(-1)[co * ci:]                                 # the pre-split copy behind the fp32 part
        rows = b * h * w
        xs = torch.empty(rows * ci * 6, dtype=torch.uint8, device=dev)
        y = torch.empty(b, h, w, co, device=dev)
        split = lambda: _lib.check_diag(lib.codeps_split_pack_rows(_lib.ptr(x), rows, ci, ci, _lib.ptr(xs), _lib.stream_ptr()), "split")   # noqa: E731
        conv = lambda: _lib.check_diag(lib.codeps_conv1x1_split_fwd(_lib.ptr(xs), rows, ci, co, _lib.ptr(w_split), _lib.ptr(scale), _lib.ptr(shift),   # noqa: E731
                                                               _lib.ptr(res), 1, _lib.ptr(y), _lib.stream_ptr()), "conv1x1s")
        split()
        conv()
        torch.cuda.synchronize()
        diff = (y - ref).abs().max().item()
        t_new, t_split = timeit(conv, args.iters), timeit(split, args.iters)
        fl = 2.0 * rows * ci * co
        print(f"{name:10s} {kname:28s} {t_ref:9.1f} {fl / t_ref / 1e6:6.1f} | {t_new:11.1f} {fl / t_new / 1e6:6.1f} {diff:10.2e} | {t_split:13.1f}", flush=True)


if __name__ == "__main__":
    main()
