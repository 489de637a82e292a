#!/usr/bin/env python3
"""Which ATen / runtime kernels does one eagerly launched adaptation step still issue, and from which source line?
    python tools/aten_audit.py [--encoder 101] > gpurun_out/aten_audit.txt
Uses torch.profiler (with_stack) around one StepRunner.step of the bench workload; prints device-kernel launches that are not
this library's, grouped by the innermost frame inside codeps_amd/."""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--encoder", type=int, default=None)
    ap.add_argument("--mode", default="adapt", choices=["adapt", "train"])
    ap.add_argument("--batch", type=int, default=None)
    args = ap.parse_args()
    mode = args.mode
    enc = args.encoder or (101 if mode == "adapt" else 50)
    batch = args.batch or (8 if mode == "adapt" else 4)
    import bench
    from codeps_amd.step import StepRunner
    dev = torch.device("cuda:0")
    net, cfg = bench.build(dev, enc, 512, 1024, mode)
    data, _ = bench.make_data(mode, 512, 1024, dev, 0, batch)
    runner = StepRunner(net, cfg, mode, 1)
    for _ in range(3):
        runner.step(data)
    torch.cuda.synchronize()
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    by_site = collections.Counter()
    shapes = collections.defaultdict(collections.Counter)
    LAUNCHING = ("copy_", "fill_", "add", "mul", "zero_", "zeros", "clone", "contiguous", "cat", "stack", "to", "_to_copy", "sum", "argmax",
                 "index", "pad", "constant_pad_nd", "sub", "div", "rsqrt", "where", "eq", "ones", "full", "masked_fill", "neg", "sqrt",
                 "slice_backward", "select_backward", "_foreach_add_", "mean", "max", "min", "exp", "abs", "linalg_vector_norm", "new_zeros",
                 "zeros_like", "ones_like", "empty_like", "index_copy_", "index_select", "repeat", "expand", "permute")
    NOLAUNCH = ("empty", "empty_like", "empty_strided", "view", "_unsafe_view", "reshape", "permute", "expand", "slice", "select", "t",
                "transpose", "as_strided", "detach", "alias", "unsqueeze", "squeeze", "_reshape_alias", "unbind", "split", "narrow",
                "lift_fresh", "is_same_size", "sym_size", "stride", "size", "numel", "_local_scalar_dense", "item", "record_stream")

    class Audit(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = func.__name__.split(".")[0]
            out = func(*args, **(kwargs or {}))
            if name in NOLAUNCH:
                return out
            site = "<autograd engine>"
            for fr in reversed(traceback.extract_stack()):
                if "codeps_amd/" in fr.filename or fr.filename.endswith("bench.py"):
                    site = f"{fr.filename.split('codeps_amd/')[-1]}:{fr.lineno} {fr.name}"
                    break
            t = next((a for a in args if torch.is_tensor(a)), None)
            by_site[(site, name)] += 1
            shapes[(site, name)][tuple(t.shape) if t is not None else ()] += 1
            return out

    with Audit():
        runner.step(data)
    torch.cuda.synchronize()
    print(f"dispatched ATen ops that may launch, one step: {sum(by_site.values())}")
    for (site, name), n in by_site.most_common():
        print(f"{n:5d}  {name:22s} {site}   " + ", ".join(f"{list(k)} x{v}" for k, v in shapes[(site, name)].most_common(3)))


if __name__ == "__main__":
    main()
