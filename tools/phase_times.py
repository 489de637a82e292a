#!/usr/bin/env python3
"""Where one eagerly launched adaptation step spends its time, by phase and stream (HIP events): forward (main stream), backward on the
main stream, when the pose / branch side streams finish their backward work, optimizer.
    python tools/phase_times.py [--steps 10]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import bench
    from codeps_amd import functional as F
    from codeps_amd.step import StepRunner
    dev = torch.device("cuda:0")
    net, cfg = bench.build(dev, 101, 512, 1024, "adapt")
    data, _ = bench.make_data("adapt", 512, 1024, dev, 0, 8)
    runner = StepRunner(net, cfg, "adapt", 1)
    for _ in range(4):
        runner.step(data)
    torch.cuda.synchronize()
    ev = lambda: torch.cuda.Event(enable_timing=True)      # noqa: E731
    acc = {}
    main_s = torch.cuda.current_stream()
    for _ in range(args.steps):
        e = {k: ev() for k in ("t0", "fwd", "bwd_main", "bwd_pose", "bwd_branch", "join", "opt")}
        e["t0"].record()
        F.set_noise_stream(runner.noise_counter, runner.seed_base)
        F.step_counter_add(runner.noise_counter, 1)
        F.refresh_weight_packs()
        for g in runner.groups:
            g.grad.zero_()
        losses, result, _, _ = net(data, "adapt", [0, -1, 1], do_panoptic_fusion=True)
        terms, wts = [], []
        for k, v in losses.items():
            w = runner.weights.get(k)
            if v is None or w is None or not v.requires_grad:
                continue
            terms.append(v)
            wts.append(float(w))
        total = F.weighted_sum(terms, wts)
        e["fwd"].record()
        total.backward()
        e["bwd_main"].record()
        if getattr(net, "pose_stream", None) is not None:
            e["bwd_pose"].record(net.pose_stream)
        if getattr(net, "branch_stream", None) is not None:
            e["bwd_branch"].record(net.branch_stream)
        for side in (getattr(net, "pose_stream", None), getattr(net, "branch_stream", None)):
            if side is not None:
                main_s.wait_stream(side)
        e["join"].record()
        runner._touched = None
        runner.optimizer_step()
        e["opt"].record()
        torch.cuda.synchronize()
        for k in ("fwd", "bwd_main", "bwd_pose", "bwd_branch", "join", "opt"):
            try:
                acc.setdefault(k, []).append(e["t0"].elapsed_time(e[k]))
            except Exception:      # noqa: BLE001  (a stream that does not exist)
                pass
    print("ms since the start of the step (median over %d eager steps):" % args.steps)
    for k, v in acc.items():
        v = sorted(v)
        print(f"  {k:11s} {v[len(v) // 2]:7.2f}")


if __name__ == "__main__":
    main()
