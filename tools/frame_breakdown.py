#!/usr/bin/env python3
"""Where a frame's evaluation passes (codeps.frame.pre_step) spend their time: host wall vs device time per call, launches."""
import os, sys, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from codeps_amd import synthetic
from codeps_amd.codeps import pre_step
from codeps_amd.step import StepRunner

dev = torch.device("cuda", 0)
net, cfg = bench.build(dev, 101, 512, 1024, "adapt", None)
data, _ = bench.make_data("adapt", 512, 1024, dev, 0, 8, False, 1)
runner = StepRunner(net, cfg, "adapt", 1)
runner.step(data)
ema = copy.deepcopy(net)
th, tw = data["target"]["rgb"][0].shape[-2:]
raw_mix = synthetic.mixup_raw_item(data["cut_mixup"]["rgb"][0].shape[0], 512, 1024, th, tw, dev, 4321)

def one():
    sample = {k: (dict(v) if isinstance(v, dict) else v) for k, v in data.items() if not k.endswith("mixup")}
    sample["cut_mixup"] = dict(raw_mix)
    return pre_step(net, ema, ["semantic_head"], sample, [0, -1, 1], True)

for _ in range(3):
    one()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 10
t0 = time.perf_counter(); e0.record()
for _ in range(N):
    one()
e1.record(); t_issue = time.perf_counter() - t0
torch.cuda.synchronize(); t_wall = time.perf_counter() - t0
print(f"pre_step: host issue {t_issue / N * 1e3:.2f} ms, wall {t_wall / N * 1e3:.2f} ms, device span {e0.elapsed_time(e1) / N:.2f} ms")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    one(); torch.cuda.synchronize()
ka = prof.key_averages()
dev_total = sum(k.device_time_total for k in ka if k.device_type.name == "CUDA") if hasattr(ka[0], "device_type") else 0
rows = sorted([k for k in ka if getattr(k, "device_time_total", 0) > 0 and k.device_type.name == "CUDA"], key=lambda k: -k.device_time_total)
n_launch = sum(k.count for k in rows)
print(f"device kernels: {n_launch} launches, {sum(k.device_time_total for k in rows) / 1e3:.2f} ms")
for k in rows[:25]:
    print(f"  {k.device_time_total / 1e3:7.3f} ms {k.count:4d}  {k.key[:100]}")
