#!/usr/bin/env python3
"""Benchmark of the CoDEPS hot path on MI355X: adapt-step images/sec (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one continual-adaptation step (scripts/adapt_codeps.py:260-299 in the reference): forward of
CodepsNet.adapt (frozen ResNet-101 encoder + Depth/Semantic/Instance/Pose heads + photometric, smoothness,
semantic losses + panoptic fusion of the target frame), backward, ONE all-reduce of the flat gradient
buffer (N > 1), fused Adam + EMA -- on the per-GPU batch of BASELINE config 3: 8 centre images
(1 target + 3 target_replay + 2 source + 2 cut_mixup) at 512x1024, synthetic data, random-init weights
(seed 20).  Weak scaling: every rank processes its own 8 images.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2516.8      # MI355X_MICROARCH.md: f32 MFMA = 1/16 of the bf16 MFMA rate (~2.5 PF dense)
PEAK_HBM_GBS = 8000.0


def build(device, encoder_layers, h, w, mode, target_hw=None):
    from codeps_amd.codeps import gen_models
    from codeps_amd.config import get_cfg_shipped
    # defaults + the shipped YAML of the mode (cfg/adapt_cityscapes_kitti_360.yaml / cfg/train_cityscapes.yaml), io_utils.py:220-237
    cfg = get_cfg_shipped(adapt=(mode == "adapt"))
    cfg.model.po_depth_net.params.nof_layers = encoder_layers
    cfg.dataset.feed_img_size = list(target_hw) if target_hw is not None else [h, w]
    cfg.adapt.source_dataset.feed_img_size = [h, w]
    torch.manual_seed(20)                   # the scripts' default seed (scripts/train_codeps.py:51)
    net = gen_models(cfg, device, list(range(9)), [9, 10, 11, 12, 13], [], adaptation_mode=(mode == "adapt"))
    return net, cfg


def adapt_split(per_gpu, rank, world):
    """Items per key (target, target_replay, source, cut_mixup) of one rank.
    per_gpu % 8 == 0: BASELINE configs[2] scaled, 1 + 3 + 2 + 2 per 8 items on every rank.
    per_gpu == 4:     BASELINE configs[3] (SURVEY.md 8d "Config 4"): 2 x {target | target_replay} + 1 source + 1 cut_mixup per
                      rank; the lower half of the ranks hold 1 target + 1 replay item, the upper half 2 replay items (N = 1: the
                      lower-half composition), so that a global batch of 4 N items is N/2 target + 3N/2 replay + N source + N mixup."""
    if per_gpu % 8 == 0:
        k = per_gpu // 8
        return k, 3 * k, 2 * k, 2 * k
    if per_gpu == 4:
        n_t = 1 if rank < max(world // 2, 1) else 0
        return n_t, 2 - n_t, 1, 1
    raise SystemExit(f"--per-gpu-batch {per_gpu}: the adapt step is defined for 4 (configs[3]) or a multiple of 8 (configs[2])")


def make_data(mode, h, w, device, rank, per_gpu, yaml_faithful=False, world=1):
    from codeps_amd import synthetic
    if mode == "adapt" and yaml_faithful:      # cfg/adapt_cityscapes_kitti_360.yaml as written: 1 + 2 target-domain items at 384x1408
        return synthetic.adapt_batch(h, w, device, 1234 + rank, 1, 2, 2, 2, target_hw=(384, 1408))
    if mode == "adapt":
        return synthetic.adapt_batch(h, w, device, 1234 + rank, *adapt_split(per_gpu, rank, world))
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    return synthetic.triplet_item(per_gpu, h, w, gen, device, labels=True), per_gpu


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(net, full_h, full_w, budget_s=24.0):
    """The oracle (CPU restatement, kind "port") timed on this host: fwd + bwd + Adam of an adapt step on a BOUNDED sample --
    1 source + 1 target centre image at full resolution -- per SURVEY.md 8d: one warm-up step, then the MEDIAN of >= 5 steps at
    min(usable cores, 32) threads, and the same at 1 thread while the time budget lasts (the 1-thread step is ~10x slower; it
    gets at least one timed step and says how many it took).  Reported baseline only.
    Threads: with all 256 hardware threads of the GPU box the oracle's oneDNN convolutions ran ~100x slower (925 s for this
    sample), so more threads is not "more baseline"."""
    import statistics
    from codeps_amd import synthetic
    from oracle import network
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    cores = max(1, min(usable, 32))
    h, w = full_h, full_w
    sds = {}
    for name in ("backbone_po_depth", "backbone_pose_sflow", "depth_head", "pose_head", "semantic_head", "instance_head"):
        sds[name] = {k: v.detach().cpu().clone() for k, v in getattr(net, name).state_dict().items()}
    params = []
    for n in ("backbone_pose_sflow", "depth_head", "pose_head", "semantic_head"):
        for k, v in sds[n].items():
            if v.is_floating_point() and "running" not in k and ".fc." not in k:
                v.requires_grad_(True)
                params.append(v)
    gen = torch.Generator().manual_seed(99)
    dev = torch.device("cpu")
    data = {"target": synthetic.triplet_item(1, h, w, gen, dev), "source": synthetic.triplet_item(1, h, w, gen, dev, labels=True)}
    cfg = dict(nof_layers=net.backbone_po_depth.num_layers, train_pose_bn=True)
    wts = dict(depth_recon=10.0, depth_smth=0.001, semantic_source=1.0)
    ms, vs = [torch.zeros_like(p) for p in params], [torch.zeros_like(p) for p in params]
    count = [0]

    def one():
        t0 = time.perf_counter()
        for p in params:
            p.grad = None
        out, _ = network.adapt_forward(sds, cfg, data)
        total = sum(wts[k] * out[k] for k in wts)
        total.backward()
        count[0] += 1
        with torch.no_grad():
            for p, m, v in zip(params, ms, vs):
                if p.grad is not None:
                    network.adam_step(p, p.grad, m, v, count[0])
        return time.perf_counter() - t0

    t_begin = time.perf_counter()
    torch.set_num_threads(cores)
    one()                                                    # warm-up (oneDNN primitive creation, allocator)
    multi = [one() for _ in range(5)]
    med = statistics.median(multi)
    torch.set_num_threads(1)
    single = []
    while not single or (time.perf_counter() - t_begin + (single[-1] if single else 0.0) < budget_s and len(single) < 5):
        single.append(one())
    torch.set_num_threads(cores)
    med1 = statistics.median(single)
    return {"value": round(2 / med, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "value_1_thread": round(2 / med1, 4), "steps_timed": 5, "steps_timed_1_thread": len(single),
            "cpu_model": _cpu_model(), "host_cores": usable,
            "sample": f"oracle adapt step (fwd+bwd+Adam, ResNet-{net.backbone_po_depth.num_layers}) on 1 source + 1 target centre "
                      f"image at {h}x{w}: 1 warm-up + median of 5 steps = {med:.2f} s at {cores} torch CPU threads; "
                      f"median of {len(single)} step(s) = {med1:.2f} s at 1 thread (no warm-up beyond the multi-thread steps)"}


def _latest_profile(suffix):
    """Newest profiles/rNN_<suffix> (the rocprofv3 summaries are refreshed per round by tools/refresh_profiles.sh)."""
    import glob
    hits = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r[0-9][0-9]_" + suffix)))
    return hits[-1] if hits else None


def pmc_traffic(default_workload):
    """HBM bytes per conv launch from the committed rocprofv3 PMC summary (FETCH_SIZE / WRITE_SIZE passes of this same
    command, corrected as MI355X_MICROARCH.md prescribes; tools/pmc_traffic.py).  Counters cannot be read from inside
    the process, so the value is the one measured when profiles/ was last refreshed -- None for other workloads."""
    path = _latest_profile("pmc_traffic.json")
    if not default_workload or path is None:
        return None, None
    with open(path) as f:
        d = json.load(f)
    return round(d["conv"]["hbm_mb_per_launch"] * 1e6), f"profiles/{os.path.basename(path)} (avg HBM bytes per conv launch)"


def photometric_pmc(default_workload):
    path = _latest_profile("pmc_traffic.json")
    if not default_workload or path is None:
        return None
    with open(path) as f:
        rows = {r["kernel"]: r for r in json.load(f)["kernels"]}
    ks = [rows[k] for k in ("photo_fwd_kernel", "photo_bwd_kernel") if k in rows]
    if not ks:
        return None
    return round(sum((r["read_mb_per_launch"] + r["write_mb_per_launch"]) * 1e6 for r in ks) / len(ks))


def photometric_shares():
    """Where the wave cycles of the photometric kernels go, from the committed PMC pass (tools/pmc_kernels.py)."""
    path = _latest_profile("pmc_kernels.json")
    if path is None:
        return None
    with open(path) as f:
        rows = {r["kernel"]: r for r in json.load(f)["kernels"]}
    out = {"source": "profiles/" + os.path.basename(path)}
    for k in ("photo_fwd_kernel", "photo_bwd_kernel"):
        if k in rows:
            out[k] = {"avg_us": rows[k]["avg_us"], "avg_waves_per_simd": rows[k]["avg_waves_per_simd"], **rows[k].get("share_of_wave_cycles", {})}
    return out


def clock_record():
    """Cycles per MFMA and the clock the chip held, stamped in the kernels (tools/clock_cycles.py; committed per round)."""
    out = {}
    for key, suffix in (("bare_loops", "clock_cycles_bare.json"), ("kernels", "clock_cycles_kernels.json")):
        path = _latest_profile(suffix)
        if path is None:
            continue
        with open(path) as f:
            d = json.load(f)
        if key == "bare_loops":
            out[key] = [{k: r[k] for k in ("loop", "operands", "cycles_per_mfma", "in_kernel_ghz", "tflops") if k in r}
                        for r in d["bare"]["rows"] if r["loop"].startswith("bare")]
        else:
            out[key] = [{k: r[k] for k in ("layer", "kernel", "in_kernel_ghz", "mfma_cycles_needed_over_cycles_elapsed") if k in r} for r in d["kernels"]["rows"]]
        out.setdefault("source", []).append("profiles/" + os.path.basename(path))
    return out or None


def eval_pass_roofline(run_eval, steps=2):
    """`--mode frame`: the conv engine inside the per-frame evaluation passes (scripts/adapt_codeps.py:166-199: EMA-teacher
    evaluation of the target frame, teacher / student passes over the mix-up items; frozen weights, eval-mode BN folded into the
    conv epilogues), measured like conv_roofline: HIP events around every conv launch, algorithmic FLOPs / event time."""
    from codeps_amd import functional as F
    F._profile["events"].clear()
    F._profile["on"] = True
    for _ in range(steps):
        run_eval()
    torch.cuda.synchronize()
    F._profile["on"] = False
    flops = secs = split_flops = split_secs = 0.0
    n = 0
    for kind, fl, a, b, _tag in F._profile["events"]:
        if not kind.startswith("conv_"):
            continue
        t = a.elapsed_time(b) * 1e-3
        flops += fl; secs += t; n += 1
        if kind.endswith("_split"):
            split_flops += fl; split_secs += t
    F._profile["events"].clear()
    if not n or split_secs == 0.0:
        return None
    peak = PEAK_BF16_MFMA_TFLOPS / 6
    return {"bound": "mfma", "kernel": "conv engine inside the evaluation passes of a frame (forward only, frozen weights)",
            "achieved": round(split_flops / split_secs / 1e12, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(split_flops / split_secs / 1e12 / peak, 4), "launches_per_frame": n // steps,
            "conv_ms_per_frame": round(secs / steps * 1e3, 3), "algorithmic_gflop_per_frame": round(flops / steps / 1e9, 1),
            "all_conv_tflops": round(flops / secs / 1e12, 2)}


def conv_roofline(runner, data, steps=2, default_workload=False):
    """Live measurement of the conv engine with HIP events on the launch stream: algorithmic (fp32-equivalent, unpadded) FLOPs
    of every launch / its event-timed duration, per kernel kind, each priced against the peak of the arithmetic it runs:
      *_split : the main implicit-GEMM kernel (forward / data gradient) and the halo-tile weight gradient -- every fp32 operand
                as three bf16 pieces, six piece products per multiply on v_mfma_f32_32x32x16_bf16 => peak = dense bf16 MFMA / 6;
      others  : 16-column kernels, 16-channel K chunks and the remaining weight gradients on the f32 MFMA.
    The `roofline` object is the dominant kind's (most time); `by_kind` and `all_conv` carry the rest."""
    from codeps_amd import functional as F
    F._profile["events"].clear()
    F._profile["on"] = True
    # per-launch durations are taken with the branches SERIALISED (no side streams): with them, a launch's events also span the
    # time it shares the chip with the other branch's kernels, which is a property of the schedule, not of the kernel -- and it
    # is what `rocprofv3 --kernel-trace` reports too (the tracer serialises dispatches)
    streams = (getattr(runner.net, "pose_stream", None), getattr(runner.net, "branch_stream", None))
    if streams[0] is not None or streams[1] is not None:
        runner.net.pose_stream = runner.net.branch_stream = None
    for _ in range(steps):
        runner.step(data)
    torch.cuda.synchronize()
    F._profile["on"] = False
    if streams[0] is not None or streams[1] is not None:
        runner.net.pose_stream, runner.net.branch_stream = streams
    by, photo = {}, {}
    for kind, flops, a, b, _tag in F._profile["events"]:
        if kind.startswith("photo_"):                     # HBM-bound loss kernels: `flops` carries algorithmic BYTES here
            d = photo.setdefault(kind, [0.0, 0.0, 0])
            d[0] += flops
            d[1] += a.elapsed_time(b) * 1e-3
            d[2] += 1
            continue
        d = by.setdefault(kind, [0.0, 0.0, 0])
        d[0] += flops
        d[1] += a.elapsed_time(b) * 1e-3
        d[2] += 1
    F._profile["events"].clear()
    flops = sum(v[0] for v in by.values())
    secs = sum(v[1] for v in by.values())
    n = sum(v[2] for v in by.values())
    # kinds: *_f16x2 = two fp16 pieces, three products per multiply on the f16 MFMA (same rate as bf16): peak = dense 16-bit MFMA / 3;
    #        *_split = three bf16 pieces, six products: / 6;  the rest: f32 MFMA
    peak_of = lambda k: (PEAK_BF16_MFMA_TFLOPS / 3 if k.endswith("_f16x2") else PEAK_BF16_MFMA_TFLOPS / 6 if k.endswith("_split")
                         else PEAK_F32_MFMA_TFLOPS)
    detail = {k: {"launches_per_step": v[2] // steps, "tflops": round(v[0] / v[1] / 1e12, 2), "ms_per_step": round(v[1] / steps * 1e3, 3),
                  "peak": round(peak_of(k), 1), "frac": round(v[0] / v[1] / 1e12 / peak_of(k), 4)} for k, v in by.items()}
    # dominant kernel: the split-operand implicit GEMM (forward + data gradient) when it runs, else everything on the f32 MFMA
    groups = {sfx: [k for k in by if k.endswith(sfx)] for sfx in ("_f16x2", "_split")}
    sfx = max(groups, key=lambda g: sum(by[k][1] for k in groups[g]))      # the arithmetic most of the conv time runs on
    dom = groups[sfx]
    products = 3 if sfx == "_f16x2" else 6
    if dom:
        dflops, dsecs, dn = (sum(by[k][i] for k in dom) for i in range(3))
        peak = PEAK_BF16_MFMA_TFLOPS / products
        name = ("conv_halo3 / conv_halo3r / conv_igemm / chain_expand_reduce_h fwd + dgrad, wgrad_halo64s: fp32 operands as two fp16 pieces of x 2^k "
                "(block-local power-of-two scales), three products on v_mfma_f32_16x16x32_f16 / 32x32x16_f16, fp32 accumulate (peak = dense 16-bit MFMA "
                f"{PEAK_BF16_MFMA_TFLOPS:.0f} / 3)") if sfx == "_f16x2" else (
                "conv_igemm_kernel fwd + dgrad, wgrad_halo64s_kernel: fp32 operands as 3 bf16 pieces, 6 products on "
                f"v_mfma_f32_32x32x16_bf16, fp32 accumulate (peak = dense bf16 MFMA {PEAK_BF16_MFMA_TFLOPS:.0f} / 6)")
    else:
        dflops, dsecs, dn, peak = flops, secs, n, PEAK_F32_MFMA_TFLOPS
        name = "conv_igemm_kernel + conv_wgrad_kernel (f32 MFMA 32x32x2)"
    ach = dflops / dsecs / 1e12
    traffic, traffic_src = pmc_traffic(default_workload)
    secondary = None
    if photo:
        pb, ps, pn = (sum(v[i] for v in photo.values()) for i in range(3))
        ptraffic = photometric_pmc(default_workload)
        secondary = {"bound": "hbm", "kernel": "photometric loss: pyramid_kernel + photo_fwd_kernel + reduce, photo_bwd_kernel + pose_reduce + depth_combine "
                                              "(depth-warp + bilinear sample + 3x3 SSIM + auto-mask, all 5 scales per launch)",
                     "achieved": round(pb / ps / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(pb / ps / 1e9 / PEAK_HBM_GBS, 4),
                     "algorithmic_mb_per_launch": round(pb / pn / 1e6, 2), "avg_launch_us": round(ps / pn * 1e6, 1),
                     "launches_per_step": pn // steps, "ms_per_step": round(ps / steps * 1e3, 3), "traffic": ptraffic,
                     "wave_cycle_shares": photometric_shares(),
                     "note": "84 B per pixel and batch item (40 forward + 44 backward, SURVEY.md 8d): achieved / frac are LIVE (HIP events of this "
                             "run).  `wave_cycle_shares` and `traffic` are read from the newest committed profiles/rNN_* file (a separate rocprofv3 "
                             "--pmc run of the same command, named in their `source` fields), not measured by this run; traffic = PMC HBM bytes per "
                             "launch incl. the warped samples and SSIM adjoint coefficients the forward keeps for the backward (60 B per pixel and "
                             "scale, written once and read once)"}
    # the vendor peak and the peak re-measured on this box (csrc/diag.hip: register-resident MFMA loop on operands
    # that change between consecutive MFMAs -- what the matrix pipe sustains under the power limit on real data -- and a stream
    # triad).  `peak` / `frac` stay on the vendor figure; the measured ones are reported beside them.
    measured = None
    try:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
        import peaks as _peaks
        m = _peaks.measure(sustain_ms=100.0)
        mpeak = m["mfma_f32_32x32x16_bf16_random_tflops"] / products if dom else m["mfma_f32_32x32x2_f32_random_tflops"]
        measured = {"mfma_bf16_zeros_tflops": m["mfma_f32_32x32x16_bf16_zeros_tflops"], "mfma_bf16_random_tflops": m["mfma_f32_32x32x16_bf16_random_tflops"],
                    "mfma_f32_random_tflops": m["mfma_f32_32x32x2_f32_random_tflops"], "hbm_triad_gbs": m["hbm_triad_gbs"], "hbm_copy_gbs": m.get("hbm_copy_gbs"),
                    "peak_of_this_arithmetic": round(mpeak, 1), "frac": round(ach / mpeak, 4),
                    "in_kernel_clock": clock_record(),
                    "note": "mfma_* / hbm_triad_gbs: LIVE micro-kernels on this GPU (no-memory MFMA loop; zeros = how vendor peaks are quoted, "
                            "random = operands changing between consecutive MFMAs; a[i] = b[i] + s c[i] over 3 x 1 GiB; hbm_copy_gbs: a 16-byte-vector copy of 1 GiB, the form the hardware guide measures its 6.29 TB/s on).  in_kernel_clock: cycles "
                            "per MFMA and the clock held, from the committed s_memtime / s_memrealtime stamp files named in its `source` (a "
                            "diagnostic build; not measured by this run)"}
    except Exception as e:                                  # noqa: BLE001  (diagnostics never fail the bench)
        print(f"[bench] measured peaks unavailable ({type(e).__name__}: {e})", file=sys.stderr)
    return {"bound": "mfma", "secondary": secondary, "kernel": name, "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(ach / peak, 4), "measured_peak": measured, "traffic": traffic, "traffic_source": traffic_src,
            "launches_per_step": dn // steps, "avg_launch_us": round(dsecs / dn * 1e6, 2),
            "all_conv": {"launches_per_step": n // steps, "tflops": round(flops / secs / 1e12, 2), "ms_per_step": round(secs / steps * 1e3, 3),
                         "algorithmic_gflop_per_step": round(flops / steps / 1e9, 1)},
            "conv_ms_per_step": round(secs / steps * 1e3, 3), "algorithmic_gflop_per_step": round(flops / steps / 1e9, 1),
            "by_kind": detail}


SECONDARY = (("configs1_pretrain", ["--mode", "train"]),
             ("configs3_slice", ["--per-gpu-batch", "4"]),
             ("configs4_slice", ["--mode", "train", "--encoder", "101", "--height", "1024", "--width", "2048", "--per-gpu-batch", "2", "--steps", "5",
                                 "--warmup", "2"]),
             ("frame", ["--mode", "frame"]))


def secondary_rows():
    """The other configurations BASELINE.json names that fit one GPU (configs[1], the per-GPU slices of configs[3] / configs[4]) and the
    frame mode, each measured by a CHILD process of this bench (a fresh GPU context each; never an exec from a process that has touched
    the GPU) with its own warm-up, graph capture, timed steps and conv roofline pass.  -> {name: {images/s, ms/step, roofline.frac, ...}}"""
    import subprocess
    out = {}
    for name, flags in SECONDARY:
        cmd = [sys.executable, os.path.abspath(__file__), *flags, "--no-cpu-baseline", "--no-secondary"]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
            d = json.loads(lines[-1])
            roof = d.get("roofline") or {}
            out[name] = {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                         "step_ms_median": d["step_ms"]["median"], "steps": d["steps"], "workload": d["config"]["workload"],
                         "launch": d["config"]["launch"], "peak_hbm_gb": d["config"]["peak_hbm_gb"],
                         "roofline_frac": roof.get("frac"), "roofline_tflops": roof.get("achieved"),
                         "conv_ms_per_step": roof.get("conv_ms_per_step"), "algorithmic_gflop_per_step": roof.get("algorithmic_gflop_per_step"),
                         "wall_s": round(time.perf_counter() - t0, 1)}
        except Exception as e:                              # noqa: BLE001  (a secondary row never fails the headline measurement)
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default="adapt", choices=["adapt", "train", "frame"],
                    help="adapt = the BASELINE metric (one adaptation step); train = pretrain step (configs[1]); frame = secondary "
                         "row: the per-frame evaluation passes of scripts/adapt_codeps.py:164-199 (codeps.frame.pre_step, EMA "
                         "teacher, shared encoder features) + the adaptation step")
    ap.add_argument("--encoder", type=int, default=None, help="ResNet depth of the shared encoder (adapt: 101, train: 50)")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--per-gpu-batch", type=int, default=None)
    ap.add_argument("--graph", type=int, default=1, help="replay forward+backward from a captured HIP graph")
    ap.add_argument("--dist-backend", default=None, help="torch.distributed backend (default: nccl = RCCL); 'gloo' + "
                    "CODEPS_ALL_RANKS_ON_GPU0=1 lets the N>1 code path be exercised on a single-GPU box")
    ap.add_argument("--yaml-faithful", action="store_true",
                    help="secondary row: the adapt YAML as written (target / replay 1 + 2 items at 384x1408, source / mixup 2 + 2 at "
                         "512x1024, B = 7) instead of the all-512x1024 B = 8 batch the BASELINE metric is quoted on")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary rows (the other BASELINE configurations that fit one GPU + the frame mode), which the default "
                         "single-GPU run measures in child processes BEFORE its own step and reports under `secondary`")
    args = ap.parse_args()
    secondary = None
    if (args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_secondary and args.mode == "adapt" and args.encoder is None
            and args.per_gpu_batch is None and (args.height, args.width) == (512, 1024) and not args.yaml_faithful and args.graph):
        secondary = secondary_rows()        # before this process touches the GPU: one context at a time

    from codeps_amd import _lib
    from codeps_amd.distributed import init_from_env
    from codeps_amd.step import StepRunner
    frame_mode = args.mode == "frame"
    if frame_mode:
        args.mode = "adapt"
    rank, local, world = init_from_env(args.dist_backend)
    if os.environ.get("CODEPS_ALL_RANKS_ON_GPU0") == "1":
        local = 0
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: codeps_amd has no CPU path")
    _lib.lib()                                           # fail loudly if the HIP library is missing
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    enc = args.encoder or (101 if args.mode == "adapt" else 50)
    per_gpu = args.per_gpu_batch or (8 if args.mode == "adapt" else 4)
    net, cfg = build(device, enc, args.height, args.width, args.mode, (384, 1408) if args.yaml_faithful else None)
    data, n_images = make_data(args.mode, args.height, args.width, device, rank, per_gpu, args.yaml_faithful, world)
    runner = StepRunner(net, cfg, args.mode, world)
    if world > 1 and args.mode == "adapt" and per_gpu == 4 and not args.yaml_faithful:
        # BASELINE configs[3]: the ranks hold different numbers of target / replay items -> count-aware SyncBN and loss means
        splits = [adapt_split(per_gpu, r, world) for r in range(world)]
        runner.set_key_counts({k: [sp[i] for sp in splits] for i, k in enumerate(("target", "target_replay", "source", "cut_mixup"))}, rank)
    if world > 1:
        from codeps_amd.distributed import SyncBatchNormReducer
        red = SyncBatchNormReducer()
        for enc_mod in (net.backbone_po_depth, net.backbone_pose_sflow):
            enc_mod.sync_bn = red
        if net.instance_head is not None:
            net.instance_head.sync_bn = red

    for _ in range(max(args.warmup, 1)):
        runner.step(data)
    torch.cuda.synchronize()

    graph = None
    if args.graph and world == 1:
        try:
            # StepRunner.capture bumps the weight-cache epoch right before the capture: the pack kernels of the trainable
            # weights are recorded into the graph, so a replay is a real step on the weights Adam has just written
            runner.capture(data)
            graph = runner
            runner.replay()
            torch.cuda.synchronize()
        except Exception as e:                            # noqa: BLE001  (eager is always valid)
            print(f"[bench] graph capture unavailable ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

    ema = None
    if frame_mode:
        import copy
        from codeps_amd import synthetic
        from codeps_amd.codeps import pre_step
        from codeps_amd.datasets import Mixup
        ema = copy.deepcopy(net)                            # scripts/adapt_codeps.py:146 (CodepsNet.ema_model)
        th, tw = data["target"]["rgb"][0].shape[-2:]
        raw_mix = synthetic.mixup_raw_item(data["cut_mixup"]["rgb"][0].shape[0], args.height, args.width, th, tw, device, 4321 + rank)
        from codeps_amd.datasets import DiversityReplayBuffer
        # cfg/adapt_cityscapes_kitti_360.yaml:26-30: target_size 300, maximize_diversity, similarity_threshold 0.95
        replay = DiversityReplayBuffer(100000, 300, 0.95, device, feature_dim=int(net.backbone_po_depth.num_ch_enc[-1]))
        frame_no = [0]

    def eval_passes():
        sample = {k: (dict(v) if isinstance(v, dict) else v) for k, v in data.items() if not k.endswith("mixup")}
        sample["cut_mixup"] = dict(raw_mix)
        pre_step(net, ema, ["semantic_head"], sample, [0, -1, 1], True)
        return sample

    def one_step():
        if frame_mode:
            # per incoming frame: teacher / student evaluation passes (scripts/adapt_codeps.py:164-199), then the mix-up item
            # is built on the device (codeps/online_adap.py:331-335 -> Mixup.do_mixup) into the step's static input buffers
            sample = eval_passes()
            mixed = Mixup.do_mixup("cut_mixup", sample["cut_mixup"], net.instance_algo, net.cfg_mixup)
            data["cut_mixup"]["rgb"][0].copy_(mixed["rgb"][0])
            data["cut_mixup"]["semantic"].copy_(mixed["semantic"])
        if graph is not None:
            result = runner.replay()[2]
        else:
            result = runner.step(data)[2]
        if frame_mode:                                      # scripts/adapt_codeps.py:307: the frame enters the replay buffer
            replay.add(frame_no[0], result["image_features"][:1])
            frame_no[0] += 1
            if frame_no[0] % 8 == 0:
                replay.sync()                               # the data loader reads buffer_indices when it draws replay samples

    if frame_mode:
        one_step()

    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    red_before = (red.n_gather, red.n_reduce) if world > 1 else (0, 0)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]     # per-step times (enqueue-only: no sync inside the timed region)
    t0 = time.perf_counter()
    for i in range(args.steps):
        marks[i].record()
        one_step()
    marks[args.steps].record()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    comm = None
    if world > 1:
        # what the collectives of this run were: proof that the backend saw `world` ranks, the flat gradient all-reduce timed on its
        # own with HIP events (average of 5 after one warm-up; it sits between backward and Adam in the step), SyncBN traffic per step
        buf = runner.grad_all
        torch.distributed.all_reduce(buf.clone())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tmp = buf.clone()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            torch.distributed.all_reduce(tmp)
        e1.record()
        torch.cuda.synchronize()
        comm = {"backend": torch.distributed.get_backend(), "ranks_seen": torch.distributed.get_world_size(),
                "grad_allreduce_mb": round(buf.numel() * 4 / 1e6, 1), "grad_allreduce_ms": round(e0.elapsed_time(e1) / 5, 3),
                "syncbn_allgathers_per_step": round((red.n_gather - red_before[0]) / args.steps, 1),
                "syncbn_allreduces_per_step": round((red.n_reduce - red_before[1]) / args.steps, 1),
                "launch": "eager (the SyncBN collectives sit inside forward / backward)",
                # the gradient collective as the step issues it (codeps_amd.step): the heads' front of the flat buffer from inside the
                # backward pass on its own stream, the encoders' back after it -- or one collective (CODEPS_AR_OVERLAP=0, unequal shards)
                "grad_allreduce_segments": {"front_mb": round(runner._early_end * 4 / 1e6, 1),
                                            "back_mb": round((buf.numel() - runner._early_end) * 4 / 1e6, 1),
                                            "steps_with_the_front_overlapped": runner.ar_overlap_steps,
                                            "overlap": "front segment all-reduced on a second stream while the encoders' backward pass runs"
                                                       if runner._early_end else "none (single collective between backward and Adam)"}}
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())

    roof = None
    if not args.no_roofline:            # every rank runs the profiled steps (they contain the SyncBN / gradient collectives); rank 0 reports
        default_wl = (args.mode == "adapt" and (args.encoder in (None, 101)) and (args.height, args.width) == (512, 1024)
                      and args.per_gpu_batch in (None, 8) and not args.yaml_faithful)
        roof = conv_roofline(runner, data, default_workload=default_wl)
        if frame_mode and roof is not None:
            roof["eval_passes"] = eval_pass_roofline(eval_passes)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.mode == "adapt":
        cpu = cpu_baseline(net, args.height, args.width)

    if rank == 0:
        total_images = n_images * world * args.steps
        if args.mode == "adapt":
            comp = " + ".join(f"{v['rgb'][0].shape[0]} {k} @ {v['rgb'][0].shape[-2]}x{v['rgb'][0].shape[-1]}" for k, v in data.items())
            which = ("BASELINE configs[2], YAML-faithful secondary row" if args.yaml_faithful else
                     "BASELINE configs[3] (per-rank slice of the B=32 / 8-GPU batch)" if per_gpu == 4 else "BASELINE configs[2]")
            workload = (f"{which}: online-adapt step cfg/adapt_cityscapes_kitti_360.yaml, per-GPU batch {n_images} = {comp}, "
                        f"ResNet-{enc} encoder frozen, trainable pose encoder + depth/pose/semantic heads, EMA semantic head")
        else:
            which = ("BASELINE configs[4] (per-GPU slice of B=16 / 8 GPUs)" if (enc == 101 and (args.height, args.width) == (1024, 2048))
                     else "BASELINE configs[1]")
            workload = f"{which}: pretrain step ResNet-{enc} + 4 heads @ {args.height}x{args.width}, per-GPU batch {n_images}"
        line = {
            "metric": ("adapt-frame images/sec (evaluation passes + step)" if frame_mode else "adapt-step images/sec")
                      if args.mode == "adapt" else "pretrain-step images/sec",
            "value": round(total_images / dt, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # `value` is the mean over the timed region (the contract); the spread of the individual steps (HIP events, this rank):
            "step_ms": {"median": round(per_step[len(per_step) // 2], 3), "min": round(per_step[0], 3), "max": round(per_step[-1], 3),
                        "images_per_sec_at_median": round(n_images * world / (per_step[len(per_step) // 2] * 1e-3), 2)},
            "dtype": "f32", "data": "synthetic",
            "arithmetic": {
                "16": ("fp32 tensors end to end.  Default since round 6 (CODEPS_SPLIT=16): the 3x3 / 1x1 forward, data-gradient and 3x3 weight-gradient "
                       "kernels feed the matrix cores every fp32 operand as TWO fp16 pieces of x 2^k (hi = rn16(x 2^k), lo = rn16(x 2^k - hi): "
                       "|x 2^k - hi - lo| <= 2^-22 |x 2^k|) and sum the THREE products hi hi + hi lo + lo hi on v_mfma_f32_*_f16 with fp32 "
                       "accumulation; 2^k is a power of two chosen per weight tensor by the pack and per block (chain kernel: per pixel) for "
                       "activations / gradients, so scaling is exact and no tensor overflows fp16.  Error against float64 measured per kernel at or "
                       "below the three-bf16-piece kernels' and within 3x of torch's own fp32 convolution (profiles/r06_f16_stage_*.json, "
                       "tests/test_precision_gpu.py).  <= 32-channel weight gradients, 16-column 1x1 "
                       "kernels, the last up-convolution and the disparity head run on the f32 MFMA"),
                "3": ("fp32 end to end; conv forward / data-gradient / halo-tile and 1x1 weight-gradient operands enter the matrix cores as "
                      "exact 3-way bf16 splits (six bf16 products per multiply, fp32 accumulate: product error ~6e-9 relative, below "
                      "fp32 rounding); the <= 32-channel weight gradients, the 1x1 16-column kernels, the last up-convolution and the "
                      "1-channel disparity head on the f32 MFMA"),
            }.get(os.environ.get("CODEPS_SPLIT", "16"), "CODEPS_SPLIT=" + os.environ.get("CODEPS_SPLIT", "16") + " (non-default arithmetic, see codeps_amd/csrc/conv.hip)"),
            "config": {"workload": workload,
                       "global_batch": n_images * world, "parallelism": f"dp{world}", "launch": "hip-graph" if graph is not None else "eager",
                       "grad_allreduce_mb": round(runner.grad_bytes / 1e6, 1),
                       "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 1e9, 2)},
        }
        if roof is not None:
            sec = roof.pop("secondary", None)
            line["roofline"] = roof
            if sec is not None:
                line["roofline_hbm"] = sec           # the HBM-bound half of north_star: the photometric pipeline
            ev = roof.pop("eval_passes", None)
            if ev is not None:
                line["roofline_eval_passes"] = ev    # --mode frame: the conv engine inside the per-frame evaluation passes
        if cpu is not None:
            line["cpu_baseline"] = cpu
        if comm is not None:
            line["comm"] = comm
        if secondary is not None:
            line["secondary"] = secondary
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
