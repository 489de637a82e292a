"""Two ranks of bench.py on ONE GPU (gloo for the collectives): the N > 1 code path of the step -- sharded batch, SyncBatchNorm
statistics exchange through codeps_bn_sync_combine, flat-gradient all-reduce, the rank-collective roofline pass -- has to run to
completion and report a finite rate.  (RCCL itself needs one GPU per rank; the driver exercises it at round end.)"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_worker(out_file, world, extra=(), tile_stats=False, env=None):
    """tile_stats=False (the exactness tests): two switches take both sides back to arithmetic whose bits do not depend on how the batch
    is cut -- CODEPS_EPI_STATS=0: BatchNorm's batch statistics come from the pass over the tensor (the round-5 default takes them from
    per-tile partials of the producing convolution where the launch plan allows, and which layers that is depends on the per-process
    batch); CODEPS_SPLIT=3: the three-bf16-piece split, exact per product (the round-6 default scales fp16 pieces per block, and a weight
    gradient's running scale sees the tiles of the images ITS process holds).  Either default moves a few results in their last bits between
    a single-process and a sharded run -- enough to put one ReLU with a ~1e-7 pre-activation on the other side of zero at 64 x 128 and move
    a flat gradient by 1e-4 .. 1e-3 of its norm, which says nothing about the SyncBatchNorm / all-reduce algebra these tests pin to 1e-5.
    test_sharded_step_with_tile_statistics runs both defaults.  The switches travel in the children's environment only."""
    worker = os.path.join(ROOT, "tests", "dist_equiv_worker.py")
    over = {} if tile_stats else {"CODEPS_EPI_STATS": "0", "CODEPS_SPLIT": "3"}
    over.update(env or {})
    return _run_worker_env(worker, out_file, world, extra, over)


def _run_worker_env(worker, out_file, world, extra, over):
    base = {k: v for k, v in os.environ.items() if k not in ("CODEPS_EPI_STATS", "CODEPS_SPLIT") or not over}
    base.update(over)
    if world == 1:
        env = {k: v for k, v in base.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        out = subprocess.run([sys.executable, worker, out_file, *extra], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
        return
    port = str(_free_port())
    procs = []
    for r in range(world):
        env = dict(base, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, worker, out_file.replace(".pt", f"_r{r}.pt"), *extra], cwd=ROOT, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        _o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-3000:]


def _compare(ref, shards):
    r0 = shards[0]
    for r in shards[1:]:
        for a, b in zip(r0["grads"] + r0["params"], r["grads"] + r["params"]):
            assert torch.equal(a, b), "ranks disagree after the all-reduce"
    for k, v in ref["bufs"].items():
        if v.is_floating_point():
            err = float((r0["bufs"][k].double() - v.double()).abs().max() / (v.double().abs().max() + 1e-12))
            assert err < 1e-5, f"SyncBN running statistic {k}: {err:.2e}"
            for r in shards[1:]:
                assert torch.equal(r0["bufs"][k], r["bufs"][k])
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))      # noqa: E731
    for gi, (g, gr) in enumerate(zip(r0["grads"], ref["grads"])):
        assert rel(g, gr) < 1e-5, f"flat gradient group {gi}: rel-L2 {rel(g, gr):.2e}"
        off, worst = 0, 0.0
        big = max(float(gr.abs().max()), 1e-30)
        for shape in ref["names"][gi]:
            n = 1
            for d in shape:
                n *= d
            a, b = g[off:off + n], gr[off:off + n]
            worst = max(worst, float((a.double() - b.double()).norm()) / (float(b.double().norm()) + 1e-4 * big * n ** 0.5))
            off += n
        assert worst < 1e-4, f"group {gi}: worst per-parameter gradient error {worst:.2e}"
    for gi, (p, pr) in enumerate(zip(r0["params"], ref["params"])):
        assert float((p - pr).abs().max()) <= 2.05e-4, "one Adam step moves a weight by at most lr = 1e-4"
        assert float(((p - pr).abs() > 1e-7).float().mean()) < 0.05, "more than 5 % of the weights stepped differently"


def test_configs3_eight_rank_layout_equals_single_process_step(tmp_path):
    """BASELINE configs[3] (SURVEY.md 8d "Config 4") with its REAL 8-rank layout: global batch 32 = 4 target + 12 replay +
    8 source + 8 mixup items; ranks 0-3 hold 1 target + 1 replay, ranks 4-7 hold 2 replay items, every rank 1 source + 1 mixup
    item.  Eight processes on GPU 0 over gloo (32 x 64 items, ResNet-18 trunk) against ONE process on the concatenated B = 32
    batch: count-aware SyncBN, the global count-weighted depth means, the broadcast car-not-moving flag and the single flat
    gradient all-reduce have to reproduce the single-process gradients (DDP + SyncBN semantics, misc/train_utils.py:37-59)."""
    single = str(tmp_path / "single.pt")
    _run_worker(single, 1, ("--configs3",))
    _run_worker(str(tmp_path / "shard.pt"), 8, ("--configs3",))
    _compare(torch.load(single), [torch.load(str(tmp_path / f"shard_r{r}.pt")) for r in range(8)])


@pytest.mark.parametrize("asym", [False, True])
def test_sharded_step_equals_single_process_step(tmp_path, asym):
    """DDP + SyncBatchNorm semantics of misc/train_utils.py:47-59 on the REAL adaptation step: two ranks, each with its slice of
    every key (asym: unequal slices -- rank 0 holds both target items, rank 1 three of the four replay items, the layout idea of
    BASELINE configs[3]), against one process on the concatenated batch: the all-reduced flat gradients (x 1 / world), the
    parameters after the fused Adam step and the pose encoder's SyncBN running statistics agree; both ranks end bit-identical."""
    extra = ("--asym",) if asym else ()
    single = str(tmp_path / "single.pt")
    _run_worker(single, 1, extra)
    _run_worker(str(tmp_path / "shard.pt"), 2, extra)
    ref = torch.load(single)
    r0, r1 = torch.load(str(tmp_path / "shard_r0.pt")), torch.load(str(tmp_path / "shard_r1.pt"))
    for a, b in zip(r0["grads"] + r0["params"], r1["grads"] + r1["params"]):
        assert torch.equal(a, b), "ranks disagree after the all-reduce"
    for k, v in ref["bufs"].items():
        if v.is_floating_point():
            err = float((r0["bufs"][k].double() - v.double()).abs().max() / (v.double().abs().max() + 1e-12))
            assert err < 1e-5, f"SyncBN running statistic {k}: {err:.2e}"
            assert torch.equal(r0["bufs"][k], r1["bufs"][k])
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))      # noqa: E731
    for gi, (g, gr) in enumerate(zip(r0["grads"], ref["grads"])):
        # whole flat buffer, then parameter by parameter (a few summation orders differ: per-rank conv tiles, the order of the
        # all-reduce): 1e-5 of the buffer's norm, 1e-4 per parameter against its own norm with a floor at 1e-6 of the largest
        assert rel(g, gr) < 1e-5, f"flat gradient group {gi}: rel-L2 {rel(g, gr):.2e}"
        off, worst = 0, 0.0
        floor = 1e-6 * max(float(gr[o:o + 1].abs().max()) for o in range(1)) + 1e-30
        big = max(float(gr.abs().max()), 1e-30)
        for shape in ref["names"][gi]:
            n = 1
            for d in shape:
                n *= d
            a, b = g[off:off + n], gr[off:off + n]
            worst = max(worst, float((a.double() - b.double()).norm()) / (float(b.double().norm()) + 1e-4 * big * n ** 0.5))
            off += n
        assert worst < 1e-4, f"group {gi}: worst per-parameter gradient error {worst:.2e}"
        del floor
    for gi, (p, pr) in enumerate(zip(r0["params"], ref["params"])):
        # the first Adam step moves every weight by lr * sign(g) (m / sqrt(v) = +-1): weights whose gradient is rounding noise
        # around zero may step the other way, so the bound is 2 lr per element and agreement on the bulk
        assert float((p - pr).abs().max()) <= 2.05e-4, "one Adam step moves a weight by at most lr = 1e-4"
        assert float(((p - pr).abs() > 1e-7).float().mean()) < 0.05, "more than 5 % of the weights stepped differently"


@pytest.mark.parametrize("mixed", [False, True])
def test_segmented_allreduce_waits_for_every_firing_of_a_marked_layer(tmp_path, mixed):
    """CODEPS_AR_OVERLAP=1 (opt-in): the heads' front of the flat buffer is all-reduced from inside the backward pass.  With the source items
    at another resolution than the target's (the shipped yamls: 512 x 1024 vs 384 x 1408) every head runs twice per step and its marked
    layer's backward node fires twice: the collective must wait for the LAST firing (round 5 counted heads, not firings, and sent a front
    that the second firing was still accumulating into -- ADVICE r05).  Same exactness bars as the single-collective tests."""
    extra = ("--mixed-res",) if mixed else ()
    single = str(tmp_path / "single.pt")
    _run_worker(single, 1, extra)
    _run_worker(str(tmp_path / "shard.pt"), 2, extra, env={"CODEPS_AR_OVERLAP": "1"})
    shards = [torch.load(str(tmp_path / f"shard_r{r}.pt")) for r in range(2)]
    assert all(sh["overlapped"] == 1 for sh in shards), "the front segment did not go out from inside the backward pass"
    # three marked heads (depth, semantic, pose); the pose head runs once per frame pair, depth / semantic once per resolution
    n_fire = shards[0]["expected_firings"]
    assert (n_fire > 3) if mixed else (n_fire == 3), n_fire
    _compare(torch.load(single), shards)


def test_sharded_step_with_tile_statistics(tmp_path):
    """The unequal two-rank layout with the shipped defaults (BatchNorm statistics from the convolutions' epilogues where the plan
    allows, the two-fp16-piece arithmetic with its block-local scales, on both sides): ranks bit-identical, SyncBN running statistics to 1e-5, flat gradients to 5e-3 of their norm -- the bound
    that survives a ReLU flipped by last-bit differences of a statistic (see _run_worker)."""
    single = str(tmp_path / "single.pt")
    _run_worker(single, 1, ("--asym",), tile_stats=True)
    _run_worker(str(tmp_path / "shard.pt"), 2, ("--asym",), tile_stats=True)
    ref = torch.load(single)
    r0, r1 = torch.load(str(tmp_path / "shard_r0.pt")), torch.load(str(tmp_path / "shard_r1.pt"))
    for a, b in zip(r0["grads"] + r0["params"], r1["grads"] + r1["params"]):
        assert torch.equal(a, b), "ranks disagree after the all-reduce"
    for k, v in ref["bufs"].items():
        if v.is_floating_point():
            err = float((r0["bufs"][k].double() - v.double()).abs().max() / (v.double().abs().max() + 1e-12))
            assert err < 1e-5, f"SyncBN running statistic {k}: {err:.2e}"
    for gi, (g, gr) in enumerate(zip(r0["grads"], ref["grads"])):
        rel = float((g.double() - gr.double()).norm() / (gr.double().norm() + 1e-30))
        assert rel < 5e-3, f"flat gradient group {gi}: rel-L2 {rel:.2e}"


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank (dormant on the one-GPU boxes of this pool)")
def test_sharded_step_over_rccl_equals_single_process_step(tmp_path):
    """The same equivalence with backend "nccl" (= RCCL), one rank per device: the gradient all-reduce and the SyncBN collectives
    over xGMI.  Skipped where a single GPU is visible; live on any box with two or more."""
    single = str(tmp_path / "single.pt")
    _run_worker(single, 1, ())
    _run_worker(str(tmp_path / "shard.pt"), 2, ("--rccl",))
    _compare(torch.load(single), [torch.load(str(tmp_path / f"shard_r{r}.pt")) for r in range(2)])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank (dormant on the one-GPU boxes of this pool)")
def test_bench_over_rccl_reports_the_collectives():
    """`bench.py --gpus 2` as the driver launches it (RCCL): the line must prove that the backend saw both ranks and time the
    flat gradient all-reduce."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--encoder", "18", "--height", "128", "--width", "256", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["comm"]["backend"] == "nccl" and d["comm"]["ranks_seen"] == 2 and d["comm"]["grad_allreduce_ms"] > 0


def test_two_ranks_on_one_gpu_complete_with_default_flags():
    env = dict(os.environ, CODEPS_ALL_RANKS_ON_GPU0="1", HSA_ENABLE_IPC_MODE_LEGACY="0", CODEPS_AR_OVERLAP="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dist-backend", "gloo",
           "--encoder", "18", "--height", "128", "--width", "256", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["launch"] == "eager"
    assert d["value"] > 0 and d["value"] == d["value"]
    assert d["roofline"]["achieved"] > 0
    # the collectives of the run, as the line reports them: both ranks seen, the flat gradient all-reduce timed, one all-gather and one
    # all-reduce per train-mode BatchNorm layer of the pose encoder (ResNet-18: 20 layers) per step
    c = d["comm"]
    assert c["backend"] == "gloo" and c["ranks_seen"] == 2 and c["grad_allreduce_ms"] > 0
    assert c["syncbn_allgathers_per_step"] == 20 and c["syncbn_allreduces_per_step"] == 20
    assert d["step_ms"]["min"] <= d["step_ms"]["median"] <= d["step_ms"]["max"]
    # CODEPS_AR_OVERLAP=1 (opt-in; the default is ONE collective): the gradient collective ran in two segments in every timed step -- the
    # heads' front of the flat buffer from inside the backward pass, the pose encoder's back after it
    sg = c["grad_allreduce_segments"]
    assert sg["front_mb"] > 0 and sg["back_mb"] > 0 and abs(sg["front_mb"] + sg["back_mb"] - c["grad_allreduce_mb"]) < 0.2
    assert sg["steps_with_the_front_overlapped"] >= d["steps"]


def test_two_ranks_pretraining_step_with_sync_batchnorm_and_deferred_apply():
    """`--mode train` (BASELINE configs[1] / configs[4]: every BatchNorm in train mode, SyncBatchNorm across the ranks,
    misc/train_utils.py:50-51) on two gloo ranks: the bottlenecks' bn2 and the instance head's depthwise stages leave their apply pass to
    the 1x1 consumer (F.batch_norm_train(defer=True)) while their statistics cross the ranks -- the step completes with finite losses on
    both settings of CODEPS_BN_DEFER and reports the same collectives."""
    res = {}
    for defer in ("1", "0"):
        env = dict(os.environ, CODEPS_ALL_RANKS_ON_GPU0="1", HSA_ENABLE_IPC_MODE_LEGACY="0", CODEPS_BN_DEFER=defer)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "train", "--steps", "2", "--warmup", "1",
               "--dist-backend", "gloo", "--encoder", "50", "--height", "128", "--width", "256", "--per-gpu-batch", "2", "--no-cpu-baseline",
               "--no-roofline", "--no-secondary"]
        out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert d["n_gpus"] == 2 and d["value"] > 0 and d["value"] == d["value"]
        res[defer] = d["comm"]
    assert res["1"]["syncbn_allgathers_per_step"] == res["0"]["syncbn_allgathers_per_step"] > 50       # ResNet-50's 53 layers + the heads'
    assert res["1"]["syncbn_allreduces_per_step"] == res["0"]["syncbn_allreduces_per_step"]

