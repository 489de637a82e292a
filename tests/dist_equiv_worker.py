"""Worker of tests/test_distributed_gpu.py::test_sharded_step_equals_single_process_step.

    python tests/dist_equiv_worker.py <out_file> [--asym | --configs3]

world == 1 (no torchrun env): the adaptation step on the WHOLE batch.  world == 2 (RANK / WORLD_SIZE / MASTER_* set by the
test): every rank builds the identical whole batch, keeps its slice of every key, and runs the sharded step -- SyncBatchNorm
statistics exchange in the train-mode pose encoder, ONE all-reduce of each flat gradient buffer, 1 / world in the optimizer
(codeps_amd/distributed.py, codeps_amd/step.py; DDP + SyncBN semantics of misc/train_utils.py:47-59).  Both ranks sit on
GPU 0 and talk over gloo (RCCL needs one GPU per rank).  Top-k hard-pixel mining is rank-local by design, so
the comparison runs with top_k = 1.0; the tie-break noise is given explicitly, sliced like the items."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

H, W = 64, 128
ITEMS = {"target": 2, "target_replay": 4, "source": 2, "cut_mixup": 2}
# --configs3: BASELINE configs[3] / SURVEY.md 8d "Config 4" exactly -- global batch 32 over 8 ranks, ranks 0-3 hold 1 target +
# 1 replay item, ranks 4-7 hold 2 replay items, every rank 1 source + 1 mixup item -- at 32 x 64 so that eight processes fit one GPU
H3, W3 = 32, 64
ITEMS3 = {"target": 4, "target_replay": 12, "source": 8, "cut_mixup": 8}


def split_configs3(rank, world):
    if world == 1:
        return {k: (0, n) for k, n in ITEMS3.items()}
    assert world == 8
    if rank < 4:
        return {"target": (rank, rank + 1), "target_replay": (rank, rank + 1), "source": (rank, rank + 1), "cut_mixup": (rank, rank + 1)}
    lo = 4 + 2 * (rank - 4)
    return {"target": (0, 0), "target_replay": (lo, lo + 2), "source": (rank, rank + 1), "cut_mixup": (rank, rank + 1)}


def split_of(rank, world, asym):
    """{key: (lo, hi)} of this rank.  Symmetric: equal contiguous slices.  asym (BASELINE configs[3]): rank 0 holds
    the 2 target items' first + ... -- here: rank 0 = 1 target + 1 replay, rank 1 = 2 replay would leave a target item out, so
    the asymmetric layout of this test is target (2, 0), replay (1, 3): 3 depth items per rank, source / mixup 1 + 1."""
    if world == 1:
        return {k: (0, n) for k, n in ITEMS.items()}
    if not asym:
        return {k: (rank * n // world, (rank + 1) * n // world) for k, n in ITEMS.items()}
    lay = {"target": [(0, 2), (2, 2)], "target_replay": [(0, 1), (1, 4)], "source": [(0, 1), (1, 2)], "cut_mixup": [(0, 1), (1, 2)]}
    return {k: v[rank] for k, v in lay.items()}


def main():
    global H, W, ITEMS
    out_file = sys.argv[1]
    asym = "--asym" in sys.argv
    c3 = "--configs3" in sys.argv
    if c3:
        H, W, ITEMS, asym = H3, W3, ITEMS3, True
    from codeps_amd import synthetic
    from codeps_amd.codeps import gen_models
    from codeps_amd.config import get_cfg_shipped
    from codeps_amd.distributed import SyncBatchNormReducer, init_from_env
    from codeps_amd.step import StepRunner
    # default: every rank on GPU 0, collectives over gloo (one-GPU boxes).  --rccl: one GPU per rank, backend "nccl" = RCCL
    rccl = "--rccl" in sys.argv
    rank, local, world = init_from_env("nccl" if rccl else "gloo")
    dev = torch.device("cuda", local if rccl else 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(20)
    cfg = get_cfg_shipped(adapt=True)
    cfg.model.po_depth_net.params.nof_layers = 18
    cfg.dataset.feed_img_size = [H, W]
    # --mixed-res: source items at another resolution than the target's, as in the shipped yamls (512 x 1024 vs 384 x 1408): every head runs
    # once per resolution, so a marked layer's backward node fires twice per step (codeps_amd.step, segmented all-reduce)
    HS, WS = (H, W + 32) if "--mixed-res" in sys.argv else (H, W)
    cfg.adapt.source_dataset.feed_img_size = [HS, WS]
    cfg.semantics.top_k = 1.0
    net = gen_models(cfg, dev, list(range(9)), [9, 10, 11, 12, 13], [], adaptation_mode=True)
    gen = torch.Generator(device=dev).manual_seed(7)
    full = {"target": synthetic.triplet_item(ITEMS["target"], H, W, gen, dev),
            "target_replay": synthetic.triplet_item(ITEMS["target_replay"], H, W, gen, dev),
            "source": synthetic.triplet_item(ITEMS["source"], HS, WS, gen, dev, labels=True)}
    mix = synthetic.triplet_item(ITEMS["cut_mixup"], H, W, gen, dev, labels=True)
    full["cut_mixup"] = {"rgb": {0: mix["rgb"][0]}, "semantic": mix["semantic"]}
    torch.manual_seed(1)
    noise = {k: [torch.randn(ITEMS[k], 2, (HS if k == "source" else H) >> s, (WS if k == "source" else W) >> s) * 1e-5 for s in range(5)]
             for k in ("source", "target", "target_replay")}

    split = (lambda r: split_configs3(r, world)) if c3 else (lambda r: split_of(r, world, asym))
    sl = split(rank)

    def cut(v, lo, hi):
        if isinstance(v, dict):
            return {k: cut(x, lo, hi) for k, x in v.items()}
        return v[lo:hi].contiguous()

    data = {k: cut(v, *sl[k]) for k, v in full.items() if sl[k][1] > sl[k][0]}
    flat = lambda k: torch.cat([n[sl[k][0]:sl[k][1]].reshape(-1) for n in noise[k]]).to(dev)      # noqa: E731
    net.depth_algo.reconstruction_loss_adapt_source.noise = [flat("source")] if "source" in data else []
    net.depth_algo.reconstruction_loss.noise = [flat(k) for k in ("target", "target_replay") if k in data]
    runner = StepRunner(net, cfg, "adapt", world)
    if world > 1:
        counts = {k: [split(r)[k][1] - split(r)[k][0] for r in range(world)] for k in ITEMS}
        red = SyncBatchNormReducer()
        net.backbone_po_depth.sync_bn = net.backbone_pose_sflow.sync_bn = red
        if asym:
            runner.set_key_counts(counts, rank)
    total, losses, _ = runner.forward_backward(data)
    runner.allreduce_gradients()                                     # the step's gradient collective (the heads' front segment has been under
                                                                     # way since the backward pass issued the last kernel that writes it)
    grads = [(g.grad / world).cpu() for g in runner.groups]
    runner.step_count += 1
    from codeps_amd import functional as F
    for g in runner.groups:
        F.adam_ema_step(g.flat, g.grad, g.m, g.v, 1, lr=runner.lr, grad_scale=1.0 / world, ema_params=g.ema, ema_alpha=runner.ema_alpha)
    torch.cuda.synchronize()
    bufs = {n: b.cpu() for n, b in net.backbone_pose_sflow.named_buffers()}
    torch.save({"grads": grads, "bufs": bufs, "overlapped": runner.ar_overlap_steps, "expected_firings": runner._seg_expected, "losses": {k: (None if v is None else float(v)) for k, v in losses.items()},
                "params": [g.flat.cpu() for g in runner.groups], "names": [[tuple(p.shape) for p in g.params] for g in runner.groups]},
               out_file)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
