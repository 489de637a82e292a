"""N > 1 path on CPU: world-size-2 gloo processes exercise the gradient all-reduce, the SyncBatchNorm statistics
combination and the item sharding used by bench.py / codeps_amd.step (the RCCL run itself is the driver's)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from codeps_amd.distributed import (SyncBatchNormReducer, allreduce_flat_gradients, init_from_env, shard_items)
    r, _, w = init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    # ---- sharded batch: each rank holds a contiguous slice of the items
    x = torch.randn(6, 5, 4, 3)                      # [items, H, W, C] (NHWC like the product)
    lo, hi = shard_items(x.shape[0], rank, world)
    xs = x[lo:hi].reshape(-1, 3)
    # ---- SyncBN forward statistics == single-process statistics of the whole batch
    red = SyncBatchNormReducer()
    mean, var, count = red(xs.mean(0), xs.var(0, unbiased=False), float(xs.shape[0]))
    full = x.reshape(-1, 3)
    ok_stats = torch.allclose(mean, full.mean(0), atol=1e-6) and torch.allclose(var, full.var(0, unbiased=False), atol=1e-6) \
        and count == full.shape[0]
    # ---- all groups of a layer in one collective
    xa, xb = x[lo:hi, :2].reshape(-1, 3), x[lo:hi, 2:].reshape(-1, 3)       # two statistic groups (row blocks)
    mg, vg, cg = red.combine_groups(torch.stack([xa.mean(0), xb.mean(0)]),
                                    torch.stack([xa.var(0, unbiased=False), xb.var(0, unbiased=False)]),
                                    [float(xa.shape[0]), float(xb.shape[0])])
    fa, fb = x[:, :2].reshape(-1, 3), x[:, 2:].reshape(-1, 3)
    ok_stats = ok_stats and torch.allclose(mg[0], fa.mean(0), atol=1e-6) and torch.allclose(vg[1], fb.var(0, unbiased=False), atol=1e-6) \
        and cg == [float(fa.shape[0]), float(fb.shape[0])]
    # ---- SyncBN backward sums
    g = torch.arange(6.0).view(2, 3) * (rank + 1)
    tot = red.sum(g.clone())
    ok_sums = torch.allclose(tot, torch.arange(6.0).view(2, 3) * 3)
    # ---- flat-gradient all-reduce + 1/world scaling == gradient of the global mean loss
    wgt = torch.ones(3, requires_grad=True)
    loss = (xs @ wgt).mean()                        # per-rank mean over its shard (equal shard sizes)
    loss.backward()
    flat = wgt.grad.clone()
    allreduce_flat_gradients(flat)
    ref = full.mean(0)                              # d/dw of the global mean
    ok_grad = torch.allclose(flat / world, ref, atol=1e-6)
    out.put((rank, ok_stats, ok_sums, ok_grad))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, a, b, c in res:
        assert a and b and c, f"rank {rank}: stats={a} sums={b} grad={c}"


def test_shard_items_and_single_process_noop():
    from codeps_amd.distributed import allreduce_flat_gradients, shard_items
    assert [shard_items(8, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 8)]
    g = torch.ones(4)
    assert allreduce_flat_gradients(g) is g          # world 1: no collective
