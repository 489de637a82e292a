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
    # ---- SyncBN forward statistics == single-process statistics of the whole batch: the product's path -- ONE all-gather of the
    # [mean | var] packet of all statistic groups (SyncBatchNormReducer.gather), merged by codeps_bn_sync_combine, here by its host mirror
    from codeps_amd.distributed import combine_host
    red = SyncBatchNormReducer()
    xa, xb = x[lo:hi, :2].reshape(-1, 3), x[lo:hi, 2:].reshape(-1, 3)       # two statistic groups (row blocks)
    pk = torch.stack([torch.stack([xa.mean(0), xb.mean(0)]), torch.stack([xa.var(0, unbiased=False), xb.var(0, unbiased=False)])])   # [2, G, C]
    gathered = red.gather(pk)                                                # [world, 2, G, C]
    mg, vg = combine_host(gathered)
    fa, fb = x[:, :2].reshape(-1, 3), x[:, 2:].reshape(-1, 3)
    ok_stats = tuple(gathered.shape) == (world, 2, 2, 3) and torch.allclose(mg[0], fa.mean(0), atol=1e-6) and \
        torch.allclose(mg[1], fb.mean(0), atol=1e-6) and torch.allclose(vg[0], fa.var(0, unbiased=False), atol=1e-6) and \
        torch.allclose(vg[1], fb.var(0, unbiased=False), atol=1e-6)
    # unequal element counts (BASELINE configs[3]: ranks hold different numbers of items of a key): rank r brings r + 1 items of group 0
    xu = x[: rank + 1].reshape(-1, 3)
    pku = torch.stack([xu.mean(0, keepdim=True), xu.var(0, unbiased=False, keepdim=True)])       # [2, 1, C]
    mu, vu = combine_host(red.gather(pku), [[float((r + 1) * 20)] for r in range(world)])
    both = torch.cat([x[: r + 1].reshape(-1, 3) for r in range(world)])
    ok_stats = ok_stats and torch.allclose(mu[0], both.mean(0), atol=1e-6) and torch.allclose(vu[0], both.var(0, unbiased=False), atol=1e-6)
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
    ref = x.reshape(-1, 3).mean(0)                  # d/dw of the global mean
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
