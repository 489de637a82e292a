"""Batch-sharded (data-parallel) execution: one process per GPU (torchrun), RCCL over xGMI through
torch.distributed (backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests).

The path shards over the batch only: every rank runs the step on its slice of
target / replay / source / mixup items; the only data-path exchange is ONE all-reduce (SUM) of the flat fp32
gradient buffer after backward (123 MB for the adapt YAML), plus -- for exact DistributedDataParallel +
SyncBatchNorm semantics (misc/train_utils.py:47-59) -- the per-layer batch-statistics exchange of the
train-mode BatchNorm layers.  Never a line-by-line copy of DDP's bucketed hooks.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment (RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            # no silent default: two jobs launched side by side on one node would meet on the same port.  torchrun exports it
            # (--master-port / --standalone); a hand-rolled launcher has to pick a free one and hand it to every rank
            raise RuntimeError("codeps_amd.distributed.init_from_env: WORLD_SIZE > 1 but MASTER_PORT is not set "
                               "(launch with `python -m torch.distributed.run --master-addr 127.0.0.1 --master-port P ...`)")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def allreduce_flat_gradients(flat_grad, group=None):
    """The single gradient collective of a step (SUM; the optimizer kernel applies 1/world)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return flat_grad


class SyncBatchNormReducer:
    """Cross-rank combination of BatchNorm batch statistics (torch.nn.SyncBatchNorm semantics):
    forward: all-gather (mean, biased var, count) per rank and merge them with the parallel-variance
    formula; backward: all-reduce (SUM) of (sum g, sum g*xhat).  Installed as `encoder.sync_bn`."""

    def __init__(self, group=None):
        self.group = group
        self._gather_out = {}       # (shape, dtype, device, stream) -> the all-gather's output buffer, reused by every step
        self.n_gather = self.n_reduce = 0      # collectives issued so far (bench.py reports them per step)

    def gather(self, packet):
        """All-gather of one contiguous statistics packet -> [world, *packet.shape]: ONE all_gather_into_tensor (no per-rank
        tensor list, no stacking copy); the merge itself is the HIP kernel codeps_bn_sync_combine."""
        world = dist.get_world_size(self.group)
        packet = packet.contiguous()
        # one output buffer per (packet shape, stream): layers of one stream are ordered by the stream, the two encoders run on two
        # streams -- no allocation per layer and step
        key = (tuple(packet.shape), packet.dtype, packet.device, torch.cuda.current_stream().cuda_stream if packet.is_cuda else 0, world)
        buf = self._gather_out.get(key)
        if buf is None:
            buf = self._gather_out[key] = torch.empty((world,) + tuple(packet.shape), dtype=packet.dtype, device=packet.device)
        self.n_gather += 1
        # concatenation along dim 0 is the output form both backends accept (gloo rejects a stacked [world, ...] output)
        dist.all_gather_into_tensor(buf.view((world * packet.shape[0],) + tuple(packet.shape[1:])), packet, group=self.group)
        return buf

    def sum(self, sums):
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.group)
        self.n_reduce += 1
        return sums


def combine_host(gathered, rank_counts=None):
    """Host mirror of codeps_bn_sync_combine's merge (csrc/nn_ops.hip bn_sync_combine_kernel) for tests: gathered [world, 2, G, C] =
    every rank's (mean, biased var) per statistic group -> (mean [G, C], var [G, C]) of the union by the parallel-variance formula;
    rank_counts [world][G] = element counts (None: equal on all ranks, 0: the rank holds nothing of that group)."""
    world, _, g, c = gathered.shape
    w = torch.ones(world, g, 1, dtype=gathered.dtype) if rank_counts is None else torch.as_tensor(rank_counts, dtype=gathered.dtype).view(world, g, 1)
    w = w.to(gathered.device)
    means, vars_ = gathered[:, 0], gathered[:, 1]
    tot = w.sum(0)
    mean = (means * w).sum(0) / tot
    var = ((vars_ + (means - mean) ** 2) * w).sum(0) / tot
    return mean, var


def shard_items(n_items, rank, world):
    """Contiguous slice [lo, hi) of n_items for this rank (equal counts: n_items % world == 0)."""
    assert n_items % world == 0, "keep per-key item counts equal across ranks so the plain mean is exact"
    per = n_items // world
    return rank * per, (rank + 1) * per
