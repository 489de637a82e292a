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
        self._counts = {}           # (device, counts) -> device tensor: the per-group element counts never change between steps
        self._gather_out = {}       # (shape, dtype, device, stream) -> the all-gather's output buffer, reused by every step
        self.n_gather = self.n_reduce = 0      # collectives issued so far (bench.py reports them per step)

    def _count_tensor(self, like, counts):
        """Device copy of the local element counts, made once: a fresh host->device copy per BatchNorm layer would
        synchronise the host with the stream 40 times per step and stop it from running ahead of the GPU."""
        key = (like.device, tuple(float(n) for n in counts))
        t = self._counts.get(key)
        if t is None:
            t = torch.tensor(key[1], dtype=like.dtype, device=like.device)
            self._counts[key] = t
        return t

    def __call__(self, mean, var, count):
        world = dist.get_world_size(self.group)
        c = mean.numel()
        packed = torch.cat([mean, var, self._count_tensor(mean, [count])])
        g = self.gather(packed)                        # [world, 2c+1]
        means, vars_, counts = g[:, :c], g[:, c:2 * c], g[:, 2 * c:]
        total = counts.sum()
        mean_all = (means * counts).sum(0) / total
        var_all = ((vars_ + (means - mean_all) ** 2) * counts).sum(0) / total
        # counts are equal on all ranks by construction (equal shards) -> no device->host read-back
        return mean_all, var_all, float(count) * world

    def combine_groups(self, mean, var, counts):
        """All statistic groups of one BatchNorm layer in ONE all-gather: mean / var [G, C], counts = G floats."""
        world = dist.get_world_size(self.group)
        g, c = mean.shape
        packed = torch.cat([mean.reshape(-1), var.reshape(-1), self._count_tensor(mean, counts)])
        allp = self.gather(packed)                                    # [world, 2*G*C + G]
        means = allp[:, :g * c].view(world, g, c)
        vars_ = allp[:, g * c:2 * g * c].view(world, g, c)
        cnts = allp[:, 2 * g * c:].view(world, g, 1)
        total = cnts.sum(0)                                           # [G, 1]
        mean_all = (means * cnts).sum(0) / total
        var_all = ((vars_ + (means - mean_all) ** 2) * cnts).sum(0) / total
        # counts are equal on all ranks by construction (equal shards) -> no host sync needed to read them back
        return mean_all, var_all, [float(n) * world for n in counts]

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


def shard_items(n_items, rank, world):
    """Contiguous slice [lo, hi) of n_items for this rank (equal counts: n_items % world == 0)."""
    assert n_items % world == 0, "keep per-key item counts equal across ranks so the plain mean is exact"
    per = n_items // world
    return rank * per, (rank + 1) * per
