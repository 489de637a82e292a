"""The timed unit: one continual-adaptation step (reference region scripts/adapt_codeps.py:260-299) or
one pre-training step (scripts/train_codeps.py:97-108).

    zero flat grads -> CodepsNet forward (adapt / train) -> sum_k w_k loss_k -> backward
    -> ONE all-reduce of the flat gradient buffer (world > 1) -> fused Adam (+ EMA teacher) on flat buffers

All trainable parameters live in ONE flat fp32 buffer (views handed back to the nn.Parameters), likewise
their gradients, Adam moments and the EMA copy; conv weight / bias gradients are accumulated straight into
the flat buffer by the wgrad / bias kernels (codeps_amd.functional.register_grad_sink).
"""
from collections import OrderedDict

import torch

from . import functional as F
from .distributed import allreduce_flat_gradients

ADAPT_MODULES = OrderedDict([("backbone_po_depth", "backbone_po_depth"), ("backbone_pose_sflow", "backbone_pose_sflow"),
                             ("depth", "depth_head"), ("pose", "pose_head"), ("semantic", "semantic_head"),
                             ("instance", "instance_head")])


def freeze_for_adaptation(net, cfg):
    """misc/train_utils.py:204-226 + scripts/adapt_codeps.py:261-265: modules with cfg.adapt.model.<x> False get
    requires_grad False and stay in eval mode (BN uses running statistics)."""
    frozen = []
    for key, attr in ADAPT_MODULES.items():
        mod = getattr(net, attr, None)
        if mod is None:
            continue
        train = bool(cfg.adapt.model.get(key, False))
        for p in mod.parameters():
            p.requires_grad_(train)
        mod.train(train)
        if not train:
            frozen.append(attr)
    return frozen


def ema_module_lists(cfg, frozen):
    """scripts/adapt_codeps.py:56-72: (modules the EMA filter is applied to, modules the teacher copies from the student).
    `adapt.ema.depth` raises like the reference (its consistency loss does not exist)."""
    if cfg.adapt.ema.get("depth", False):
        raise NotImplementedError("Consistency loss for depth is not implemented yet")         # adapt_codeps.py:70-72
    ema = []
    if cfg.adapt.ema.get("semantic", False) and cfg.adapt.model.get("semantic", False):
        ema.append("semantic_head")
    if cfg.adapt.ema.get("instance", False) and cfg.adapt.model.get("instance", False):
        ema.append("instance_head")
    inverted = [m for m in ("backbone_po_depth", "backbone_pose_sflow", "depth_head", "pose_head", "semantic_head", "instance_head")
                if m not in ema and m not in frozen]
    return ema, inverted


class FlatState:
    """Flat parameter / gradient / Adam / EMA buffers over a list of parameters."""

    def __init__(self, params, with_ema, grad=None):
        """grad: this group's slice of the step's ONE flat gradient buffer (StepRunner.grad_all), so that the whole step has a
        single all-reduce; None allocates a private buffer."""
        self.params = params
        n = sum(p.numel() for p in params)
        dev = params[0].device
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev) if grad is None else grad
        assert self.grad.numel() == n and self.grad.is_contiguous()
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in params:
            k = p.numel()
            self.flat[off:off + k] = p.data.reshape(-1)
            p.data = self.flat[off:off + k].view_as(p)
            g = self.grad[off:off + k].view_as(p)
            p.grad = g
            if p.dim() == 4 and p.shape[1] != 1 or (p.dim() == 1 and (getattr(p, "_codeps_conv_bias", False) or
                                                                     getattr(p, "_codeps_bn_affine", False))):
                F.register_grad_sink(p, g)
            off += k
        self.ema = self.flat.clone() if with_ema else None
        self.numel = n
        self.offsets = [0]
        for p in params:
            self.offsets.append(self.offsets[-1] + p.numel())
        # torch.optim.Adam keeps one step count per parameter and skips parameters whose .grad is None (the reference zeroes
        # with set_to_none): `steps` is that count -- an int while every parameter has received every step, else a list
        self.steps = 0

    def adam(self, touched, lr, grad_scale, ema_alpha, apply_ema):
        """One optimizer step.  touched: None (every parameter has a gradient) or one bool per parameter -- the parameters the
        step's losses reach; the others keep value, moments and step count, exactly as torch.optim.Adam treats grad None
        (scripts/adapt_codeps.py:289-291).  The EMA teacher filter runs over every parameter either way (online_adap.py:279-289)."""
        ema = self.ema if apply_ema else None
        if (touched is None or all(touched)) and isinstance(self.steps, int):
            self.steps += 1
            F.adam_ema_step(self.flat, self.grad, self.m, self.v, self.steps, lr=lr, grad_scale=grad_scale, ema_params=ema, ema_alpha=ema_alpha)
            return
        n = len(self.params)
        if isinstance(self.steps, int):
            self.steps = [self.steps] * n
        touched = [True] * n if touched is None else list(touched)
        i = 0
        while i < n:                                       # runs of neighbouring parameters with the same fate and step count
            j = i + 1
            while j < n and touched[j] == touched[i] and self.steps[j] == self.steps[i]:
                j += 1
            a, b = self.offsets[i], self.offsets[j]
            if touched[i]:
                step = self.steps[i] + 1
                for k in range(i, j):
                    self.steps[k] = step
                F.adam_ema_step(self.flat[a:b], self.grad[a:b], self.m[a:b], self.v[a:b], step, lr=lr, grad_scale=grad_scale,
                                ema_params=None if ema is None else ema[a:b], ema_alpha=ema_alpha)
            elif ema is not None:
                ema[a:b].mul_(ema_alpha).add_(self.flat[a:b], alpha=1.0 - ema_alpha)
            i = j
        if len(set(self.steps)) == 1:
            self.steps = self.steps[0]


def _mark_conv_biases(net):
    """Tag the 1-D parameters whose backward kernels can accumulate straight into the flat gradient buffer."""
    for m in net.modules():
        if isinstance(m, torch.nn.Conv2d) and m.bias is not None:
            m.bias._codeps_conv_bias = True
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.affine:
            m.weight._codeps_bn_affine = True
            m.bias._codeps_bn_affine = True


class StepRunner:
    """mode "adapt": CodepsNet.adapt with the adapt loss weights; mode "train": CodepsNet.forward("train")."""

    def __init__(self, net, cfg, mode="adapt", world_size=1, group=None, lr=None):
        self.net, self.cfg, self.mode, self.world, self.group = net, cfg, mode, world_size, group
        self.weights = dict(cfg.losses.weights)
        self.lr = float(cfg.train.optimizer.learning_rate if lr is None else lr)
        self._one = None
        _mark_conv_biases(net)
        ema_mods = []
        if mode == "adapt":
            self.frozen = freeze_for_adaptation(net, cfg)
            ema_mods, _ = ema_module_lists(cfg, self.frozen)
            self.ema_alpha = float(cfg.adapt.ema.alpha)
        else:
            net.train()
            self.frozen, self.ema_alpha = [], 0.0
        # parameters that actually receive gradients (dead dispconvs.1-3 / fc are excluded instead of relying on
        # find_unused_parameters)
        def live(name):
            return not (".fc." in name or name.startswith("fc.") or any(f"dispconvs.{i}." in name for i in (1, 2, 3)))
        ema_params, other = [], []
        # heads first, encoders last: the backward pass finishes the heads' gradients long before the encoders' (the pose encoder's
        # stem weight gradient is the step's last kernel), so the flat buffer's front can be all-reduced while the back is still being
        # written (world > 1: _segment_ready)
        n_heads = 0
        for attr in sorted(ADAPT_MODULES.values(), key=lambda a: a.startswith("backbone")):
            mod = getattr(net, attr, None)
            if mod is None:
                continue
            for name, p in mod.named_parameters():
                if p.requires_grad and live(name):
                    (ema_params if attr in ema_mods else other).append(p)
                    if not attr.startswith("backbone") and attr not in ema_mods:
                        n_heads += p.numel()
        # ONE flat gradient buffer for the whole step (north_star: "a single RCCL all-reduce of gradients"); the EMA'd modules'
        # parameters are its first range, so the two FlatStates (they differ only in whether an EMA copy is kept) are views of it
        sets = [(ps, with_ema) for ps, with_ema in ((ema_params, True), (other, False)) if ps]
        total = sum(p.numel() for ps, _ in sets for p in ps)
        self.grad_all = torch.zeros(total, dtype=torch.float32, device=sets[0][0][0].device)
        self.groups, off = [], 0
        for ps, with_ema in sets:
            n = sum(p.numel() for p in ps)
            self.groups.append(FlatState(ps, with_ema, self.grad_all[off:off + n]))
            off += n
        self.step_count = 0
        # ---- world > 1: the gradient all-reduce in two segments.  Segment A = the EMA'd modules + the other heads (the flat buffer's
        # front), launched on its own stream from inside the backward pass as soon as the last kernel that writes it has been issued
        # (the data-gradient / weight-gradient node of every head's FIRST convolution is that head's last backward node: it calls
        # _segment_ready through functional's `_codeps_grad_done` hook); segment B = the encoders, after the backward pass.  The
        # collectives stay SUMs of the same buffer: results are those of the single all-reduce.  OPT-IN (CODEPS_AR_OVERLAP=1) since round 6:
        # the default is the ONE collective north_star describes until a box with >= 2 GPUs has shown that the overlap wins.
        # A marked layer's backward node fires once per forward call that takes part in the graph -- a head runs once per resolution (the
        # shipped adaptation yamls: source 512 x 1024, target 384 x 1408), twice for the pose pairs, once per key with a trainable backbone --
        # so the walk over the step's autograd graph (`_reached`) COUNTS the marked layers' backward nodes that the losses reach, and the front
        # collective starts when as many have fired.
        import os
        self._seg, self._seg_marks, self._early_end, self.comm_stream = None, [], 0, None
        self.ar_overlap_steps = 0
        self._seg_expected = 0
        n_front = sum(p.numel() for p in ema_params) + n_heads
        if world_size > 1 and os.environ.get("CODEPS_AR_OVERLAP", "0") == "1" and 0 < n_front < total and self.grad_all.is_cuda:
            for attr in ("depth_head", "semantic_head", "pose_head"):
                mod = getattr(net, attr, None)
                if mod is None or attr in self.frozen:
                    continue
                first = mod.net[0] if attr == "pose_head" else mod.upconvs_0["4"][0]
                if first.weight.requires_grad:
                    self._seg_marks.append(first.weight)
            trainable_heads = [a for a in ADAPT_MODULES.values() if not a.startswith("backbone") and getattr(net, a, None) is not None
                               and any(p.requires_grad for p in getattr(net, a).parameters())]
            # (a trainable head without a mark -- the instance head of the pre-training step -- keeps the single collective)
            if self._seg_marks and len(self._seg_marks) == len(trainable_heads):
                self._early_end = n_front
                self.comm_stream = torch.cuda.Stream()
                for w in self._seg_marks:
                    w._codeps_grad_done = self._segment_ready
        # device step counter mixed into the photometric tie-break noise (bumped on the stream at the start of every step, so a
        # replayed HIP graph draws new noise) + this runner's host seed base for the calls within one step
        self.noise_counter = torch.zeros(1, dtype=torch.int64, device=self.groups[0].flat.device)
        self.seed_base = 0x5EED
        self._graph = None
        # Branch concurrency: the pose networks (they depend on the images only) and the instance / semantic branch run on side
        # streams next to the shared encoder / the depth branch -- forward by these forks, backward through autograd's stream
        # bookkeeping (every backward op runs on its forward's stream); inside the captured HIP graph they become parallel
        # branches.  Measured at config 3: 41.4 -> 37.5 ms per step with the pose stream alone.  CODEPS_STREAMS=0 serialises.
        import os
        if os.environ.get("CODEPS_STREAMS", "1") != "0" and getattr(net, "depth_algo", None) is not None:
            net.pose_stream = torch.cuda.Stream()          # (a HIGH-priority pose stream starves the critical path: 178 vs 213 images/s)
            if mode == "adapt" and os.environ.get("CODEPS_STREAMS", "1") != "1p":
                net.branch_stream = torch.cuda.Stream()
        self.key_scale = {}             # per-loss re-weighting of a batch-sharded step with unequal shards (set_key_counts)
        self._reach_cache = {}          # active losses -> per group, which parameters the backward pass reaches
        self._found_ids = frozenset()
        self._unequal_shards = False
        self._touched = None

    def set_key_counts(self, items, rank):
        """Batch-sharded step whose ranks hold DIFFERENT numbers of items per key (BASELINE configs[3], SURVEY.md 8d: ranks
        0-3 hold 1 target + 1 replay item, ranks 4-7 two replay items).  items = {key: [items on rank 0, 1, ...]}.
        With it the count-weighted depth means (global image counts), SyncBatchNorm (each rank's statistics weighed with its
        element count) and the car-not-moving rule (evaluated by the rank that owns the first target item, flag broadcast)
        reproduce the single-process step on the concatenated batch EXACTLY.  The other per-key means are re-weighted by
        local / (global / world) items, which is exact for plain means only: the semantic losses mine hard pixels (top-k over
        all pixels of a key's batch, algos/semantic_seg.py:45-67), and the count-weighted mean of per-rank top-k means is not
        the top-k of the concatenated batch -- the approximation the reference's DistributedDataParallel run makes too (each
        rank mines its own pixels); identical only for top_k = 1.0 or a key held by one rank.  Equal shards need none of
        this (the plain DDP mean is exact)."""
        world = self.world
        assert all(len(v) == world for v in items.values()), "one item count per rank and key"
        self._unequal_shards = True
        h, w = self.cfg.dataset.feed_img_size
        hs, ws = self.cfg.adapt.source_dataset.feed_img_size
        res = {k: ((hs, ws) if k == "source" or k.endswith("mixup") else (h, w)) for k in items}
        self.net.depth_algo.shard_layout = dict(world=world, rank=rank, group=self.group, items={k: list(v) for k, v in items.items()}, res=res)
        self.key_scale = {}
        for loss_name, key in (("semantic_source", "source"), ("center_source", "source"), ("offset_source", "source")):
            if key in items and sum(items[key]) > 0:
                self.key_scale[loss_name] = items[key][rank] * world / sum(items[key])
        for key in items:
            if key.endswith("mixup") and sum(items[key]) > 0:
                self.key_scale["semantic_" + key] = items[key][rank] * world / sum(items[key])

    @property
    def grad_bytes(self):
        return self.grad_all.numel() * 4

    def forward_backward(self, data):
        F.set_noise_stream(self.noise_counter, self.seed_base)
        F.step_counter_add(self.noise_counter, 1)
        F.refresh_weight_packs()                 # every trainable weight's packs in one launch (after the first step built the table)
        self.grad_all.zero_()
        if self.mode == "adapt":
            losses, result, _, _ = self.net(data, "adapt", [0, -1, 1], do_panoptic_fusion=True)
        else:
            losses, result, _ = self.net(data, "train", [0, -1, 1])
        terms, wts = [], []
        for k, v in losses.items():
            w = self.weights.get(k)
            if v is None or w is None or not v.requires_grad:        # scripts/adapt_codeps.py:274-288
                continue
            terms.append(v)
            wts.append(float(w) * self.key_scale.get(k, 1.0))
        total = F.weighted_sum(terms, wts)
        self._touched = self._reached(tuple(k for k, v in losses.items() if any(v is t for t in terms)), total)
        self._seg = None
        if self._early_end and not torch.cuda.is_current_stream_capturing():
            # backward nodes of the marked layers that will fire (counted on the graph by _reached; a marked head the losses do not reach, or
            # unequal shards -- no walk -- leave it at 0: allreduce_gradients then sends the whole buffer in one collective)
            need = self._seg_expected if all(id(w) in self._found_ids for w in self._seg_marks) else 0
            if need:
                self._seg = {"need": need, "events": [], "work": None}
        if self._one is None or self._one.device != total.device:
            self._one = torch.ones((), dtype=total.dtype, device=total.device)
        F.arm_colsum_deferral(True)              # bias-gradient finishes of this backward pass: one launch per stream (F.flush_colsums)
        try:
            total.backward(self._one)            # (the default seed is an ATen ones_like launch per step)
            F.flush_colsums()                    # (each stream's on that stream: before the side streams are joined below)
        finally:
            F.arm_colsum_deferral(False)
        for side in (getattr(self.net, "pose_stream", None), getattr(self.net, "branch_stream", None)):
            if side is not None:      # the gradient sinks are written by kernels autograd does not track: join the side streams by hand
                torch.cuda.current_stream().wait_stream(side)
        if F._pack_table["dev"] is None and F._pack_table.get("builds", 0) < 8 and not torch.cuda.is_current_stream_capturing():
            F.build_pack_table()                 # the first step created every layer's persistent pack buffers: table them
        return total.detach(), losses, result

    def _reached(self, active, total):
        """Which parameters receive a gradient from this step's losses (one bool per parameter and group; None = all).  A
        property of the autograd graph, not of the values: the graph is walked once per set of active losses.  The conv / BN
        kernels write their weight gradients straight into the flat buffer, but the weights are inputs of the autograd
        Functions that launch them, so they appear as leaves of the graph like any other parameter.  Batch-sharded steps whose
        ranks hold different keys (set_key_counts) would need the union over ranks: they treat every parameter as reached."""
        if self.world > 1 and self._unequal_shards:
            self._seg_expected = 0
            return None
        hit = self._reach_cache.get(active)
        if hit is None:
            seen, found, stack = set(), set(), [total.grad_fn]
            marks, marked_nodes = {id(w) for w in self._seg_marks}, 0
            while stack:
                fn = stack.pop()
                if fn is None or fn in seen:
                    continue
                seen.add(fn)
                var = getattr(fn, "variable", None)
                if var is not None:
                    found.add(id(var))
                wref = getattr(fn, "weight_ref", None)          # a convolution's backward node (functional._Conv2d: the node IS its ctx)
                if marks and wref is not None and id(wref()) in marks:
                    marked_nodes += 1                            # ... of a marked layer: it will fire once (a head runs once per resolution / pair / key)
                stack.extend(nf for nf, _ in fn.next_functions)
            masks = [[id(p) in found for p in g.params] for g in self.groups]
            hit = self._reach_cache[active] = (None if all(all(m) for m in masks) else masks, frozenset(found), marked_nodes)
        self._found_ids = hit[1]
        self._seg_expected = hit[2]
        return hit[0]

    def _segment_ready(self):
        """Called (on autograd's thread) right after the last backward kernels of one head have been issued.  When every head the
        step's losses reach has reported, the flat buffer's front is complete on the device once the recorded events have passed: its
        all-reduce starts on the communication stream, next to the encoders' backward pass."""
        seg = self._seg
        if seg is None or seg["work"] is not None:
            return
        F.flush_colsums(current_only=True)                   # the head's pending bias-gradient finishes belong in front of the event
        ev = torch.cuda.Event()
        ev.record()                                          # this node's stream: everything the head's backward issued precedes it
        seg["events"].append(ev)
        if len(seg["events"]) < seg["need"]:
            return
        import torch.distributed as dist
        with torch.cuda.stream(self.comm_stream):
            for e in seg["events"]:
                self.comm_stream.wait_event(e)
            seg["work"] = dist.all_reduce(self.grad_all[:self._early_end], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def allreduce_gradients(self):
        """The step's gradient collective after forward_backward: the whole flat buffer, or -- when the backward pass has already
        started the front segment's all-reduce on the communication stream (_segment_ready) -- the back segment, then the join."""
        seg, self._seg = self._seg, None
        if seg is not None and seg["work"] is not None:
            allreduce_flat_gradients(self.grad_all[self._early_end:], self.group)        # segment B: the encoders
            seg["work"].wait()                                                            # segment A has been under way since the heads finished
            self.ar_overlap_steps += 1
        else:
            allreduce_flat_gradients(self.grad_all, self.group)
        return self.grad_all

    def optimizer_step(self, apply_ema=True):
        """ONE all-reduce of the step's flat gradient buffer, then the fused Adam kernel per flat group (EMA'd modules / the
        rest).  apply_ema False leaves the EMA copy alone: the reference filters the teacher once per FRAME, after its
        `nof_adaptation_steps` optimizer steps (scripts/adapt_codeps.py:266-299)."""
        self.step_count += 1
        self.allreduce_gradients()
        for i, g in enumerate(self.groups):
            g.adam(None if self._touched is None else self._touched[i], self.lr, 1.0 / self.world, self.ema_alpha, apply_ema)

    def step(self, data, apply_ema=True):
        out = self.forward_backward(data)
        self.optimizer_step(apply_ema)
        return out

    # ------------------------------------------------------------------ HIP-graph form of the same step
    def capture(self, data):
        """Capture forward + backward of THIS batch's buffers into a HIP graph.  `data` holds the static input tensors (refill
        them in place between replays).  Everything a step decides per frame lives on the stream: the weight packs of the
        trainable parameters are re-filled inside the graph (the cache epoch is bumped right before the capture, so the pack
        kernels are recorded and write persistent buffers), the tie-break noise is seeded from a device counter and the
        "car not moving" rule runs in a kernel -- a replay is a real adaptation step on the current weights
        (tests/test_graph_gpu.py compares K replayed steps with K eagerly launched ones bit for bit).  The gradient
        all-reduce and the optimizer stay outside the graph (Adam's bias correction is a host scalar)."""
        counter_before = self.noise_counter.clone()
        rule = getattr(self.net.depth_algo, "_rule_state", None) if getattr(self.net, "depth_algo", None) is not None else None
        rule_before = None if rule is None else rule.clone()
        bn_before = [(b, b.clone()) for b in self._bn_buffers()]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                     # warm-up on a side stream: allocations + lazily created state
            self.forward_backward(data)
        torch.cuda.current_stream().wait_stream(side)
        F.invalidate_weight_cache()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            total, losses, result = self.forward_backward(data)
            static = (total, OrderedDict((k, None if v is None else v.detach()) for k, v in losses.items()),
                      OrderedDict((k, v.detach() if torch.is_tensor(v) else v) for k, v in result.items()))
            del total, losses, result                     # nothing that holds the autograd tape survives the capture
        # the warm-up advanced per-step state: put it back so that the first replay is the step the eager path would run next
        self.noise_counter.copy_(counter_before)
        rule = getattr(self.net.depth_algo, "_rule_state", None) if getattr(self.net, "depth_algo", None) is not None else None
        if rule is not None:
            rule.copy_(rule_before) if rule_before is not None else rule.zero_()
        for b, old in bn_before:
            b.copy_(old)
        self._graph, self._static = graph, static
        return static

    def _bn_buffers(self):
        return [b for m in self.net.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.training
                for b in (m.running_mean, m.running_var, m.num_batches_tracked) if b is not None]

    def replay(self, apply_ema=True):
        """One step from the captured graph: forward + backward replayed, then all-reduce + Adam/EMA.  Returns the static
        (total, losses, result) tensors the replay has just refreshed."""
        if self._graph is None:
            raise RuntimeError("StepRunner.replay: call capture(data) first")
        self._graph.replay()
        self.optimizer_step(apply_ema)
        return self._static
