"""torch.autograd.Function wrappers over the C ABI (include/codeps_hip.h).

PyTorch supplies device memory (caching allocator), the current HIP stream and the autograd tape;
all arithmetic happens in libcodeps_hip.so.  Every function raises on CPU tensors -- no fallback.
"""
import ctypes
import os
import weakref

import torch

from . import _lib
from ._lib import CodepsHipError, check, lib, ptr, require_cuda, stream_ptr

_seed_counter = [0x5EED]
_noise_counter = [None]       # device uint64 step counter mixed into the photometric tie-break seed (set by the step harness)


def set_noise_stream(counter, seed_base=None):
    """Bind the photometric tie-break noise to a DEVICE step counter (int64 tensor [1], bumped on the stream by
    `step_counter_add`) and, optionally, restart the host-side per-call seed sequence.  With both, the seeds a step uses are a
    function of (call position within the step, device counter): a captured HIP graph then draws fresh noise on every replay and
    agrees bit for bit with the eagerly launched step."""
    _noise_counter[0] = counter
    if seed_base is not None:
        _seed_counter[0] = int(seed_base)


def step_counter_add(counter, increment=1):
    require_cuda(counter)
    check(lib().codeps_step_counter_add(ptr(counter), int(increment), stream_ptr()), "codeps_step_counter_add")


def counters_add(counters, increments):
    """counters[i] += increments[i] for 0-dim / 1-element int64 device tensors, one launch (C ABI: codeps_counters_add)."""
    n = len(counters)
    if n == 0:
        return
    for t in counters:
        require_cuda(t)
        if t.dtype != torch.int64 or t.numel() != 1:
            raise CodepsHipError("counters_add: int64 tensors with one element")
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in counters])
    incs = (ctypes.c_int * n)(*[int(v) for v in increments])
    check(lib().codeps_counters_add(ptrs, incs, n, stream_ptr()), "codeps_counters_add")


def concat_rows(parts, out=None):
    """torch.cat(parts, 0) for contiguous fp32 device tensors of equal trailing shape in one launch of this library (<= 8 parts per
    launch; C ABI: codeps_concat_f32).  `out`: optional destination (flat length = the parts' total)."""
    parts = [p_.contiguous() for p_ in parts]
    for p_ in parts:
        require_cuda(p_)
        if p_.dtype != torch.float32:
            raise CodepsHipError("concat_rows: fp32 tensors")
    total = sum(p_.numel() for p_ in parts)
    if out is None:
        out = torch.empty((sum(p_.shape[0] for p_ in parts),) + tuple(parts[0].shape[1:]), dtype=torch.float32, device=parts[0].device)
    flat, pos = out.view(-1), 0
    if flat.numel() < total:
        raise CodepsHipError("concat_rows: destination too small")
    for i in range(0, len(parts), 8):
        grp = parts[i:i + 8]
        srcs = (ctypes.c_void_p * len(grp))(*[g.data_ptr() for g in grp])
        cnts = (ctypes.c_longlong * len(grp))(*[g.numel() for g in grp])
        check(lib().codeps_concat_f32(srcs, cnts, len(grp), ctypes.c_void_p(flat.data_ptr() + 4 * pos), stream_ptr()), "codeps_concat_f32")
        pos += sum(g.numel() for g in grp)
    return out


def _f32c(t):
    """fp32 + contiguous without copying when already so."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class _PhotometricLoss(torch.autograd.Function):
    """Fused multi-scale photometric loss (C ABI: codeps_photometric_{fwd,bwd})."""

    @staticmethod
    def forward(ctx, img_t, img_m, img_p, depth, pose_m, pose_p, intr, noise, num_scales, alpha, seed, counter=None):
        require_cuda(img_t, img_m, img_p, depth, pose_m, pose_p, intr, noise, counter)
        img_t, img_m, img_p, depth = (_f32c(x) for x in (img_t, img_m, img_p, depth))
        pose_m, pose_p, intr = _f32c(pose_m), _f32c(pose_p), _f32c(intr)
        noise = None if noise is None else _f32c(noise)
        b, _, h, w = img_t.shape
        desc = _lib.PhotometricDesc(b, h, w, num_scales, alpha, seed)
        nbytes = lib().codeps_photometric_workspace_bytes(ctypes.byref(desc))
        if nbytes == 0:
            raise _lib.CodepsHipError("photometric: " + lib().codeps_last_error().decode())
        ws = torch.empty(nbytes, dtype=torch.uint8, device=img_t.device)
        loss = torch.empty(1, dtype=torch.float32, device=img_t.device)
        t0 = _prof_begin()
        check(lib().codeps_photometric_fwd(ctypes.byref(desc), ptr(img_t), ptr(img_m), ptr(img_p), ptr(depth),
                                           ptr(pose_m), ptr(pose_p), ptr(intr), ptr(noise), ptr(counter), ptr(loss), ptr(ws),
                                           nbytes, stream_ptr()), "codeps_photometric_fwd")
        _prof_end(t0, "photo_fwd", 40.0 * b * h * w)          # algorithmic bytes: 9 image planes + depth read once
        ctx.save_for_backward(img_t, img_m, img_p, depth, pose_m, pose_p, intr, ws)
        ctx.desc = desc
        return loss[0]

    @staticmethod
    def backward(ctx, grad_loss):
        img_t, img_m, img_p, depth, pose_m, pose_p, intr, ws = ctx.saved_tensors
        g = _f32c(grad_loss).reshape(1)
        gd = torch.empty_like(depth)
        gpm, gpp = torch.empty_like(pose_m), torch.empty_like(pose_p)
        t0 = _prof_begin()
        check(lib().codeps_photometric_bwd(ctypes.byref(ctx.desc), ptr(img_t), ptr(img_m), ptr(img_p), ptr(depth),
                                           ptr(pose_m), ptr(pose_p), ptr(intr), ptr(g), ptr(gd), ptr(gpm), ptr(gpp),
                                           ptr(ws), ws.numel(), stream_ptr()), "codeps_photometric_bwd")
        _prof_end(t0, "photo_bwd", 44.0 * depth.numel())       # the same planes again + d depth written
        return None, None, None, gd, gpm, gpp, None, None, None, None, None, None


def photometric_loss(images, depth, poses, intrinsics, num_scales=5, alpha=0.85, noise=None, seed=None):
    """Multi-scale photometric reconstruction loss with auto-masking (reference:
    ReconstructionLoss.__call__, algos/depth.py:239-326).

    images = (I_t, I_t-1, I_t+1) [B,3,H,W]; depth [B,1,H,W]; poses = (T_t->t-1, T_t->t+1) [B,4,4];
    intrinsics [B,4] (fx,fy,cx,cy) at full resolution, kept on the device (no host sync).
    noise: optional flat tensor of tie-break noise (already x1e-5), scale-major then [B,2,Hs,Ws];
    default = in-kernel counter-based normal noise seeded from `seed` / an internal counter.
    """
    if seed is None:
        _seed_counter[0] += 1
        seed = _seed_counter[0]
    return _PhotometricLoss.apply(images[0], images[1], images[2], depth, poses[0].reshape(-1, 16),
                                  poses[1].reshape(-1, 16), intrinsics, noise, int(num_scales), float(alpha),
                                  int(seed), _noise_counter[0] if noise is None else None)


# =============================================================================================
# NHWC activations.  Inside codeps_amd every feature map is a contiguous fp32 tensor [B, H, W, Cp]
# with Cp = channel count padded to a multiple of 16 (padding channels hold zeros).  The reference's
# NCHW interface is kept at module boundaries through zero-copy views (see `to_nchw` / `from_nchw`).
# =============================================================================================
def pad16(c):
    return (int(c) + 15) // 16 * 16


def from_nchw(x, c_pad=None):
    """NCHW tensor (any strides) -> NHWC [B,H,W,Cp].  Zero-copy when `x` is a view handed out by
    `to_nchw` (or any channels-last view of the right width); otherwise one layout-conversion kernel."""
    require_cuda(x)
    b, c, h, w = x.shape
    cp = pad16(c) if c_pad is None else c_pad
    base = getattr(x, "_codeps_nhwc", None)        # padded NHWC parent of a to_nchw() view
    if base is not None and tuple(base.shape) == (b, h, w, cp):
        return base
    if x.dtype == torch.float32 and c == cp and x.stride(1) == 1 and x.permute(0, 2, 3, 1).is_contiguous():
        return x.permute(0, 2, 3, 1)
    return _NchwToNhwc.apply(x, cp)


class _NchwToNhwc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cp):
        x = _f32c(x)
        b, c, h, w = x.shape
        ctx.c = c
        y = torch.empty(b, h, w, cp, dtype=torch.float32, device=x.device)
        check(lib().codeps_nchw_to_nhwc(ptr(x), b, c, h, w, ptr(y), cp, 0, 1, stream_ptr()), "codeps_nchw_to_nhwc")
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _f32c(dy)
        b, h, w, cp = dy.shape
        dx = torch.empty(b, ctx.c, h, w, dtype=torch.float32, device=dy.device)
        check(lib().codeps_nhwc_to_nchw(ptr(dy), b, ctx.c, h, w, cp, ptr(dx), stream_ptr()), "codeps_nhwc_to_nchw")
        return dx, None


def to_nchw(y, c):
    """NHWC [B,H,W,Cp] -> NCHW-shaped zero-copy view [B,c,H,W] (channels-last strides)."""
    v = y.permute(0, 3, 1, 2)[:, :c]
    v._codeps_nhwc = y
    return v


def _to_nhwc_into(im, y, c_pad, c_off, zero_rest, xpad):
    b, c, h, w = im.shape
    check(lib().codeps_nchw_to_nhwc_xpad(ptr(im), b, c, h, w, ptr(y), c_pad, c_off, zero_rest, xpad[0], xpad[1], stream_ptr()),
          "codeps_nchw_to_nhwc_xpad")


STEM_XPAD = (4, 2)      # zero columns left / right of every row in the double-pixel stem layout (see stem_conv2d)


_FRAMES_MULTI = os.environ.get("CODEPS_FRAMES_MULTI", "1") != "0"       # A/B switch: 0 = one launch per frame


def _frames_to_nhwc(slabs, y, c_pad, xpad):
    """slabs = [(frame_a [B_i,C,H,W], frame_b or None), ...] -> y [sum B_i, H, W + xpad, c_pad] in ONE launch (C ABI codeps_frames_to_nhwc);
    more than 16 slabs: the per-frame launches."""
    _, c, h, w = slabs[0][0].shape
    if len(slabs) > 16 or not _FRAMES_MULTI:
        off = 0
        for fa, fb in slabs:
            b = fa.shape[0]
            _to_nhwc_into(fa, y[off:off + b], c_pad, 0, 1, xpad)
            if fb is not None:
                _to_nhwc_into(fb, y[off:off + b], c_pad, c, 0, xpad)
            off += b
        return y
    n = len(slabs)
    pa = (ctypes.c_void_p * n)(*[fa.data_ptr() for fa, _ in slabs])
    pb = (ctypes.c_void_p * n)(*[(fb.data_ptr() if fb is not None else None) for _, fb in slabs])
    nb = (ctypes.c_int * n)(*[fa.shape[0] for fa, _ in slabs])
    check(lib().codeps_frames_to_nhwc(pa, pb, nb, n, c, h, w, ptr(y), c_pad, xpad[0], xpad[1], stream_ptr()), "codeps_frames_to_nhwc")
    return y


def images_to_nhwc(images, c_pad=16, xpad=(0, 0)):
    """[B,3,H,W] NCHW image(s) -> one NHWC tensor with the images stacked along channels and zero
    padding up to c_pad (stem input; two frames for the pose net); xpad = extra zero columns (left, right)."""
    imgs = [_f32c(i) for i in images]
    require_cuda(*imgs)
    b, c, h, w = imgs[0].shape
    y = torch.empty(b, h, w + sum(xpad), c_pad, dtype=torch.float32, device=imgs[0].device)
    if len(imgs) <= 2:
        return _frames_to_nhwc([(imgs[0], imgs[1] if len(imgs) == 2 else None)], y, c_pad, xpad)
    for k, im in enumerate(imgs):
        _to_nhwc_into(im, y, c_pad, k * c, 1 if k == 0 else 0, xpad)
    return y


def frame_pairs_to_nhwc(pairs, c_pad=16, xpad=(0, 0)):
    """[(frame_a [B_i,3,H,W], frame_b [B_i,3,H,W]), ...] -> NHWC [sum B_i, H, W, c_pad] with the two frames stacked
    along channels (the pose net's input, algos/depth.py:408-413) and the pairs concatenated along the batch."""
    require_cuda(*[t for p in pairs for t in p])
    _, c, h, w = pairs[0][0].shape
    total = sum(p[0].shape[0] for p in pairs)
    y = torch.empty(total, h, w + sum(xpad), c_pad, dtype=torch.float32, device=pairs[0][0].device)
    return _frames_to_nhwc([(_f32c(fa), _f32c(fb)) for fa, fb in pairs], y, c_pad, xpad)


def image_batches_to_nhwc(images, c_pad=16, xpad=(0, 0)):
    """[B_i,C,H,W] NCHW batches (same C,H,W) -> one NHWC tensor [sum B_i, H, W, c_pad] (batch concat on the fly)."""
    imgs = [_f32c(i) for i in images]
    require_cuda(*imgs)
    _, c, h, w = imgs[0].shape
    total = sum(i.shape[0] for i in imgs)
    y = torch.empty(total, h, w + sum(xpad), c_pad, dtype=torch.float32, device=imgs[0].device)
    return _frames_to_nhwc([(im, None) for im in imgs], y, c_pad, xpad)


_ws_cache = {}


def _weak(obj, cache, key):
    """Weak reference to `obj` that drops cache[key] when the object dies (the caches are keyed by id(), which is re-used)."""
    return weakref.ref(obj, lambda _r, c=cache, k=key: c.pop(k, None))


def _workspace(nbytes, device, tag):
    """Per-(device, tag) scratch buffer, grown on demand (stream-ordered reuse, like a cuDNN workspace)."""
    key = (device.index, tag, torch.cuda.current_stream(device).cuda_stream)     # per stream: concurrent branches must not share scratch
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _mark_written():
    """(event, stream id) after a cached buffer was (re)written on the current stream: a user on ANOTHER stream must wait for it
    (branches of a step run on side streams, codeps_amd.codeps.online_adap)."""
    ev = torch.cuda.Event()
    ev.record()
    return ev, torch.cuda.current_stream().cuda_stream


def _wait_written(mark):
    if mark is not None and mark[1] != torch.cuda.current_stream().cuda_stream:
        torch.cuda.current_stream().wait_event(mark[0])


_pack_cache = {}           # frozen parameters: valid while (storage, version) is unchanged
_pack_cache_train = {}     # trainable parameters: PERSISTENT pack buffers, re-filled in place once per weights epoch
_weights_epoch = [0]       # bumped by every in-place optimizer update done through this library


def _is_volatile(t):
    """True for tensors whose VALUES may change without their torch version changing: trainable parameters (the fused optimizer
    writes them through raw pointers), teacher views of the optimizer's buffers, and anything derived from those per epoch."""
    return bool(t.requires_grad or getattr(t, "_codeps_volatile", False))


def invalidate_weight_cache(everything=False):
    """Call after modifying parameters in place by means torch cannot see (raw-pointer kernels), and immediately BEFORE
    capturing a HIP graph of a step: the packs of trainable parameters keep their buffers (a captured graph reads those
    addresses for as long as it lives) and are re-filled in place by the next `pack_weights` call of the new epoch -- inside a
    capture that re-fill is recorded, so every replay repacks the weights the optimizer has just written."""
    _weights_epoch[0] += 1
    if everything:
        _pack_cache.clear()


def _arith():
    """The conv engine's operand arithmetic (codeps_conv2d_set_arith; a pack made under 16 carries every form, one made under the
    others lacks the fp16 pieces: the arithmetic is part of a pack's stamp)."""
    return lib().codeps_conv2d_set_arith(-1)


def pack_weights(weight, out_c_pad, in_c_pad, need_dgrad):
    """OIHW parameter -> (w_fwd, w_dgrad) packs.  Frozen weights are packed once per (storage, tensor version).  Trainable
    ones (and frozen-looking ones whose storage is rewritten by raw-pointer kernels: an EMA teacher bound to the optimizer's
    flat buffers, codeps.frame.OnlineAdapter) own persistent pack buffers that are re-filled once per weights epoch, i.e.
    once per optimizer step (the pose encoder runs 6x per step)."""
    volatile = weight.requires_grad or getattr(weight, "_codeps_volatile", False)
    cache = _pack_cache_train if volatile else _pack_cache
    key = (id(weight), out_c_pad, in_c_pad)
    stamp = (weight._version, _weights_epoch[0] if volatile else 0, _arith())
    ent = cache.get(key)          # [weakref to the parameter, storage, stamp, w_fwd, w_dgrad]: ids / addresses get reused
    same = ent is not None and ent[0]() is weight and ent[1] == weight.data_ptr()
    if same and ent[2] == stamp and (ent[4] is not None or not need_dgrad):
        _wait_written(ent[5])
        return ent[3], ent[4]
    co, ci, kh, kw = weight.shape
    w = _f32c(weight.detach())
    # each pack carries its pre-split (3 x bf16) copy behind the fp32 part when its K extent allows the split-operand kernel
    wf = ent[3] if same else torch.empty(lib().codeps_conv2d_packed_weight_floats(out_c_pad, in_c_pad, kh, kw, 0),
                                         dtype=torch.float32, device=w.device)
    wd = ent[4] if same else None
    if wd is None and need_dgrad:
        wd = torch.empty(lib().codeps_conv2d_packed_weight_floats(out_c_pad, in_c_pad, kh, kw, 1), dtype=torch.float32,
                         device=w.device)
    check(lib().codeps_conv2d_pack_weights(co, ci, kh, kw, out_c_pad, in_c_pad, ptr(w), ptr(wf), ptr(wd), stream_ptr()),
          "codeps_conv2d_pack_weights")
    cache[key] = [_weak(weight, cache, key), weight.data_ptr(), stamp, wf, wd, _mark_written()]
    return wf, wd


_pack_table = {"dev": None, "n": 0, "groups": 0, "entries": [], "epoch": -1}


def build_pack_table():
    """Gather the persistent packs of all live trainable nn.Parameters (the entries `pack_weights` has created so far) into ONE
    device table for `refresh_weight_packs`.  Call after a first step has touched every layer; packs created later, and the
    packs of per-step temporaries (re-laid-out stem / GUDA tap weights), keep the one-by-one path."""
    live = []
    for key, ent in list(_pack_cache_train.items()):
        w = ent[0]()
        if w is None or not isinstance(w, torch.nn.Parameter) or not w.is_contiguous() or w.dtype != torch.float32:
            continue
        live.append((key, ent, w))
    _pack_table.update(dev=None, n=0, groups=0, entries=[], epoch=-1)
    if not live:
        return 0
    item = lib().codeps_pack_item_bytes()
    n_items = sum(1 + (e[4] is not None) for _k, e, _w in live)
    host = ctypes.create_string_buffer(item * n_items)
    base = ctypes.addressof(host)
    i, groups = 0, 0
    for (key, ent, w) in live:
        co, ci, kh, kw = w.shape
        _id, cop, cip = key
        for dgrad, buf in ((0, ent[3]), (1, ent[4])):
            if buf is None:
                continue
            g = lib().codeps_pack_item_fill(ctypes.c_void_p(base + i * item), ptr(w), ptr(buf), co, ci, kh, kw, cop, cip, dgrad, groups)
            if g <= 0:
                raise CodepsHipError("codeps_pack_item_fill: " + lib().codeps_last_error().decode())
            groups += g
            i += 1
    dev = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(live[0][2].device)
    _pack_table.update(dev=dev, n=n_items, groups=groups, epoch=-1, arith=_arith(),
                       entries=[(e, w.data_ptr(), weakref.ref(w), e[4] is not None) for _k, e, w in live])
    _pack_table["builds"] = _pack_table.get("builds", 0) + 1
    return n_items


def refresh_weight_packs():
    """Re-fill every pack of the table for the current weights epoch in ONE launch (no-op when they are current).  Falls back
    to the one-by-one path (drops the table) as soon as a parameter of the table died or moved."""
    t = _pack_table
    if t["dev"] is not None and t.get("arith") != _arith():
        t.update(dev=None, n=0, groups=0, entries=[], epoch=-1)          # the table's items were filled for another arithmetic
    if t["dev"] is None or t["epoch"] == _weights_epoch[0]:
        return
    for ent, dptr, ref, had_wd in t["entries"]:
        w = ref()
        if w is None or w.data_ptr() != dptr or ent[0]() is not w or (ent[4] is not None) != had_wd:
            t.update(dev=None, n=0, groups=0, entries=[], epoch=-1)      # (a pack buffer the table does not know must not be stamped)
            return
    check(lib().codeps_conv2d_pack_weights_multi(ptr(t["dev"]), t["n"], t["groups"], stream_ptr()), "codeps_conv2d_pack_weights_multi")
    t["epoch"] = _weights_epoch[0]
    mark = _mark_written()
    for ent, _dptr, ref, _had in t["entries"]:
        ent[2] = (ref()._version, _weights_epoch[0], t["arith"])
        ent[5] = mark


_padvec_cache = {}


def _pad_vec(v, n):
    """Per-channel vector zero-padded to n entries.  A padded copy lives in a persistent buffer per source tensor and is
    re-filled by ONE copy when the source may have changed (tensor version / weights epoch): no per-call zeros + slice copy."""
    if v is None:
        return None
    src = v
    v = _f32c(v.detach())
    if v.numel() == n:
        return v
    key = (id(src), n)
    stamp = (v.data_ptr(), src._version, _weights_epoch[0])
    hit = _padvec_cache.get(key)
    if hit is not None and hit[4]() is src and hit[0] == stamp[0] and hit[3].device == v.device:
        _wait_written(hit[5])
        if (hit[1], hit[2]) != stamp[1:]:
            concat_rows([v.reshape(-1)], out=hit[3])               # (the padding lanes keep their zeros)
            hit[1], hit[2], hit[5] = stamp[1], stamp[2], _mark_written()
        return hit[3]
    out = torch.zeros(n, dtype=torch.float32, device=v.device)
    out[:v.numel()] = v
    _padvec_cache[key] = [stamp[0], stamp[1], stamp[2], out, _weak(src, _padvec_cache, key), _mark_written()]
    return out


def _conv_desc(x0, x1, out_c_pad, kh, kw, stride, pad, pad_mode, act, up0):
    b = x0.shape[0]
    c0 = x0.shape[3]
    c1 = 0 if x1 is None else x1.shape[3]
    if up0:
        h, w = x0.shape[1] * 2, x0.shape[2] * 2
    else:
        h, w = x0.shape[1], x0.shape[2]
    if x1 is not None and (x1.shape[1] != h or x1.shape[2] != w):
        raise _lib.CodepsHipError(f"conv2d: skip tensor is {tuple(x1.shape[1:3])}, expected {(h, w)}")
    (sh, sw), (ph, pw) = stride, pad
    ho, wo = (h + 2 * ph - kh) // sh + 1, (w + 2 * pw - kw) // sw + 1
    return _lib.ConvDesc(b, h, w, c0, c1, int(up0), out_c_pad, ho, wo, kh, kw, sh, ph, pad_mode, act, sw, pw)


ACT = {None: 0, "none": 0, "relu": 1, "elu": 2}

# Flat-gradient sink: the step harness (codeps_amd/step.py) registers, for every trainable conv weight / bias,
# a view into ONE flat fp32 gradient buffer.  The conv backward then accumulates straight into that view
# (wgrad / bias-reduce epilogues with accumulate=1) and reports no gradient to autograd -> no per-parameter
# accumulate kernels, and the buffer is ready for a single RCCL all-reduce.
_grad_sink = {}            # id(param) -> (weakref(param), view into the flat gradient buffer)

# Bias gradients whose column-sum partials a data-gradient epilogue left and whose destination is a slot of the flat gradient buffer: their
# finishes are collected per stream and run as ONE launch (codeps_colsum_finish_multi) when the step harness asks (StepRunner: after the
# backward pass, and before a gradient segment is handed to the all-reduce) -- 25 launches of 6 us per adaptation step were one each.
_DEFER_COLSUM = os.environ.get("CODEPS_DEFER_COLSUM", "1") != "0"
_colsum_pending = {}       # stream handle -> (torch stream, [(partials tensor, tiles, c, sink view, c_real)])
_colsum_armed = [False]    # only a harness that promises to flush arms the deferral


def arm_colsum_deferral(on):
    """StepRunner.forward_backward: True for the span of one backward pass (stale entries of an aborted pass are dropped)."""
    _colsum_pending.clear()
    _colsum_armed[0] = bool(on) and _DEFER_COLSUM


def _colsum_finish(part, tiles, c, out, c_real, accumulate):
    if accumulate and _colsum_armed[0]:
        st = torch.cuda.current_stream()
        items = _colsum_pending.setdefault(st.cuda_stream, (st, []))[1]
        if any(t[3].data_ptr() == out.data_ptr() for t in items):      # a module applied twice: its two finishes must not share a launch
            flush_colsums(current_only=True)
            items = _colsum_pending.setdefault(st.cuda_stream, (st, []))[1]
        items.append((part, tiles, c, out, c_real))
        return
    check(lib().codeps_colsum_finish(ptr(part), tiles, c, ptr(out), c_real, 1 if accumulate else 0, stream_ptr()), "codeps_colsum_finish")


def flush_colsums(current_only=False):
    """Run the pending bias-gradient finishes, each stream's on that stream (current_only: just the current stream's)."""
    if not _colsum_pending:
        return
    cur = torch.cuda.current_stream().cuda_stream
    for key in list(_colsum_pending.keys()):
        if current_only and key != cur:
            continue
        st, items = _colsum_pending.pop(key)
        with torch.cuda.stream(st):
            for i in range(0, len(items), 32):
                chunk = items[i:i + 32]
                n = len(chunk)
                parts = (ctypes.c_void_p * n)(*[t[0].data_ptr() for t in chunk])
                outs = (ctypes.c_void_p * n)(*[t[3].data_ptr() for t in chunk])
                tiles = (ctypes.c_int * n)(*[t[1] for t in chunk])
                cs = (ctypes.c_int * n)(*[t[2] for t in chunk])
                crs = (ctypes.c_int * n)(*[t[4] for t in chunk])
                check(lib().codeps_colsum_finish_multi(parts, tiles, cs, outs, crs, n, st.cuda_stream), "codeps_colsum_finish_multi")


def register_grad_sink(param, view):
    _grad_sink[id(param)] = (_weak(param, _grad_sink, id(param)), view)


def grad_sink(param):
    """Flat-gradient view registered for exactly this tensor object (ids are re-used once a tensor dies: the weak reference
    guards against a stale entry answering for a new parameter), else None."""
    if param is None:
        return None
    hit = _grad_sink.get(id(param))
    if hit is None:
        return None
    if hit[0]() is param:
        return hit[1]
    del _grad_sink[id(param)]
    return None

# Optional conv-engine profiler: {"events": [(kind, flops, start_evt, end_evt)]} filled while enabled (bench.py).
_profile = {"on": False, "events": []}


def _prof_begin():
    if not _profile["on"]:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _prof_end(start, kind, flops, desc=None, source=0, tag=None, kernel=None):
    """kernel: name of the kernel for launches that do not go through the conv engine's dispatcher (tools/conv_layers.py asks the
    library for the others)."""
    if start is None:
        return
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    if desc is not None:
        tag = (f"B{desc.batch} {desc.in_h}x{desc.in_w} {desc.c0}+{desc.c1}->{desc.out_c} k{desc.kh} s{desc.stride}"
               f"{' up' if desc.up0 else ''}")
        # arithmetic of the launch: two fp16 pieces (the kernel that ran says so in its name), three bf16 pieces, or the f32 MFMA
        if kind in ("conv_fwd", "conv_dgrad", "conv_wgrad") and "f16x2" in lib().codeps_conv2d_last_kernel().decode():
            kind += "_f16x2"
        elif kind == "conv_fwd" and lib().codeps_conv2d_arith(ctypes.byref(desc), 0, 0) > 0:
            kind = "conv_fwd_split"
        elif kind == "conv_dgrad" and lib().codeps_conv2d_arith(ctypes.byref(desc), 1, source) > 0:
            kind = "conv_dgrad_split"
        elif kind == "conv_wgrad" and lib().codeps_conv2d_arith(ctypes.byref(desc), 2, 0) > 0:
            kind = "conv_wgrad_split"
    _profile["events"].append((kind, flops, start, e, tag))


_UP_DGRAD16 = os.environ.get("CODEPS_UP_DGRAD16", "1") != "0"     # A/B switch: 0 = the 36-pair gather form of the data-gradient kernel


_EPI = os.environ.get("CODEPS_EPILOGUE", "1") != "0"       # A/B switch: 0 = activation / BatchNorm backward and statistics as separate passes
# ... and one by one: BatchNorm statistics from the forward epilogue / activation gates / the BatchNorm gate of the data gradient
_STEM_STATS = os.environ.get("CODEPS_STEM_STATS", "1") != "0"      # A/B switch: the stem convolution's statistics epilogue
_EPI_PARTS = {k: os.environ.get("CODEPS_EPI_" + k.upper(), "1") != "0" for k in ("stats", "act", "bn")}


class _Gate:
    """Travels as `._codeps_gate` on the OUTPUT of an activation -- conv + ELU / ReLU (`kind` "act") or train-mode BatchNorm + ReLU
    ("bn") -- and tells a convolution that consumes the tensor EXCLUSIVELY (`conv2d(..., exclusive=True)`: it, plus whoever takes its
    fork alias, are the tensor's only consumers) how to apply that activation's backward in its own data-gradient epilogue
    (codeps_conv_epilogue).  The consumer's backward leaves a hand-off record; the producer's backward, which autograd runs next with
    exactly that gradient tensor, finds its incoming gradient already gated (and its bias / BatchNorm sums as per-tile partials)."""
    __slots__ = ("kind", "act", "want_colsum", "has_res", "x", "mean", "invstd", "gamma", "beta", "groups", "handoff")

    def __init__(self, kind, act):
        # (the gated tensor itself is not kept here -- the consumer has it as its x0 -- so that the tensor, which carries this object,
        # and its grad_fn's context, which also does, form no reference cycle)
        self.kind, self.act = kind, act
        self.want_colsum = self.has_res = False
        self.x = self.mean = self.invstd = self.gamma = self.beta = self.groups = None
        self.handoff = None

    def take(self, dy):
        """The hand-off of this backward pass, if the consumer gated the gradient: checked against the tensor that arrived."""
        h, self.handoff = self.handoff, None
        if h is None:
            return None
        if h["ptr"] != dy.data_ptr() or h["version"] != dy._version or h["shape"] != tuple(dy.shape):
            raise _lib.CodepsHipError("a gradient gated in a convolution's epilogue reached its producer modified: the tensor passed with "
                                      "exclusive=True has another consumer (autograd summed a second gradient into it)")
        return h


class _ConvOpts:
    """Non-tensor options / results of one conv2d call: in -- in_gate (the gate of x0, consumed exclusively), bn_stats (the output feeds a
    train-mode BatchNorm: leave its batch statistics as per-tile partials); out -- gate (of the output, when it has an activation),
    stats (the partials record, None when the launch plan could not produce them)."""
    __slots__ = ("in_gate", "bn_stats", "gate", "stats", "in_defer")

    def __init__(self, in_gate=None, bn_stats=False, in_defer=None):
        self.in_gate, self.bn_stats, self.gate, self.stats, self.in_defer = in_gate, bn_stats, None, None, in_defer


def _epilogue_for(gate, x0, rows, cols, device):
    """codeps_conv_epilogue for a data-gradient launch whose result is the gradient of x0, the tensor `gate` came with."""
    e = _lib.ConvEpilogue()
    if gate.kind == "act":
        e.gate, e.gate_act = x0.data_ptr(), gate.act
        e.stat_mode = 2 if gate.want_colsum else 0
    else:
        e.gate = x0.data_ptr() if gate.has_res else None         # a residual fed the ReLU: the mask is (block output > 0)
        e.gate_act, e.stat_mode = 3, 3
        e.bn_x, e.bn_mean, e.bn_invstd = gate.x.data_ptr(), gate.mean.data_ptr(), gate.invstd.data_ptr()
        e.bn_gamma, e.bn_beta = gate.gamma.data_ptr(), gate.beta.data_ptr()
        e.bn_relu, e.bn_groups = 1, len(gate.groups)
        for i, (b0, _b1) in enumerate(gate.groups):
            e.bn_group_b0[i] = b0
    part = None
    if e.stat_mode:
        nbytes = lib().codeps_conv2d_epilogue_part_bytes(rows, cols, e.stat_mode)
        part = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
        e.stat_part, e.stat_part_bytes = part.data_ptr(), nbytes
    return e, part


def _handoff(gate, e, part, dx):
    """Record what the launch did (gate applied: bit 0; statistics written: bit 1) for the producer's backward."""
    if not (e.applied & 1):
        return
    h = {"ptr": dx.data_ptr(), "version": dx._version, "shape": tuple(dx.shape), "part": None}
    if e.applied & 2:
        h["part"] = part
        h["segs"] = [(e.seg_tile0[k], e.seg_tiles_per_image[k], e.seg_tile_rows[k], e.seg_tiles[k]) for k in range(e.n_seg)]
    gate.handoff = h


def _handoff_blocks(gate, dx, part, blocks):
    """Hand-off of a gate applied by a kernel outside the conv engine (upconv16 / disp_head data gradients): dx is gated; part
    [blocks][cols] = its column sums per block (None: the producer sums dx itself)."""
    gate.handoff = {"ptr": dx.data_ptr(), "version": dx._version, "shape": tuple(dx.shape), "part": part, "segs": [(0, 0, 0, blocks)]}


def _act_gate_of(x, exclusive):
    """The activation gate x carries, if the caller is x's only reader and the activation gates are switched on."""
    g = getattr(x, "_codeps_gate", None) if exclusive else None
    return g if (g is not None and g.kind == "act" and _EPI and _EPI_PARTS["act"]) else None


_DEFER = os.environ.get("CODEPS_BN_DEFER", "1") != "0"     # A/B switch: 0 = every train-mode BatchNorm writes its normalised tensor


def in_affine_apply(x, table):
    """relu(BatchNorm(x)) from a codeps_bn_in_affine table [B][4][C]: the apply pass a deferred BatchNorm skipped, for a consumer that
    cannot normalise on its operand load."""
    x = x.contiguous()
    b, c = x.shape[0], x.shape[-1]
    y = torch.empty_like(x)
    check(lib().codeps_in_affine_apply(ptr(x), ptr(table), b, x.numel() // c // b, c, ptr(y), stream_ptr()), "codeps_in_affine_apply")
    return y


class _Conv2d(torch.autograd.Function):
    """y = act(conv(cat(up(x0), x1), W) * scale + shift + residual) on NHWC tensors (C ABI conv2d)."""

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, scale, shift, residual, stride, pad, pad_mode, act, up0, macs_per_out, fork=False, opts=None):
        require_cuda(x0, x1, weight)
        co, ci, kh, kw = weight.shape
        cop = pad16(co)
        c_in = x0.shape[3] + (0 if x1 is None else x1.shape[3])
        if x1 is not None and c_in != ci:
            raise _lib.CodepsHipError("conv2d: concat sources must not carry channel padding")
        need_dgrad = ctx.needs_input_grad[0] or (x1 is not None and ctx.needs_input_grad[1])
        wf, wd = pack_weights(weight, cop, c_in, need_dgrad)
        desc = _conv_desc(x0, x1, cop, kh, kw, stride, pad, pad_mode, act, up0)
        sc = _pad_vec(scale, cop)
        sh = _pad_vec(shift if shift is not None else bias, cop)
        if residual is not None:
            residual = residual.contiguous()
        y = torch.empty(desc.batch, desc.out_h, desc.out_w, cop, dtype=torch.float32, device=x0.device)
        flops = 2.0 * desc.batch * desc.out_h * desc.out_w * co * (macs_per_out or ci * kh * kw)         # algorithmic (unpadded)
        t0 = _prof_begin()
        wsb = lib().codeps_conv2d_fwd_workspace_bytes(ctypes.byref(desc))          # split-K scratch (small-M, deep-K shapes only)
        ws = _workspace(wsb, x0.device, "splitk") if wsb else None
        # x0 = the INPUT of a train-mode BatchNorm + ReLU whose apply pass was left to this consumer (batch_norm_train(defer=True)): the kernel
        # normalises on its operand load where the shape allows (codeps_conv2d_in_affine_ok), else the tensor is made here after all
        table = opts.in_defer if opts is not None else None
        fused_in = False
        if table is not None:
            if x1 is None and not up0 and lib().codeps_conv2d_in_affine_ok(ctypes.byref(desc), 0):
                fused_in = True
            else:
                x0 = in_affine_apply(x0, table)
                table = None
        epi = spart = None
        if opts is not None and opts.bn_stats and _EPI and _EPI_PARTS["stats"] and not act and sc is None and sh is None and residual is None:
            # the output feeds a train-mode BatchNorm: (shift, sum, sum of squares) per M tile from the epilogue instead of a pass over y
            epi = _lib.ConvEpilogue()
            epi.stat_mode = 1
            nbytes = lib().codeps_conv2d_epilogue_part_bytes(desc.batch * desc.out_h * desc.out_w, cop, 1)
            spart = torch.empty(nbytes // 4, dtype=torch.float32, device=x0.device)
            epi.stat_part, epi.stat_part_bytes = spart.data_ptr(), nbytes
        if fused_in:
            check(lib().codeps_conv2d_fwd_in(ctypes.byref(desc), ptr(x0), ptr(table), ptr(wf), ptr(sc), ptr(sh), ptr(residual), ptr(y),
                                             ptr(ws), ws.numel() if ws is not None else 0, ctypes.byref(epi) if epi is not None else None,
                                             stream_ptr()), "codeps_conv2d_fwd_in")
        else:
            check(lib().codeps_conv2d_fwd_ex(ctypes.byref(desc), ptr(x0), ptr(x1), ptr(wf), ptr(sc), ptr(sh), ptr(residual), ptr(y),
                                             ptr(ws), ws.numel() if ws is not None else 0, ctypes.byref(epi) if epi is not None else None,
                                             stream_ptr()), "codeps_conv2d_fwd")
        _prof_end(t0, "conv_fwd", flops, desc)
        if epi is not None and (epi.applied & 2):
            opts.stats = {"part": spart, "tile0": epi.seg_tile0[0], "tpi": epi.seg_tiles_per_image[0], "rows": epi.seg_tile_rows[0]}
        # the gate of x0 (its producer's activation backward, applied in this launch's data-gradient epilogue) and the gate this
        # convolution's own activation offers to ITS exclusive consumer
        ctx.in_gate = opts.in_gate if (opts is not None and _EPI and ctx.needs_input_grad[0] and opts.in_gate is not None and
                                       _EPI_PARTS[opts.in_gate.kind]) else None
        ctx.out_gate = None
        # (grad mode is off inside forward(): needs_input_grad is what says whether a backward will run)
        if opts is not None and _EPI and act and scale is None and residual is None and \
                (ctx.needs_input_grad[0] or ctx.needs_input_grad[2] or (x1 is not None and ctx.needs_input_grad[1])):
            g = _Gate("act", act)
            g.want_colsum = bias is not None and ctx.needs_input_grad[3]
            ctx.out_gate = opts.gate = g
        ctx.desc, ctx.co, ctx.ci, ctx.has_bias = desc, co, ci, bias is not None
        ctx.flops = flops
        ctx.w_sink = grad_sink(weight) if weight.requires_grad else None
        ctx.b_sink = grad_sink(bias) if (bias is not None and bias.requires_grad) else None
        ctx.c0_real = min(ci, x0.shape[3])
        ctx.weight_ref = weakref.ref(weight)
        ctx.has_res = residual is not None
        ctx.save_for_backward(x0, x1, y if act else None, wd, sc, table)         # (table: x0 is still BatchNorm's input -- the weight gradient normalises too)
        ctx.fork = bool(fork)
        if fork:
            # second output = x0 itself (an alias): whatever else consumes the tensor (a residual connection, a down-sampling
            # shortcut) takes THIS output, so its gradient arrives here with dy and the data-gradient kernel adds it in its
            # epilogue -- instead of autograd summing two full-size tensors with a library kernel afterwards
            return y, x0
        return y

    @staticmethod
    def backward(ctx, dy, g_fork=None):
        x0, x1, y, wd, sc, in_table = ctx.saved_tensors
        desc = ctx.desc
        add = g_fork.contiguous() if (ctx.fork and g_fork is not None) else None
        gated = ctx.out_gate.take(dy) if ctx.out_gate is not None else None
        dy = dy.contiguous()
        rows, cop = dy.numel() // dy.shape[-1], dy.shape[-1]
        need_b = ctx.has_bias and ctx.needs_input_grad[3]
        dbias = None
        if need_b:
            dbias = ctx.b_sink if ctx.b_sink is not None else torch.empty(ctx.co, dtype=torch.float32, device=dy.device)
        b_acc = 1 if ctx.b_sink is not None else 0
        dres = None
        if gated is not None:
            # dy IS dz: the consumer's data-gradient epilogue multiplied by act'(y); the bias gradient from its per-tile column sums
            dz = dy
            if need_b:
                if gated["part"] is not None:
                    tiles = sum(sg[3] for sg in gated["segs"])
                    _colsum_finish(gated["part"], tiles, cop, dbias, ctx.co, b_acc)
                else:
                    wsb = lib().codeps_colsum_workspace_bytes(rows, cop, 1)
                    ws = _workspace(wsb, dy.device, "colsum")
                    check(lib().codeps_act_bwd(ptr(dy), None, None, 0, rows, cop, None, ptr(dbias), ctx.co, b_acc, ptr(ws), ws.numel(),
                                               stream_ptr()), "codeps_act_bwd (bias gradient)")
                if ctx.b_sink is not None:
                    dbias = None
        elif desc.act or sc is not None or need_b:
            wsb = lib().codeps_colsum_workspace_bytes(rows, cop, 1)
            ws = _workspace(wsb, dy.device, "colsum")
            if ctx.has_res and ctx.needs_input_grad[6] and sc is not None:
                dres = torch.empty_like(dy)
                check(lib().codeps_act_bwd(ptr(dy), ptr(y), None, desc.act, rows, cop, ptr(dres), None, ctx.co, 0, ptr(ws),
                                           ws.numel(), stream_ptr()), "codeps_act_bwd")
            dz = torch.empty_like(dy) if (desc.act or sc is not None) else dy
            check(lib().codeps_act_bwd(ptr(dy), ptr(y), ptr(sc), desc.act, rows, cop, ptr(dz) if dz is not dy else None,
                                       ptr(dbias), ctx.co, b_acc, ptr(ws), ws.numel(), stream_ptr()), "codeps_act_bwd")
            if ctx.b_sink is not None:
                dbias = None            # already accumulated into the flat gradient buffer
        else:
            dz = dy
        if ctx.has_res and ctx.needs_input_grad[6] and dres is None:
            dres = dz
        dx0 = dx1 = dw = None
        if ctx.needs_input_grad[0] and desc.pad_mode == 1:
            # reflect padding: data gradient w.r.t. the (virtual) padded tensor, then fold the border back
            p_ = desc.pad
            dpad = _lib.ConvDesc(desc.batch, desc.in_h + 2 * p_, desc.in_w + 2 * p_, desc.c0, 0, 0, desc.out_c, desc.out_h,
                                 desc.out_w, desc.kh, desc.kw, desc.stride, 0, 0, desc.act, desc.stride, 0)
            dxp = torch.empty(desc.batch, desc.in_h + 2 * p_, desc.in_w + 2 * p_, desc.c0, dtype=torch.float32, device=dy.device)
            check(lib().codeps_conv2d_dgrad(ctypes.byref(dpad), 0, ptr(dz), ptr(wd), ptr(dxp), None, 0, stream_ptr()),
                  "codeps_conv2d_dgrad")
            dx0 = torch.empty_like(x0)
            check(lib().codeps_reflect_pad_fold(ptr(dxp), desc.batch, desc.in_h, desc.in_w, desc.c0, p_, ptr(dx0), stream_ptr()),
                  "codeps_reflect_pad_fold")
        elif ctx.needs_input_grad[0] and _UP_DGRAD16 and desc.up0 and (desc.kh, desc.kw, desc.stride, desc.pad) == (3, 3, 1, 1) and \
                ctx.c0_real == x0.shape[3] and ctx.c0_real > 16 and ctx.weight_ref() is not None:
            # nearest x2 + 3x3: the gradient w.r.t. the half-resolution source is a 4x4 / stride-2 forward convolution of dz with
            # pre-summed taps (16 instead of 36 (sub-pixel, tap) pairs) -- and it runs on the forward kernels
            wsrc = ctx.weight_ref()
            wt = torch.empty(ctx.c0_real, ctx.co, 4, 4, dtype=torch.float32, device=dy.device)
            check(lib().codeps_up_dgrad_weight(ptr(_f32c(wsrc.detach())), ctx.co, ctx.ci, ctx.c0_real, ptr(wt), stream_ptr()),
                  "codeps_up_dgrad_weight")
            c0p = x0.shape[3]
            wf2, _ = pack_weights(wt, c0p, cop, False)
            d2 = _conv_desc(dz, None, c0p, 4, 4, (2, 2), (1, 1), 0, 0, False)
            dx0 = torch.empty_like(x0)
            t0 = _prof_begin()
            wsb = lib().codeps_conv2d_fwd_workspace_bytes(ctypes.byref(d2))
            ws2 = _workspace(wsb, dy.device, "splitk") if wsb else None
            e2 = p2 = None
            if ctx.in_gate is not None:
                e2, p2 = _epilogue_for(ctx.in_gate, x0, dx0.numel() // c0p, c0p, dy.device)
            # (the second gradient of a forked input rides as the forward kernel's residual: added before the gate)
            check(lib().codeps_conv2d_fwd_ex(ctypes.byref(d2), ptr(dz), None, ptr(wf2), None, None, ptr(add), ptr(dx0), ptr(ws2),
                                             ws2.numel() if ws2 is not None else 0, ctypes.byref(e2) if e2 is not None else None,
                                             stream_ptr()), "codeps_conv2d_fwd (up-conv data gradient)")
            add = None
            if e2 is not None:
                _handoff(ctx.in_gate, e2, p2, dx0)
            _prof_end(t0, "conv_dgrad", ctx.flops * (ctx.c0_real / ctx.ci) * (16.0 / 36.0), d2)      # the FLOPs this form performs
        elif ctx.needs_input_grad[0]:
            dx0 = torch.empty_like(x0)
            t0 = _prof_begin()
            wsb = lib().codeps_conv2d_dgrad_workspace_bytes(ctypes.byref(desc), 0)            # split-K scratch (small-M, deep-K shapes only)
            ws2 = _workspace(wsb, dy.device, "splitk") if wsb else None
            e2 = p2 = None
            if ctx.in_gate is not None:
                e2, p2 = _epilogue_for(ctx.in_gate, x0, dx0.numel() // dx0.shape[-1], dx0.shape[-1], dy.device)
            check(lib().codeps_conv2d_dgrad_ex(ctypes.byref(desc), 0, ptr(dz), ptr(wd), ptr(add), ptr(dx0), ptr(ws2),
                                               ws2.numel() if ws2 is not None else 0, ctypes.byref(e2) if e2 is not None else None,
                                               stream_ptr()), "codeps_conv2d_dgrad_ex")
            add = None                                             # consumed by the kernel's epilogue
            if e2 is not None:
                _handoff(ctx.in_gate, e2, p2, dx0)
            _prof_end(t0, "conv_dgrad", ctx.flops * (ctx.c0_real / ctx.ci), desc)
        if x1 is not None and ctx.needs_input_grad[1]:
            dx1 = torch.empty_like(x1)
            t0 = _prof_begin()
            wsb = lib().codeps_conv2d_dgrad_workspace_bytes(ctypes.byref(desc), 1)
            ws2 = _workspace(wsb, dy.device, "splitk") if wsb else None
            check(lib().codeps_conv2d_dgrad(ctypes.byref(desc), 1, ptr(dz), ptr(wd), ptr(dx1), ptr(ws2), ws2.numel() if ws2 is not None else 0,
                                            stream_ptr()), "codeps_conv2d_dgrad")
            _prof_end(t0, "conv_dgrad", ctx.flops * (1 - ctx.c0_real / ctx.ci), desc, 1)
        if ctx.needs_input_grad[2]:
            sink = ctx.w_sink
            dw = sink if sink is not None else torch.empty(ctx.co, ctx.ci, desc.kh, desc.kw, dtype=torch.float32, device=dy.device)
            wsb = lib().codeps_conv2d_wgrad_workspace_bytes(ctypes.byref(desc))
            ws = _workspace(wsb, dy.device, "wgrad")
            t0 = _prof_begin()
            if in_table is not None and lib().codeps_conv2d_in_affine_ok(ctypes.byref(desc), 1):
                check(lib().codeps_conv2d_wgrad_in(ctypes.byref(desc), ptr(x0), ptr(in_table), ptr(dz), ptr(dw), ctx.co, ctx.ci,
                                                   1 if sink is not None else 0, ptr(ws), ws.numel(), stream_ptr()), "codeps_conv2d_wgrad_in")
            else:
                xw = in_affine_apply(x0, in_table) if in_table is not None else x0         # (a weight-gradient kernel without the normalise-on-load)
                check(lib().codeps_conv2d_wgrad(ctypes.byref(desc), ptr(xw), ptr(x1), ptr(dz), ptr(dw), ctx.co, ctx.ci,
                                                1 if sink is not None else 0, ptr(ws), ws.numel(), stream_ptr()), "codeps_conv2d_wgrad")
            _prof_end(t0, "conv_wgrad", ctx.flops, desc)
            if sink is not None:
                dw = None               # accumulated in place into the flat gradient buffer
        if add is not None:                                        # (a data-gradient form without the fused add, or no data gradient)
            dx0 = add if dx0 is None else dx0.add_(add)
        wref = ctx.weight_ref()
        done = getattr(wref, "_codeps_grad_done", None) if wref is not None else None
        if done is not None:        # codeps_amd.step: this node is the last one of its module's backward (segmented gradient all-reduce)
            done()
        return dx0, dx1, dw, dbias, None, None, dres, None, None, None, None, None, None, None, None


def _pair(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


def conv2d(x0, weight, bias=None, *, x1=None, scale=None, shift=None, residual=None, stride=1, pad=0, pad_mode="zeros",
           act=None, up0=False, macs_per_out=None, fork=False, exclusive=False, bn_stats=False):
    """NHWC convolution.  x0 [B,H,W,C0p]; optional skip x1 concatenated along channels after an
    optional nearest-x2 upsample of x0; weight OIHW; stride / pad: int or (vertical, horizontal);
    returns [B,Ho,Wo,pad16(Cout)].  macs_per_out: algorithmic multiply-adds per output element for the profiler when
    the weight tensor is a re-laid-out form of the reference's (stem_conv2d).  fork=True -> (y, x0'): x0' is x0 for every
    other consumer of the tensor; the gradient they send back is added inside the data-gradient kernel (no separate sum).
    exclusive=True: this call (and the taker of its fork alias) is the ONLY consumer of x0 -- when x0 came out of a conv + activation or
    a train-mode BatchNorm + ReLU of this library, that producer's activation backward (and its bias / BatchNorm sums) then happens in
    this convolution's data-gradient epilogue (`_Gate`).  bn_stats=True: the result feeds a train-mode BatchNorm; `y._codeps_stats`
    then carries its batch statistics as per-tile partials (None when the launch plan cannot produce them).
    x0 with `._codeps_defer` (batch_norm_train(defer=True)) holds the INPUT of a train-mode BatchNorm + ReLU: the forward and weight-gradient
    kernels normalise on their operand load (1x1 shapes, codeps_conv2d_in_affine_ok) or the tensor is materialised here."""
    if fork and getattr(x0, "_codeps_defer", None) is not None:
        raise _lib.CodepsHipError("conv2d: a deferred BatchNorm output (batch_norm_train(defer=True)) has ONE reader; fork=True hands the tensor on")
    opts = _ConvOpts(getattr(x0, "_codeps_gate", None) if exclusive else None, bn_stats, getattr(x0, "_codeps_defer", None))
    out = _Conv2d.apply(x0, x1, weight, bias, scale, shift, residual, _pair(stride), _pair(pad),
                        1 if pad_mode == "reflect" else 0, ACT[act], bool(up0), macs_per_out, bool(fork), opts)
    y = out[0] if fork else out
    if opts.gate is not None:
        y._codeps_gate = opts.gate
    if opts.stats is not None:
        y._codeps_stats = opts.stats
    return out


_stem_cache = {}


class _StemWeight(torch.autograd.Function):
    """conv1 weight -> double-pixel form, one launch each way (C ABI: codeps_stem_weight_{fwd,bwd}); the gradient goes straight
    into the flat gradient buffer when the step harness registered one for the parameter."""

    @staticmethod
    def forward(ctx, weight):
        w = _f32c(weight.detach())
        co, ci = w.shape[:2]
        out = torch.empty(co, 16, 7, 4, dtype=torch.float32, device=w.device)
        check(lib().codeps_stem_weight_fwd(ptr(w), co, ci, ptr(out), stream_ptr()), "codeps_stem_weight_fwd")
        ctx.meta = (co, ci)
        ctx.sink = grad_sink(weight) if weight.requires_grad else None
        return out

    @staticmethod
    def backward(ctx, g):
        co, ci = ctx.meta
        sink = ctx.sink
        gw = sink if sink is not None else torch.empty(co, ci, 7, 7, dtype=torch.float32, device=g.device)
        check(lib().codeps_stem_weight_bwd(ptr(g.contiguous()), co, ci, 1 if sink is not None else 0, ptr(gw), stream_ptr()),
              "codeps_stem_weight_bwd")
        return None if sink is not None else gw


def stem_weight(weight):
    """ResNet conv1 weight [co, ci <= 8, 7, 7] -> its double-pixel form [co, 16, 7, 4]:
    W'[o][p * 8 + c][ky][j] = W[o][c][ky][2 j + p - 1]  (tap 2j + p - 1 = -1 and channels >= ci are zero).
    Differentiable, cached for frozen weights."""
    co, ci, kh, kw = weight.shape
    if (kh, kw) != (7, 7) or ci > 8:
        raise _lib.CodepsHipError("stem_weight: expects a 7x7 kernel over <= 8 input channels")
    require_cuda(weight)
    frozen = not (weight.requires_grad and torch.is_grad_enabled())
    if frozen:
        hit = _stem_cache.get(id(weight))
        if hit is not None and hit[0]() is weight and hit[1] == (weight.data_ptr(), weight._version, _weights_epoch[0] if weight.requires_grad else 0):
            return hit[2]
        with torch.no_grad():
            wp = _StemWeight.apply(weight)
        _stem_cache[id(weight)] = (_weak(weight, _stem_cache, id(weight)),
                                   (weight.data_ptr(), weight._version, _weights_epoch[0] if weight.requires_grad else 0), wp)
        return wp
    return _StemWeight.apply(weight)


def stem_conv2d(x, weight, *, scale=None, shift=None, act=None, bn_stats=False):
    """ResNet conv1 (7x7, stride 2, padding 3, <= 8 input channels; models/resnet_encoder.py:115) on the double-pixel
    stem layout: x [B, H, W + 6, 8] = images with STEM_XPAD zero columns (images_to_nhwc(..., 8, STEM_XPAD)).
    Two neighbouring pixels are one 16-channel pixel of a [B, H, (W + 6) / 2, 16] view and conv1 is a 7x4 convolution
    of stride (2, 1), padding (3, 0) over it: the GEMM K dimension is 448 instead of 784 (16-channel padded 7x7)."""
    b, h, wp, c = x.shape
    if c != 8 or wp % 2:
        raise _lib.CodepsHipError("stem_conv2d: expects [B, H, W + 6, 8] with even W")
    return conv2d(x.view(b, h, wp // 2, 16), stem_weight(weight), scale=scale, shift=shift, stride=(2, 1), pad=(3, 0), act=act,
                  macs_per_out=weight.shape[1] * 49, bn_stats=bn_stats and _STEM_STATS)


# ---------------------------------------------------------------------------------------- frozen 3-channel stem
_STEM3 = os.environ.get("CODEPS_STEM3", "1") != "0"      # A/B switch: 0 = the double-pixel form for the frozen 3-channel stem too
STEM3_XPAD = (3, 3)     # zero columns left / right of every row of 4-channel pixels in the frozen-stem layout (stem3_conv2d)
_stem3_cache = {}


def stem3_weight(weight):
    """Frozen ResNet conv1 weight [co, 3, 7, 7] -> [co, 32, 7, 1]: W'[o][4 j + c][ky] = W[o][c][ky][j] (j < 7, c < 3, else 0) -- the
    K layout of stem3_conv2d.  Cached per (tensor, version)."""
    key = id(weight)
    hit = _stem3_cache.get(key)
    # a trainable conv1 reaches this form during validation (eval mode under no_grad): the fused optimizer writes p.data through
    # raw pointers, so only the weights epoch tells that the values changed
    stamp = (weight.data_ptr(), weight._version, _weights_epoch[0] if _is_volatile(weight) else 0)
    if hit is not None and hit[0]() is weight and hit[1] == stamp:
        return hit[2]
    co, ci, kh, kw = weight.shape
    if (ci, kh, kw) != (3, 7, 7):
        raise _lib.CodepsHipError("stem3_weight: expects a [co, 3, 7, 7] weight")
    with torch.no_grad():
        w = _f32c(weight.detach())
        out = torch.zeros(co, 8, 4, 7, dtype=torch.float32, device=w.device)          # [o][j][c][ky]
        out[:, :7, :3, :] = w.permute(0, 3, 1, 2)                                        # W[o][c][ky][j] -> [o][j][c][ky]
        out = out.reshape(co, 32, 7, 1).contiguous()
    _stem3_cache[key] = (_weak(weight, _stem3_cache, key), stamp, out)
    return out


def stem3_conv2d(x4, weight, *, scale=None, shift=None, act=None):
    """FROZEN ResNet conv1 on 3-channel images (7x7, stride 2, padding 3; models/resnet_encoder.py:115), inference only.
    x4 [B, H, W + 6, 4]: the images as rows of 4-channel pixels with STEM3_XPAD zero columns (images_to_nhwc(..., 4, STEM3_XPAD)).
    Output pixel x reads the 8 stored pixels 2 x .. 2 x + 7 of 7 rows = 32 contiguous floats per row: conv1 is a 7 x 1 convolution
    over OVERLAPPING 32-float "pixels" that start every 8 floats (codeps_conv_desc.pix_pitch) -- K = 224 in 7 steps of 32
    instead of the double-pixel form's K = 448 in 28 steps of 16 (stem_conv2d, which the 6-channel pose stem keeps)."""
    require_cuda(x4, weight)
    b, h, wp, c = x4.shape
    w = wp - 6
    if c != 4 or w % 2 or h % 2:
        raise _lib.CodepsHipError("stem3_conv2d: expects [B, H, W + 6, 4] with even H and W")
    if torch.is_grad_enabled() and (weight.requires_grad or x4.requires_grad):
        raise _lib.CodepsHipError("stem3_conv2d is the frozen-encoder form (no gradients); use stem_conv2d")
    w32 = stem3_weight(weight)
    co = weight.shape[0]
    cop = pad16(co)
    wf, _ = pack_weights(w32, cop, 32, False)
    desc = _lib.ConvDesc(b, h, w // 2, 32, 0, 0, cop, h // 2, w // 2, 7, 1, 2, 3, 0, ACT[act], 1, 0, 8, wp // 2)
    sc, sh = _pad_vec(scale, cop), _pad_vec(shift, cop)
    y = torch.empty(b, h // 2, w // 2, cop, dtype=torch.float32, device=x4.device)
    flops = 2.0 * b * (h // 2) * (w // 2) * co * 147
    t0 = _prof_begin()
    check(lib().codeps_conv2d_fwd(ctypes.byref(desc), ptr(x4.contiguous()), None, ptr(wf), ptr(sc), ptr(sh), None, ptr(y), None, 0, stream_ptr()),
          "codeps_conv2d_fwd (frozen stem)")
    _prof_end(t0, "conv_fwd", flops, desc)
    return y


# ---------------------------------------------------------------------------------------- chained 1x1 pair (frozen bottlenecks)
_chain_cache = {}
_CHAIN = os.environ.get("CODEPS_CHAIN1X1", "1") != "0"       # A/B switch: 0 = the two layers as separate conv launches


def _chain_perm(n2, device):
    """k order of the reduce weights inside every 32-channel group: position 8 kg + j holds channel 16 (j >> 2) + 4 kg + (j & 3)
    (the order in which an MFMA accumulator tile hands its channels over as the next MFMA's B operand, csrc/chain1x1.hip)."""
    p = torch.arange(32, device=device)
    inner = 16 * ((p & 7) >> 2) + 4 * (p >> 3) + (p & 3)
    return (torch.arange(0, n2, 32, device=device).view(-1, 1) + inner.view(1, -1)).reshape(-1)


def _pack_regions(wf, total):
    """Byte views of a pack buffer (csrc/conv.hip, pack_off_*): the three-bf16-piece copy, the two-fp16-piece copy, the scale slot."""
    b = wf.view(torch.uint8)
    off16 = (total + (total * 6 + 3) // 4) * 4
    return b[total * 4:][:total * 6], b[off16:][:total * 4], wf[(off16 // 4) + total:][:4]


def chain_weights(w_expand, scale2, shift2, w_reduce, scale3=None):
    """Weight blobs of codeps_chain1x1_expand_reduce for FROZEN 1x1 weights (cached per parameter pair / version):
    w2 rows = the pre-split expand pack row (8 x [3 pieces][32 bf16]) + 16 bytes (scale, shift, 0, 0); w3 = the pre-split pack
    of the reduce weights with the permuted k order.  Under arithmetic 16 (two fp16 pieces, codeps_chain1x1_expand_reduce_h) the rows
    hold 8 x [2 pieces][32 fp16], the affine's scale carries the pack's 2^-kb, and a third result is scale3 x the reduce pack's 2^-kb."""
    h16 = _arith() == 16
    key = (id(w_expand), id(w_reduce), h16)
    # trainable (or teacher-view) weights are rewritten through raw pointers by the fused optimizer: their entry lives for one
    # weights epoch.  The folded BatchNorm is identified by OBJECT (the entry keeps scale2 / shift2 alive, so their addresses
    # cannot be handed to a newer fold while the entry exists)
    volatile = _is_volatile(w_expand) or _is_volatile(w_reduce) or _is_volatile(scale2) or _is_volatile(shift2)
    stamp = (w_expand._version, w_reduce._version, scale2._version, shift2._version, w_expand.data_ptr(), w_reduce.data_ptr(),
             _weights_epoch[0] if volatile else 0, None if scale3 is None else (scale3.data_ptr(), scale3._version))
    hit = _chain_cache.get(key)
    if hit is not None and hit[0] == stamp and hit[3]() is w_expand and hit[4]() is w_reduce and hit[5] is scale2 and hit[6] is shift2:
        return (hit[1], hit[2], hit[7]) if h16 else (hit[1], hit[2])
    n2, kmid = w_expand.shape[:2]
    c3 = w_reduce.shape[0]
    if (kmid not in ((256, 128, 64) if h16 else (256,)) or c3 != kmid or w_reduce.shape[1] != n2 or n2 % 64 or tuple(w_expand.shape[2:]) != (1, 1)
            or tuple(w_reduce.shape[2:]) != (1, 1)):
        raise CodepsHipError("chain_weights: expects 1x1 weights [N2, K, 1, 1] and [K, N2, 1, 1] with N2 % 64 == 0 and K = 256 "
                             "(128 and 64 as well on the fp16-pair arithmetic)")
    dev = w_expand.device
    with torch.no_grad():
        wf, _ = pack_weights(w_expand, n2, kmid, False)
        s3, s2h, slot2 = _pack_regions(wf, n2 * kmid)
        aff = torch.zeros(n2, 4, dtype=torch.float32, device=dev)
        aff[:, 0], aff[:, 1] = _f32c(scale2.detach()), _f32c(shift2.detach())
        if h16:
            aff[:, 0] *= slot2[0]                                 # 2^-kb of the expand pack (a device scalar: no sync)
        rowb = kmid * 4 + 16 if h16 else lib().codeps_chain1x1_w2_row_bytes()
        split = s2h.view(n2, kmid * 4) if h16 else s3.view(n2, kmid * 6)
        w2 = torch.cat([split, aff.view(torch.uint8).view(n2, 16)], 1).contiguous()
        assert w2.shape[1] == rowb
        w3p = _f32c(w_reduce.detach())[:, _chain_perm(n2, dev)].contiguous()
        wf3, _ = pack_weights(w3p, c3, n2, False)
        r3, r2h, slot3 = _pack_regions(wf3, c3 * n2)
        w3 = (r2h if h16 else r3).clone()
        sc3 = (_f32c(scale3.detach()) * slot3[0]).contiguous() if (h16 and scale3 is not None) else None
    _chain_cache[key] = (stamp, w2, w3, _weak(w_expand, _chain_cache, key), weakref.ref(w_reduce), scale2, shift2, sc3)
    return (w2, w3, sc3) if h16 else (w2, w3)


def chain_expand_reduce(y1, residual, w_expand, scale2, shift2, w_reduce, scale3, shift3):
    """Inference-only fusion of two frozen 1x1 layers of consecutive ResNet bottlenecks (models/resnet_encoder.py:115-126):
        out = relu(bn3(conv3(y1)) + residual)            [B,H,W,N2]
        y0  = relu(bn1'(conv1'(out)))                    [B,H,W,K]      (the next bottleneck's first layer; K = y1's width)
    in ONE launch (C ABI: codeps_chain1x1_expand_reduce / _hk); the BatchNorms arrive folded as (scale, shift)."""
    require_cuda(y1, residual, w_expand, w_reduce)
    if torch.is_grad_enabled() and (y1.requires_grad or residual.requires_grad or w_expand.requires_grad or w_reduce.requires_grad):
        raise CodepsHipError("chain_expand_reduce is an inference kernel (frozen encoder): no autograd support")
    b, h, w_, kmid = y1.shape
    n2 = w_expand.shape[0]
    rows = b * h * w_
    h16 = _arith() == 16
    if h16:
        w2, w3, sc3 = chain_weights(w_expand, scale2, shift2, w_reduce, scale3)
    else:
        w2, w3 = chain_weights(w_expand, scale2, shift2, w_reduce)
        sc3 = _f32c(scale3)
    y1, residual = y1.contiguous(), residual.contiguous()
    out = torch.empty(b, h, w_, n2, dtype=torch.float32, device=y1.device)
    y0 = torch.empty(b, h, w_, kmid, dtype=torch.float32, device=y1.device)
    flops = 2.0 * rows * (n2 * kmid + kmid * n2)
    t0 = _prof_begin()
    wsb = lib().codeps_chain1x1_workspace_bytes_k(rows, n2, kmid)
    ws = _workspace(wsb, y1.device, "chain1x1") if wsb else None
    tail = (ptr(out), ptr(y0), ptr(ws), ws.numel() if ws is not None else 0, stream_ptr())
    head = (ptr(y1), ptr(residual), ptr(w2), ptr(w3), ptr(sc3), ptr(_f32c(shift3)), rows, n2)
    if h16:
        check(lib().codeps_chain1x1_expand_reduce_hk(*head, kmid, *tail), "codeps_chain1x1_expand_reduce_hk")
    else:
        check(lib().codeps_chain1x1_expand_reduce(*head, *tail), "codeps_chain1x1_expand_reduce")
    _prof_end(t0, "conv_fwd_f16x2" if h16 else "conv_fwd_split", flops, tag=f"B{b} {h}x{w_} {kmid}->{n2}->{kmid} k1 chain",
              kernel=f"chain_expand_reduce_h_kernel<{kmid}>" if h16 else "chain_expand_reduce_kernel")
    return out, y0


_CHAIN_NARROW = os.environ.get("CODEPS_CHAIN_NARROW", "1") != "0"    # A/B switch: 0 = mid widths 128 / 64 as separate launches


def chain_ok(y1, n2):
    """Shapes the chained kernel takes: 256 mid channels (128 and 64 too on the fp16-pair arithmetic), N2 % 64 == 0, pixel count a
    multiple of 128."""
    k = y1.shape[-1]
    return (_CHAIN and (k == 256 or (k in (128, 64) and _CHAIN_NARROW and _arith() == 16)) and n2 % 64 == 0
            and (y1.numel() // k) % 128 == 0)


# ---------------------------------------------------------------------------------------- BatchNorm
def bn_fold(weight, bias, running_mean, running_var, eps):
    """Eval-mode BatchNorm as a per-channel affine (scale, shift) for the conv epilogue."""
    scale = weight.detach() * torch.rsqrt(running_var + eps)
    return scale, bias.detach() - running_mean * scale


def _group_arrays(groups, per_item):
    n = len(groups)
    arr = ctypes.c_longlong * n
    return n, arr(*[g[0] * per_item for g in groups]), arr(*[(g[1] - g[0]) * per_item for g in groups])


def _img_groups(groups):
    n = len(groups)
    arr = ctypes.c_int * n
    return n, arr(*[g[0] for g in groups]), arr(*[g[1] for g in groups])


class _BatchNormTrain(torch.autograd.Function):
    """Train-mode BatchNorm (+ residual, + ReLU) on NHWC with batch statistics (C ABI: codeps_bn_{fwd,bwd}_mg);
    statistics pass through `reduce_fn` (cross-rank combination for SyncBN; None = local).

    `groups` = list of (b0, b1) batch ranges with INDEPENDENT statistics: several reference-style calls (each with
    its own batch statistics and its own running-stat momentum update, applied in list order) share one set of
    batched convolutions -- e.g. the six pose-net calls of an adaptation step run as one batch of 12 frame pairs.
    All groups are handled by one launch per phase."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, momentum, eps, act, reduce_fn, groups, opts=None):
        require_cuda(x, gamma, beta)
        x = x.contiguous()
        cp = x.shape[-1]
        c = gamma.numel()
        bsz = x.shape[0]
        per_item = x.numel() // cp // bsz
        plan = getattr(groups, "plan", None) if reduce_fn is not None else None
        groups = [(0, bsz)] if not groups else list(groups)
        ng, row0, rows = _group_arrays(groups, per_item)
        g, b_ = _pad_vec(gamma, cp), _pad_vec(beta, cp)
        if residual is not None:
            residual = residual.contiguous()
        dev = x.device
        y = torch.empty_like(x)
        mean = torch.empty(ng, cp, dtype=torch.float32, device=dev)
        var, invstd = torch.empty_like(mean), torch.empty_like(mean)
        ws = _workspace(lib().codeps_bn_mg_workspace_bytes(ng, cp), dev, "bn")
        counts = [float((b1 - b0) * per_item) for b0, b1 in groups]
        rm = running_mean if (running_mean is not None and running_mean.numel() == c) else None
        tiles = opts.stats if (opts is not None and _EPI) else None       # statistics partials the producing convolution's epilogue left
        # defer: the apply pass is left to the consumer (a 1x1 convolution that normalises on its operand load): no y, a [B][4][C] table
        defer = bool(opts is not None and opts.want_defer and _DEFER and act == 1 and residual is None and len(groups) <= 8)
        table = torch.empty(bsz, 4, cp, dtype=torch.float32, device=dev) if defer else None
        if defer:
            y = None

        def apply_pass(mean_, var_, invstd_, run_m, run_v):
            if defer:
                ngi, gb0, gb1 = _img_groups(groups)
                check(lib().codeps_bn_in_affine(ptr(mean_), ptr(var_), ptr(invstd_), ptr(g), ptr(b_), ngi, gb0, gb1, bsz, per_item, cp, c, momentum,
                                                ptr(run_m), ptr(run_v) if run_m is not None else None, ptr(table), stream_ptr()), "codeps_bn_in_affine")
            else:
                check(lib().codeps_bn_fwd_mg(ptr(x), ng, row0, rows, cp, c, ptr(g), ptr(b_), ptr(residual), act, eps, momentum, ptr(run_m),
                                             ptr(run_v) if run_m is not None else None, ptr(mean_), ptr(var_), ptr(invstd_), ptr(y), 0, 1,
                                             ptr(ws), ws.numel(), stream_ptr()), "codeps_bn_fwd_mg")

        def local_stats(mean_o, var_o, invstd_o):
            """per-group (mean, biased var, invstd) of THIS rank's rows"""
            if tiles is not None:
                ngi, gb0, gb1 = _img_groups(groups)
                check(lib().codeps_bn_finalize_tiles(ptr(tiles["part"]), tiles["tile0"], tiles["tpi"], tiles["rows"], ngi, gb0, gb1, cp, eps,
                                                     ptr(mean_o), ptr(var_o), ptr(invstd_o), stream_ptr()), "codeps_bn_finalize_tiles")
            else:
                check(lib().codeps_bn_fwd_mg(ptr(x), ng, row0, rows, cp, c, None, None, None, act, eps, momentum, None, None, ptr(mean_o),
                                             ptr(var_o), ptr(invstd_o), None, 1, 0, ptr(ws), ws.numel(), stream_ptr()), "codeps_bn_fwd_mg")

        if reduce_fn is None and tiles is not None and defer and (ng == 1 or rm is None):
            ngi, gb0, gb1 = _img_groups(groups)      # statistics finish + table (+ the running statistics of a one-group layer) in ONE launch
            check(lib().codeps_bn_finalize_tiles_table(ptr(tiles["part"]), tiles["tile0"], tiles["tpi"], tiles["rows"], ngi, gb0, gb1, cp, c, eps, momentum,
                                                       ptr(g), ptr(b_), ptr(mean), ptr(var), ptr(invstd), ptr(rm), ptr(running_var) if rm is not None else None,
                                                       ptr(table), stream_ptr()), "codeps_bn_finalize_tiles_table")
        elif reduce_fn is None and tiles is not None:
            local_stats(mean, var, invstd)
            # (apply_only with the running statistics: their momentum updates ride in the apply pass)
            apply_pass(mean, var, invstd, rm, running_var)
        elif reduce_fn is None and defer:
            check(lib().codeps_bn_fwd_mg(ptr(x), ng, row0, rows, cp, c, None, None, None, act, eps, momentum, ptr(rm),
                                         ptr(running_var) if rm is not None else None, ptr(mean), ptr(var), ptr(invstd), None, 1, 0,
                                         ptr(ws), ws.numel(), stream_ptr()), "codeps_bn_fwd_mg")        # (statistics pass: the momentum updates ride in its finish)
            apply_pass(mean, var, invstd, None, None)
        elif reduce_fn is None:
            check(lib().codeps_bn_fwd_mg(ptr(x), ng, row0, rows, cp, c, ptr(g), ptr(b_), ptr(residual), act, eps, momentum, ptr(rm),
                                         ptr(running_var) if rm is not None else None, ptr(mean), ptr(var), ptr(invstd), ptr(y), 0, 0,
                                         ptr(ws), ws.numel(), stream_ptr()), "codeps_bn_fwd_mg")
        elif plan is not None:
            # SyncBN over UNEQUAL shards (BASELINE configs[3]: ranks hold different numbers of items of a key, or none): the
            # packet has one slot per GLOBAL statistic group, this rank fills the slots of the groups it holds, and the combine
            # kernel weighs every rank's slot with its element count (0 = ignored)
            G, l2g = plan["n_global"], plan["index"](dev)
            pk_local = torch.empty(2, ng, cp, dtype=torch.float32, device=dev)
            local_stats(pk_local[0], pk_local[1], invstd)
            pk = torch.zeros(2, G, cp, dtype=torch.float32, device=dev)
            pk.index_copy_(1, l2g, pk_local)
            gathered = reduce_fn.gather(pk)                                     # [world, 2, G, cp]
            world = gathered.shape[0]
            rank_counts = [plan["items"][r][g] * per_item for r in range(world) for g in range(G)]
            gcounts = [sum(plan["items"][r][g] for r in range(world)) * per_item for g in range(G)]
            mean_g = torch.empty(G, cp, dtype=torch.float32, device=dev)
            var_g, invstd_g = torch.empty_like(mean_g), torch.empty_like(mean_g)
            check(lib().codeps_bn_sync_combine(ptr(gathered), world, G, (ctypes.c_float * G)(*gcounts), (ctypes.c_float * (world * G))(*rank_counts),
                                               cp, c, eps, momentum, ptr(rm), ptr(running_var) if rm is not None else None, ptr(mean_g),
                                               ptr(var_g), ptr(invstd_g), stream_ptr()), "codeps_bn_sync_combine")
            mean, var, invstd = mean_g.index_select(0, l2g), var_g.index_select(0, l2g), invstd_g.index_select(0, l2g)
            counts = [gcounts[g] for g in plan["local_to_global"]]
            apply_pass(mean, var, invstd, None, None)
        else:       # SyncBN: local statistics -> all-gather -> one combine kernel (merge + running stats + invstd) -> apply
            pk = torch.empty(2, ng, cp, dtype=torch.float32, device=dev)        # [mean | var] = the all-gather packet, no packing pass
            local_stats(pk[0], pk[1], invstd)
            gathered = reduce_fn.gather(pk)                                     # [world, 2, ng, cp], ONE collective per layer
            world = gathered.shape[0]
            counts = [n * world for n in counts]                                # equal shards on every rank by construction
            gc = (ctypes.c_float * ng)(*counts)
            check(lib().codeps_bn_sync_combine(ptr(gathered), world, ng, gc, None, cp, c, eps, momentum, ptr(rm),
                                               ptr(running_var) if rm is not None else None, ptr(mean), ptr(var), ptr(invstd),
                                               stream_ptr()), "codeps_bn_sync_combine")
            apply_pass(mean, var, invstd, None, None)
        ctx.inv_count = [1.0 / n for n in counts]
        # ReLU without a residual: the backward recomputes the mask from x (y is neither kept nor re-read)
        ctx.save_for_backward(x, y if (act and residual is not None) else None, g, b_, mean, invstd)
        ctx.act, ctx.c, ctx.reduce_fn, ctx.has_res, ctx.groups, ctx.per_item = act, c, reduce_fn, residual is not None, groups, per_item
        ctx.plan = plan
        # BatchNorm + ReLU: what an exclusive consumer needs to run the mask and the backward sums in its data-gradient epilogue
        ctx.out_gate = None
        if opts is not None and _EPI and act == 1 and ctx.needs_input_grad[0] and len(groups) <= 8:     # (codeps_conv_epilogue carries 8 statistic groups)
            gt = _Gate("bn", 3)
            gt.has_res = residual is not None
            gt.x, gt.mean, gt.invstd, gt.gamma, gt.beta, gt.groups = x, mean, invstd, g, b_, groups
            ctx.out_gate = opts.gate = gt
        # flat-gradient sinks of the affine parameters (codeps_amd.step): the backward accumulates dgamma / dbeta in place
        gs, bs = grad_sink(gamma), grad_sink(beta)
        ctx.sinks = (gs, bs) if (gs is not None and bs is not None and gamma.requires_grad and beta.requires_grad) else None
        if defer:
            opts.defer = table
            return x                        # (an alias: BatchNorm's input, raw -- batch_norm_train marks it with the table)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, g, b_, mean, invstd = ctx.saved_tensors
        gated = ctx.out_gate.take(dy) if ctx.out_gate is not None else None
        dy = dy.contiguous()
        cp = x.shape[-1]
        ng, row0, rows = _group_arrays(ctx.groups, ctx.per_item)
        inv_count = (ctypes.c_float * ng)(*ctx.inv_count)
        dev = x.device
        sums = torch.empty(ng, 2, cp, dtype=torch.float32, device=dev)
        if ctx.sinks is not None:
            dgamma, dbeta, acc = ctx.sinks[0], ctx.sinks[1], 1
        else:
            dgamma = torch.empty(ctx.c, dtype=torch.float32, device=dev)
            dbeta, acc = torch.empty_like(dgamma), 0
        dx = torch.empty_like(x)
        ws = _workspace(lib().codeps_bn_mg_workspace_bytes(ng, cp), dev, "bn")
        if gated is not None:
            # dy arrived masked (relu') from the consumer's data-gradient epilogue, with (sum g, sum g * xhat) per tile: the reduce pass
            # is a finish over the partials, the apply pass reads dy without a mask, and the residual's gradient IS dy
            segs = gated["segs"]
            ns = len(segs)
            iarr = ctypes.c_int * ns
            ngi, gb0, gb1 = _img_groups(ctx.groups)
            t0s, tpis, trs = iarr(*[sg[0] for sg in segs]), iarr(*[sg[1] for sg in segs]), iarr(*[sg[2] for sg in segs])

            def tiles_pass(phase, global_sums):
                check(lib().codeps_bn_bwd_tiles(ptr(gated["part"]), ns, t0s, tpis, trs, ng, row0, rows, gb0, gb1, ptr(dy), ptr(x), cp, ctx.c,
                                                ptr(mean), ptr(invstd), ptr(g), inv_count, ptr(sums), ptr(global_sums), ptr(dgamma), ptr(dbeta),
                                                acc, ptr(dx), phase, stream_ptr()), "codeps_bn_bwd_tiles")
            if ctx.reduce_fn is None:
                tiles_pass(0, None)
            else:
                tiles_pass(1, None)
                if ctx.plan is not None:
                    l2g = ctx.plan["index"](dev)
                    sums_g = torch.zeros(ctx.plan["n_global"], 2, cp, dtype=torch.float32, device=dev)
                    sums_g.index_copy_(0, l2g, sums)
                    gsums = ctx.reduce_fn.sum(sums_g).index_select(0, l2g)
                else:
                    gsums = ctx.reduce_fn.sum(sums.clone())          # (the local sums stay: dgamma / dbeta are this rank's)
                tiles_pass(2, gsums)
            if ctx.sinks is not None:
                dgamma = dbeta = None
            return dx, dgamma, dbeta, (dy if (ctx.has_res and ctx.needs_input_grad[3]) else None), None, None, None, None, None, None, None, None
        dres = torch.empty_like(x) if (ctx.has_res and ctx.needs_input_grad[3]) else None
        if ctx.reduce_fn is None:
            check(lib().codeps_bn_bwd_mg(ptr(dy), ptr(y), ptr(x), ng, row0, rows, cp, ctx.c, ptr(mean), ptr(invstd), ptr(g), ptr(b_), inv_count,
                                         ctx.act, ptr(sums), ptr(dgamma), ptr(dbeta), acc, ptr(dx), ptr(dres), 0, 0, ptr(ws), ws.numel(),
                                         stream_ptr()), "codeps_bn_bwd_mg")
        else:
            check(lib().codeps_bn_bwd_mg(ptr(dy), ptr(y), ptr(x), ng, row0, rows, cp, ctx.c, ptr(mean), ptr(invstd), ptr(g), ptr(b_), None,
                                         ctx.act, ptr(sums), ptr(dgamma), ptr(dbeta), acc, None, None, 1, 0, ptr(ws), ws.numel(), stream_ptr()),
                  "codeps_bn_bwd_mg")
            if ctx.plan is not None:           # unequal shards: sum the groups in their global slots
                l2g = ctx.plan["index"](dev)
                sums_g = torch.zeros(ctx.plan["n_global"], 2, cp, dtype=torch.float32, device=dev)
                sums_g.index_copy_(0, l2g, sums)
                sums = ctx.reduce_fn.sum(sums_g).index_select(0, l2g)
            else:
                sums = ctx.reduce_fn.sum(sums)
            check(lib().codeps_bn_bwd_mg(ptr(dy), ptr(y), ptr(x), ng, row0, rows, cp, ctx.c, ptr(mean), ptr(invstd), ptr(g), ptr(b_), inv_count,
                                         ctx.act, ptr(sums), ptr(dgamma), ptr(dbeta), 0, ptr(dx), ptr(dres), 0, 1, ptr(ws), ws.numel(),
                                         stream_ptr()), "codeps_bn_bwd_mg")
        if ctx.sinks is not None:
            dgamma = dbeta = None           # already accumulated into the flat gradient buffer
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None, None


class _BatchNormEval(torch.autograd.Function):
    """Eval-mode BatchNorm (running statistics) as its own op, used only when the BN affine parameters
    need gradients (otherwise eval BN is folded into the producing conv's epilogue)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, eps, act):
        require_cuda(x, gamma, beta)
        x = x.contiguous()
        cp = x.shape[-1]
        rows = x.numel() // cp
        c = gamma.numel()
        mean = _pad_vec(running_mean, cp)
        invstd = _pad_vec(torch.rsqrt(running_var.detach() + eps), cp)
        g, b_ = _pad_vec(gamma, cp), _pad_vec(beta, cp)
        if residual is not None:
            residual = residual.contiguous()
        y = torch.empty_like(x)
        check(lib().codeps_bn_apply(ptr(x), ptr(mean), ptr(invstd), ptr(g), ptr(b_), ptr(residual), act, rows, cp, ptr(y),
                                    stream_ptr()), "codeps_bn_apply")
        ctx.save_for_backward(x, y if act else None, mean, invstd, g)
        ctx.act, ctx.c, ctx.has_res = act, c, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, invstd, g = ctx.saved_tensors
        dy = dy.contiguous()
        cp = x.shape[-1]
        rows = x.numel() // cp
        sums = torch.empty(2, cp, dtype=torch.float32, device=x.device)
        ws = _workspace(lib().codeps_colsum_workspace_bytes(rows, cp, 2), x.device, "colsum")
        check(lib().codeps_bn_bwd_reduce(ptr(dy), ptr(y), ptr(x), ptr(mean), ptr(invstd), ctx.act, rows, cp, ptr(sums), ptr(ws),
                                         ws.numel(), stream_ptr()), "codeps_bn_bwd_reduce")
        dgamma, dbeta = sums[1, :ctx.c].clone(), sums[0, :ctx.c].clone()
        zero = torch.zeros_like(sums)                      # fixed statistics: no mean / variance terms in dx
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if (ctx.has_res and ctx.needs_input_grad[3]) else None
        check(lib().codeps_bn_bwd_apply(ptr(dy), ptr(y), ptr(x), ptr(mean), ptr(invstd), ptr(g), ptr(zero), 0.0, ctx.act, rows, cp,
                                        ptr(dx), ptr(dres), stream_ptr()), "codeps_bn_bwd_apply")
        return dx, dgamma, dbeta, dres, None, None, None, None


def batch_norm_eval(x, gamma, beta, running_mean, running_var, eps=1e-5, residual=None, act=None):
    return _BatchNormEval.apply(x, gamma, beta, residual, running_mean, running_var, float(eps), ACT[act])


class _BnOpts:
    __slots__ = ("stats", "gate", "want_defer", "defer")

    def __init__(self, stats=None, want_defer=False):
        self.stats, self.gate, self.want_defer, self.defer = stats, None, want_defer, None


def batch_norm_train(x, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5, residual=None, act=None,
                     reduce_fn=None, groups=None, defer=False):
    """x may carry `._codeps_stats` (conv2d(..., bn_stats=True)): the statistics pass then is a finish over per-tile partials.  A ReLU
    output carries `._codeps_gate` for a convolution that consumes it exclusively (conv2d(..., exclusive=True)).
    defer=True (BatchNorm + ReLU without a residual whose ONLY reader is the next F.conv2d call): the apply pass is not run; the result is
    BatchNorm's INPUT marked with `._codeps_defer` (a [B][4][C] table of mean / invstd / gamma / beta per image) and conv2d normalises
    on its operand load, or materialises the tensor itself when its kernels cannot.  Nothing but F.conv2d may read such a tensor."""
    opts = _BnOpts(getattr(x, "_codeps_stats", None), defer)
    y = _BatchNormTrain.apply(x, gamma, beta, residual, running_mean, running_var, float(momentum), float(eps), ACT[act],
                              reduce_fn, groups, opts)
    if opts.gate is not None:
        y._codeps_gate = opts.gate
    if opts.defer is not None:
        y._codeps_defer = opts.defer
    return y


# ---------------------------------------------------------------------------------------- pooling / resize
class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        require_cuda(x)
        x = x.contiguous()
        b, h, w, c = x.shape
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = torch.empty(b, ho, wo, c, dtype=torch.float32, device=x.device)
        arg = torch.empty(b, ho, wo, c, dtype=torch.uint8, device=x.device) if ctx.needs_input_grad[0] else None
        check(lib().codeps_maxpool3x3s2_fwd(ptr(x), b, h, w, c, ptr(y), ptr(arg), stream_ptr()), "codeps_maxpool3x3s2_fwd")
        ctx.save_for_backward(arg)
        ctx.shape = (b, h, w, c)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        b, h, w, c = ctx.shape
        dx = torch.empty(b, h, w, c, dtype=torch.float32, device=dy.device)
        check(lib().codeps_maxpool3x3s2_bwd(ptr(dy.contiguous()), ptr(arg), b, h, w, c, ptr(dx), stream_ptr()),
              "codeps_maxpool3x3s2_bwd")
        return dx


def max_pool3x3s2(x):
    return _MaxPool.apply(x)


_BNPOOL = os.environ.get("CODEPS_BNPOOL", "1") != "0"      # A/B switch: 0 = train-mode stem as separate BatchNorm and max-pool ops


class _BnReluPoolTrain(torch.autograd.Function):
    """maxpool3x3s2(relu(batch_norm_train(z))) of the stem (models/resnet_encoder.py:118-121) without the full-resolution activation
    or its gradient (C ABI: codeps_bn_fwd_mg(stats_only) + codeps_bn_relu_pool_{fwd,bwd}); `groups` as in _BatchNormTrain.  Bit-identical
    to the two separate ops.  `reduce_fn` (round 5): SyncBatchNorm -- statistics pass, all-gather + codeps_bn_sync_combine (equal or
    unequal shards, exactly as _BatchNormTrain), pooled pass; backward: this rank's sums, all-reduce, dz pass
    (misc/train_utils.py:50-51)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, momentum, eps, groups, reduce_fn=None):
        require_cuda(z, gamma, beta)
        z = z.contiguous()
        bsz, h, w, cp = z.shape
        c = gamma.numel()
        per_item = h * w
        plan = getattr(groups, "plan", None) if reduce_fn is not None else None
        groups = [(0, bsz)] if not groups else list(groups)
        ng, row0, rows = _group_arrays(groups, per_item)
        g, b_ = _pad_vec(gamma, cp), _pad_vec(beta, cp)
        dev = z.device
        mean = torch.empty(ng, cp, dtype=torch.float32, device=dev)
        var, invstd = torch.empty_like(mean), torch.empty_like(mean)
        ws = _workspace(lib().codeps_bn_mg_workspace_bytes(ng, cp), dev, "bn")
        rm = running_mean if (running_mean is not None and running_mean.numel() == c) else None
        rv = running_var if rm is not None else None
        counts = [float((b1 - b0) * per_item) for b0, b1 in groups]

        tiles = getattr(z, "_codeps_stats", None) if (_EPI and _EPI_PARTS["stats"]) else None     # partials from the stem convolution's epilogue

        def local_stats(mean_o, var_o, rm_, rv_):
            if tiles is not None:
                ngi, gb0, gb1 = _img_groups(groups)
                check(lib().codeps_bn_finalize_tiles(ptr(tiles["part"]), tiles["tile0"], tiles["tpi"], tiles["rows"], ngi, gb0, gb1, cp, eps,
                                                     ptr(mean_o), ptr(var_o), ptr(invstd), stream_ptr()), "codeps_bn_finalize_tiles")
                return                      # (the running statistics' momentum updates ride in the pooled pass below)
            check(lib().codeps_bn_fwd_mg(ptr(z), ng, row0, rows, cp, c, None, None, None, 1, eps, momentum, ptr(rm_), ptr(rv_), ptr(mean_o),
                                         ptr(var_o), ptr(invstd), None, 1, 0, ptr(ws), ws.numel(), stream_ptr()), "codeps_bn_fwd_mg (statistics)")
        if reduce_fn is None:
            local_stats(mean, var, rm, rv)
        elif plan is not None:          # unequal shards: one packet slot per GLOBAL statistic group (see _BatchNormTrain)
            G, l2g = plan["n_global"], plan["index"](dev)
            pk_local = torch.empty(2, ng, cp, dtype=torch.float32, device=dev)
            local_stats(pk_local[0], pk_local[1], None, None)
            pk = torch.zeros(2, G, cp, dtype=torch.float32, device=dev)
            pk.index_copy_(1, l2g, pk_local)
            gathered = reduce_fn.gather(pk)
            world = gathered.shape[0]
            rank_counts = [plan["items"][r][gg] * per_item for r in range(world) for gg in range(G)]
            gcounts = [sum(plan["items"][r][gg] for r in range(world)) * per_item for gg in range(G)]
            mean_g = torch.empty(G, cp, dtype=torch.float32, device=dev)
            var_g, invstd_g = torch.empty_like(mean_g), torch.empty_like(mean_g)
            check(lib().codeps_bn_sync_combine(ptr(gathered), world, G, (ctypes.c_float * G)(*gcounts), (ctypes.c_float * (world * G))(*rank_counts),
                                               cp, c, eps, momentum, ptr(rm), ptr(rv), ptr(mean_g), ptr(var_g), ptr(invstd_g), stream_ptr()),
                  "codeps_bn_sync_combine")
            mean, var, invstd = mean_g.index_select(0, l2g), var_g.index_select(0, l2g), invstd_g.index_select(0, l2g)
            counts = [gcounts[gg] for gg in plan["local_to_global"]]
        else:
            pk = torch.empty(2, ng, cp, dtype=torch.float32, device=dev)
            local_stats(pk[0], pk[1], None, None)
            gathered = reduce_fn.gather(pk)
            world = gathered.shape[0]
            counts = [n * world for n in counts]
            check(lib().codeps_bn_sync_combine(ptr(gathered), world, ng, (ctypes.c_float * ng)(*counts), None, cp, c, eps, momentum, ptr(rm), ptr(rv),
                                               ptr(mean), ptr(var), ptr(invstd), stream_ptr()), "codeps_bn_sync_combine")
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = torch.empty(bsz, ho, wo, cp, dtype=torch.float32, device=dev)
        arg = torch.empty(bsz, ho, wo, cp, dtype=torch.uint8, device=dev)
        run = tiles is not None and reduce_fn is None and rm is not None
        check(lib().codeps_bn_relu_pool_fwd(ptr(z), ng, row0, rows, bsz, h, w, cp, ptr(mean), ptr(invstd), ptr(g), ptr(b_), ptr(y), ptr(arg),
                                            ptr(rm) if run else None, ptr(rv) if run else None, ptr(var) if run else None, momentum, c,
                                            stream_ptr()), "codeps_bn_relu_pool_fwd")
        ctx.inv_count = [1.0 / n for n in counts]
        ctx.save_for_backward(z, arg, g, b_, mean, invstd)
        ctx.c, ctx.groups, ctx.geom, ctx.reduce_fn, ctx.plan = c, groups, (bsz, h, w, cp), reduce_fn, plan
        gs, bs = grad_sink(gamma), grad_sink(beta)
        ctx.sinks = (gs, bs) if (gs is not None and bs is not None and gamma.requires_grad and beta.requires_grad) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        z, arg, g, b_, mean, invstd = ctx.saved_tensors
        bsz, h, w, cp = ctx.geom
        ng, row0, rows = _group_arrays(ctx.groups, h * w)
        inv_count = (ctypes.c_float * ng)(*ctx.inv_count)
        dev = z.device
        sums = torch.empty(ng, 2, cp, dtype=torch.float32, device=dev)
        if ctx.sinks is not None:
            dgamma, dbeta, acc = ctx.sinks[0], ctx.sinks[1], 1
        else:
            dgamma = torch.empty(ctx.c, dtype=torch.float32, device=dev)
            dbeta, acc = torch.empty_like(dgamma), 0
        dz = torch.empty_like(z)
        ws = _workspace(lib().codeps_bn_mg_workspace_bytes(ng, cp), dev, "bn")
        dyc = dy.contiguous()

        def run(phase, s_):
            check(lib().codeps_bn_relu_pool_bwd(ptr(dyc), ptr(arg), ptr(z), ng, row0, rows, bsz, h, w, cp, ctx.c, ptr(mean), ptr(invstd),
                                                ptr(g), ptr(b_), inv_count, ptr(s_), ptr(dgamma), ptr(dbeta), acc, ptr(dz), ptr(ws), ws.numel(),
                                                phase, stream_ptr()), "codeps_bn_relu_pool_bwd")
        if ctx.reduce_fn is None:
            run(0, sums)
        else:
            run(1, sums)                                   # this rank's sums (+ its dgamma / dbeta: DDP adds the ranks' later)
            if ctx.plan is not None:
                l2g = ctx.plan["index"](dev)
                sums_g = torch.zeros(ctx.plan["n_global"], 2, cp, dtype=torch.float32, device=dev)
                sums_g.index_copy_(0, l2g, sums)
                sums = ctx.reduce_fn.sum(sums_g).index_select(0, l2g).contiguous()
            else:
                sums = ctx.reduce_fn.sum(sums)
            run(2, sums)
        if ctx.sinks is not None:
            dgamma = dbeta = None
        return dz, dgamma, dbeta, None, None, None, None, None, None


def bn_relu_maxpool_train(z, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5, groups=None, reduce_fn=None):
    """z may carry `._codeps_stats` (stem_conv2d(..., bn_stats=True)): the statistics pass over the stem's output is then a finish over
    per-tile partials."""
    return _BnReluPoolTrain.apply(z, gamma, beta, running_mean, running_var, momentum, eps, groups, reduce_fn)


class _Bilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out_h, out_w, align, out, c_off):
        require_cuda(x)
        x = x.contiguous()
        b, h, w, c = x.shape
        if out is None:
            out, c_off = torch.empty(b, out_h, out_w, c, dtype=torch.float32, device=x.device), 0
            ctx.mark = False
        else:
            ctx.mark_dirty(out)
            ctx.mark = True
        check(lib().codeps_bilinear_fwd(ptr(x), b, h, w, c, out_h, out_w, int(align), ptr(out), out.shape[-1], c_off,
                                        stream_ptr()), "codeps_bilinear_fwd")
        ctx.geom = (b, h, w, c, out_h, out_w, int(align), out.shape[-1], c_off)
        return out

    @staticmethod
    def backward(ctx, dy):
        b, h, w, c, oh, ow, align, yc, c_off = ctx.geom
        dx = torch.empty(b, h, w, c, dtype=torch.float32, device=dy.device)
        check(lib().codeps_bilinear_bwd(ptr(dy.contiguous()), b, h, w, c, oh, ow, align, yc, c_off, ptr(dx), stream_ptr()),
              "codeps_bilinear_bwd")
        return dx, None, None, None, (dy if ctx.mark else None), None


def bilinear_resize(x, out_h, out_w, align_corners):
    return _Bilinear.apply(x, int(out_h), int(out_w), bool(align_corners), None, 0)


class _BilinearConcat(torch.autograd.Function):
    """cat([resize(x_i)], channel) materialised once: every source is resized straight into its channel
    slice of the destination (GUDA fusion, models/semantic_head.py:64-70; instance decoder cat)."""

    @staticmethod
    def forward(ctx, out_h, out_w, align, *xs):
        require_cuda(*xs)
        xs = [x.contiguous() for x in xs]
        b = xs[0].shape[0]
        ctot = sum(x.shape[-1] for x in xs)
        out = torch.empty(b, out_h, out_w, ctot, dtype=torch.float32, device=xs[0].device)
        off, geoms = 0, []
        for x in xs:
            _, h, w, c = x.shape
            # (same size: the interpolation is the identity under both align_corners conventions -- the kernel copies)
            check(lib().codeps_bilinear_fwd(ptr(x), b, h, w, c, out_h, out_w, int(align), ptr(out), ctot, off,
                                            stream_ptr()), "codeps_bilinear_fwd")
            geoms.append((h, w, c, off))
            off += c
        ctx.geoms, ctx.meta = geoms, (b, out_h, out_w, int(align), ctot)
        return out

    @staticmethod
    def backward(ctx, dy):
        b, oh, ow, align, ctot = ctx.meta
        dy = dy.contiguous()
        grads = []
        for i, (h, w, c, off) in enumerate(ctx.geoms):
            if not ctx.needs_input_grad[3 + i]:
                grads.append(None)
                continue
            if (h, w) == (oh, ow):
                grads.append(dy[..., off:off + c].contiguous())
                continue
            dx = torch.empty(b, h, w, c, dtype=torch.float32, device=dy.device)
            check(lib().codeps_bilinear_bwd(ptr(dy), b, h, w, c, oh, ow, align, ctot, off, ptr(dx), stream_ptr()),
                  "codeps_bilinear_bwd")
            grads.append(dx)
        return (None, None, None, *grads)


def bilinear_concat(xs, out_h, out_w, align_corners):
    return _BilinearConcat.apply(int(out_h), int(out_w), bool(align_corners), *xs)


class _GudaGather(torch.autograd.Function):
    """out = base + sum_k sum_tap up(y_k[tap])(p + delta_tap)   (C ABI: codeps_guda_gather_{fwd,bwd})."""

    @staticmethod
    def forward(ctx, base, *ys):
        require_cuda(base, *ys)
        base = base.contiguous()
        ys = [y.contiguous() for y in ys]
        b, h, w, c = base.shape
        if c != 16 or any(y.shape[-1] != 144 for y in ys) or not 1 <= len(ys) <= 3:
            raise CodepsHipError("guda_gather: base must carry 16 channels and every branch 9 x 16 tap channels")
        n = len(ys)
        out = torch.empty_like(base)
        yp = (ctypes.c_void_p * n)(*[y.data_ptr() for y in ys])
        bh = (ctypes.c_int * n)(*[y.shape[1] for y in ys])
        bw = (ctypes.c_int * n)(*[y.shape[2] for y in ys])
        check(lib().codeps_guda_gather_fwd(ptr(base), b, h, w, n, yp, bh, bw, ptr(out), stream_ptr()), "codeps_guda_gather_fwd")
        ctx.geoms = [(y.shape[1], y.shape[2]) for y in ys]
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        b, h, w, _ = dout.shape
        grads = []
        for i, (bh, bw) in enumerate(ctx.geoms):
            if not ctx.needs_input_grad[1 + i]:
                grads.append(None)
                continue
            dy = torch.empty(b, bh, bw, 144, dtype=torch.float32, device=dout.device)
            ws = _workspace(lib().codeps_guda_gather_bwd_workspace_bytes(b, h, bw), dout.device, "guda")
            check(lib().codeps_guda_gather_bwd(ptr(dout), b, h, w, bh, bw, ptr(dy), ptr(ws), ws.numel(), stream_ptr()),
                  "codeps_guda_gather_bwd")
            grads.append(dy)
        return (dout if ctx.needs_input_grad[0] else None, *grads)


class _GudaWeights(torch.autograd.Function):
    """semconv_guda weight [n, Ct, 3, 3] -> (taps_k [144, c_k] per coarse branch ..., base [n, c_f, 3, 3]) and back
    (C ABI: codeps_guda_weight_{fwd,bwd})."""

    @staticmethod
    def forward(ctx, weight, chans):
        w = _f32c(weight.detach())
        n, ct = w.shape[:2]
        nc = len(chans) - 1
        sizes = [144 * c for c in chans[:-1]] + [n * chans[-1] * 9]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=w.device)
        outs, off = [], 0
        for i, sz in enumerate(sizes):
            outs.append(flat[off:off + sz].view(144, chans[i]) if i < nc else flat[off:off + sz].view(n, chans[-1], 3, 3))
            off += sz
        ca = (ctypes.c_int * (nc + 1))(*chans)
        pa = (ctypes.c_void_p * (nc + 1))(*[o.data_ptr() for o in outs])
        check(lib().codeps_guda_weight_fwd(ptr(w), n, ct, nc, ca, pa, stream_ptr()), "codeps_guda_weight_fwd")
        ctx.meta = (n, ct, nc, tuple(chans))
        ctx.sink = grad_sink(weight) if weight.requires_grad else None
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        n, ct, nc, chans = ctx.meta
        gs = [None if g is None else g.contiguous() for g in gs]
        dev = next(g for g in gs if g is not None).device
        sink = ctx.sink
        gw = sink if sink is not None else torch.empty(n, ct, 3, 3, dtype=torch.float32, device=dev)
        ca = (ctypes.c_int * (nc + 1))(*chans)
        pa = (ctypes.c_void_p * (nc + 1))(*[None if g is None else g.data_ptr() for g in gs])
        check(lib().codeps_guda_weight_bwd(pa, n, ct, nc, ca, 1 if sink is not None else 0, ptr(gw), stream_ptr()), "codeps_guda_weight_bwd")
        return (None if sink is not None else gw), None


def guda_fusion(coarse, fine, weight, bias):
    """conv3x3(cat([bilinear_up(c) for c in coarse] + [fine], channel), weight, bias, padding=1) without the
    concatenated tensor (GUDA fusion, models/semantic_head.py:64-72).  Convolution and up-sampling are both linear and the
    up-sampling is per channel, so the channel contraction of every 3x3 tap is done at the coarse resolution
    (a 1x1 conv to 9 x 16 tap channels on the conv engine: 1/4 ... 1/64 of the multiply-adds) and only the tap planes
    are interpolated at full resolution."""
    n = weight.shape[0]
    if n > 16 or weight.shape[-2:] != (3, 3):
        raise CodepsHipError("guda_fusion: expects a 3x3 convolution to <= 16 classes")
    chans = [t.shape[-1] for t in coarse] + [fine.shape[-1]]
    if sum(chans) != weight.shape[1]:
        raise CodepsHipError("guda_fusion: channel counts of the branches do not add up to the weight's input channels")
    parts = _GudaWeights.apply(weight, tuple(chans))            # ONE launch: per-branch tap matrices + the fine branch's block
    ys = [conv2d(t, taps.view(144, ck, 1, 1)) for t, taps, ck in zip(coarse, parts[:-1], chans[:-1])]
    base = conv2d(fine, parts[-1], bias, pad=1)
    return _GudaGather.apply(base, *ys)


# ---------------------------------------------------------------------------------------- depthwise 5x5
_dw_cache = {}


def _dw_pack(weight, c):
    """Depthwise weight [C,1,5,5] -> [25][Cp] tap-major pack; cached for frozen weights (identity + version checked)."""
    frozen = not _is_volatile(weight)
    key = (id(weight), c)
    if frozen:
        hit = _dw_cache.get(key)
        if hit is not None and hit[0]() is weight and hit[1] == (weight.data_ptr(), weight._version):
            return hit[2]
    creal = weight.shape[0]
    w25 = torch.zeros(25, c, dtype=torch.float32, device=weight.device)
    w25[:, :creal] = weight.detach().reshape(creal, 25).t()
    if frozen:
        _dw_cache[key] = (_weak(weight, _dw_cache, key), (weight.data_ptr(), weight._version), w25)
    return w25


class _DwConv5(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, scale, shift, act):
        require_cuda(x, weight)
        x = x.contiguous()
        b, h, w, c = x.shape
        creal = weight.shape[0]
        w25 = _dw_pack(weight, c)
        sc, sh = _pad_vec(scale, c), _pad_vec(shift, c)
        y = torch.empty_like(x)
        check(lib().codeps_dwconv5x5(ptr(x), ptr(w25), ptr(sc), ptr(sh), act, b, h, w, c, 0, ptr(y), stream_ptr()),
              "codeps_dwconv5x5")
        ctx.save_for_backward(x, w25, y if act else None, sc)
        ctx.act, ctx.creal = act, creal
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w25, y, sc = ctx.saved_tensors
        b, h, w, c = x.shape
        dy = dy.contiguous()
        rows = b * h * w
        if ctx.act or sc is not None:
            dz = torch.empty_like(dy)
            check(lib().codeps_act_bwd(ptr(dy), ptr(y), ptr(sc), ctx.act, rows, c, ptr(dz), None, c, 0, None, 0, stream_ptr()),
                  "codeps_act_bwd")
        else:
            dz = dy
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            check(lib().codeps_dwconv5x5(ptr(dz), ptr(w25), None, None, 0, b, h, w, c, 1, ptr(dx), stream_ptr()),
                  "codeps_dwconv5x5")
        if ctx.needs_input_grad[1]:
            d25 = torch.empty(25, c, dtype=torch.float32, device=x.device)
            wsb = lib().codeps_colsum_workspace_bytes(rows, c, 25)
            ws = _workspace(wsb, x.device, "colsum")
            check(lib().codeps_dwconv5x5_wgrad(ptr(x), ptr(dz), b, h, w, c, ptr(d25), ptr(ws), ws.numel(), stream_ptr()),
                  "codeps_dwconv5x5_wgrad")
            dw = d25[:, :ctx.creal].t().reshape(ctx.creal, 1, 5, 5).contiguous()
        return dx, dw, None, None, None


def dwconv5x5(x, weight, scale=None, shift=None, act=None):
    return _DwConv5.apply(x, weight, scale, shift, ACT[act])


# ---------------------------------------------------------------------------------------- output heads
class _Softmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c):
        require_cuda(x)
        x = x.contiguous()
        cp = x.shape[-1]
        y = torch.empty_like(x)
        check(lib().codeps_softmax_fwd(ptr(x), x.numel() // cp, c, cp, ptr(y), stream_ptr()), "codeps_softmax_fwd")
        ctx.save_for_backward(y)
        ctx.c = c
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        cp = y.shape[-1]
        dx = torch.empty_like(y)
        check(lib().codeps_softmax_bwd(ptr(dy.contiguous()), ptr(y), y.numel() // cp, ctx.c, cp, ptr(dx), stream_ptr()),
              "codeps_softmax_bwd")
        return dx, None


def channel_argmax(probs, dtype=torch.int64):
    """torch.argmax(probs, dim=1) for an NCHW tensor that is (a view of) an NHWC tensor of this library -> labels [B,H,W] in
    int64 (torch's dtype) or uint8.  An int64 result carries its uint8 form as `._codeps_u8` (what panoptic_fusion reads): one
    pass, no conversion kernels (algos/semantic_seg.py:70-86)."""
    require_cuda(probs)
    b, c, h, w = probs.shape
    probs = probs.detach()
    st = probs.stride()
    # in place only for a view that starts at channel 0 of a 16-byte aligned pixel (the kernel issues f32x4 loads and the widened view
    # must stay inside the storage): anything else goes through a contiguous copy
    if (probs.dtype == torch.float32 and st[1] == 1 and st[3] >= c and st[3] % 4 == 0 and st[2] == w * st[3] and st[0] == h * w * st[3]
            and probs.storage_offset() % st[3] == 0 and probs.data_ptr() % 16 == 0):
        x = probs.as_strided((b, h, w, st[3]), (st[0], st[2], st[3], 1))      # a channel view of a padded NHWC tensor: read in place
    else:
        x = from_nchw(probs).contiguous()
    cp = x.shape[-1]
    o8 = torch.empty(b, h, w, dtype=torch.uint8, device=probs.device)
    o64 = torch.empty(b, h, w, dtype=torch.int64, device=probs.device) if dtype == torch.int64 else None
    check(lib().codeps_channel_argmax(ptr(x), b * h * w, c, cp, ptr(o64), ptr(o8), stream_ptr()), "codeps_channel_argmax")
    if o64 is None:
        return o8
    o64._codeps_u8 = (o8, o64._version)        # valid while the int64 map is unmodified (panoptic_fusion checks the version)
    return o64


def channel_softmax(x, c):
    """Softmax over the first c channels of NHWC logits (padding channels -> 0)."""
    return _Softmax.apply(x, int(c))


class _DispDepth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, min_depth, max_depth):
        require_cuda(logits)
        logits = logits.contiguous()
        b, h, w, cp = logits.shape
        disp = torch.empty(b, 1, h, w, dtype=torch.float32, device=logits.device)
        depth = torch.empty_like(disp)
        check(lib().codeps_disp_depth_fwd(ptr(logits), b * h * w, cp, min_depth, max_depth, ptr(disp), ptr(depth), stream_ptr()),
              "codeps_disp_depth_fwd")
        ctx.save_for_backward(disp)
        ctx.meta = (cp, min_depth, max_depth)
        return depth, disp

    @staticmethod
    def backward(ctx, ddepth, ddisp):
        (disp,) = ctx.saved_tensors
        cp, mn, mx = ctx.meta
        b, _, h, w = disp.shape
        dx = torch.empty(b, h, w, cp, dtype=torch.float32, device=disp.device)
        ddepth = None if ddepth is None else ddepth.contiguous()
        ddisp = None if ddisp is None else ddisp.contiguous()
        check(lib().codeps_disp_depth_bwd(ptr(ddisp), ptr(ddepth), ptr(disp), b * h * w, cp, mn, mx, ptr(dx), stream_ptr()),
              "codeps_disp_depth_bwd")
        return dx, None, None


def disp_depth(logits, min_depth=0.1, max_depth=100.0):
    """sigmoid + disp_to_depth on channel 0 of NHWC logits -> (depth, disp), both [B,1,H,W]."""
    return _DispDepth.apply(logits, float(min_depth), float(max_depth))


class _UpConv16(torch.autograd.Function):
    """act(conv3x3(nearest_x2(x)) + bias), 16 -> 16 channels, as four 2x2 convolutions of the half-resolution tensor
    (C ABI: codeps_upconv16_*)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, opts=None):
        require_cuda(x, weight)
        x = x.contiguous()
        b, hs, ws_, c = x.shape
        wt = _f32c(weight.detach())
        bs = None if bias is None else _f32c(bias.detach())
        y = torch.empty(b, 2 * hs, 2 * ws_, 16, dtype=torch.float32, device=x.device)
        flops = 2.0 * b * 4 * hs * ws_ * 16 * 16 * 4                    # performed: four 2x2 convolutions (the reference's 3x3 form: x 9 / 4)
        t0 = _prof_begin()
        nb = lib().codeps_upconv16_workspace_bytes()
        wsp = _workspace(nb, x.device, "upconv16_w")
        check(lib().codeps_upconv16_fwd(ptr(x), ptr(wt), ptr(bs), b, hs, ws_, act, ptr(y), ptr(wsp), nb, stream_ptr()), "codeps_upconv16_fwd")
        _prof_end(t0, "conv_fwd", flops)
        ctx.act, ctx.flops, ctx.has_bias = act, flops, bias is not None
        ctx.w_sink = grad_sink(weight) if weight.requires_grad else None
        ctx.b_sink = grad_sink(bias) if (bias is not None and bias.requires_grad) else None
        ctx.save_for_backward(x, wt, y if act else None)
        # gates (see _Gate): the one x came with (its producer's activation backward runs in this layer's data-gradient kernel) and the
        # one this layer's own activation offers to its exclusive reader (disp_head)
        ctx.in_gate = opts.in_gate if (opts is not None and ctx.needs_input_grad[0]) else None
        ctx.out_gate = None
        if opts is not None and _EPI and _EPI_PARTS["act"] and act and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            g = _Gate("act", act)
            g.want_colsum = bias is not None and ctx.needs_input_grad[2]
            ctx.out_gate = opts.gate = g
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wt, y = ctx.saved_tensors
        gated = ctx.out_gate.take(dy) if ctx.out_gate is not None else None
        dy = dy.contiguous()
        b, hs, ws_, c = x.shape
        rows = dy.numel() // 16
        dev = dy.device
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        dbias = None
        if need_b:
            dbias = ctx.b_sink if ctx.b_sink is not None else torch.empty(16, dtype=torch.float32, device=dev)
        dz = dy
        if gated is not None:                   # dy arrived multiplied by act'(y), with its column sums per block
            if need_b:
                if gated["part"] is not None:
                    _colsum_finish(gated["part"], gated["segs"][0][3], 16, dbias, 16, 1 if ctx.b_sink is not None else 0)
                else:
                    ws = _workspace(lib().codeps_colsum_workspace_bytes(rows, 16, 1), dev, "colsum")
                    check(lib().codeps_act_bwd(ptr(dy), None, None, 0, rows, 16, None, ptr(dbias), 16, 1 if ctx.b_sink is not None else 0, ptr(ws),
                                               ws.numel(), stream_ptr()), "codeps_act_bwd (bias gradient)")
                if ctx.b_sink is not None:
                    dbias = None
        elif ctx.act or need_b:
            wsb = lib().codeps_colsum_workspace_bytes(rows, 16, 1)
            ws = _workspace(wsb, dev, "colsum")
            dz = torch.empty_like(dy) if ctx.act else dy
            check(lib().codeps_act_bwd(ptr(dy), ptr(y), None, ctx.act, rows, 16, ptr(dz) if dz is not dy else None, ptr(dbias), 16,
                                       1 if ctx.b_sink is not None else 0, ptr(ws), ws.numel(), stream_ptr()), "codeps_act_bwd")
            if ctx.b_sink is not None:
                dbias = None
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            t0 = _prof_begin()
            nb = lib().codeps_upconv16_workspace_bytes()
            wsp = _workspace(nb, dev, "upconv16_w")
            gt = ctx.in_gate
            if gt is not None:
                blocks = lib().codeps_upconv16_dgrad_blocks(b, hs, ws_)
                part = torch.empty(blocks, 16, dtype=torch.float32, device=dev) if gt.want_colsum else None
                check(lib().codeps_upconv16_dgrad_gated(ptr(dz), ptr(wt), b, hs, ws_, ptr(dx), ptr(x), gt.act, ptr(part), ptr(wsp), nb, stream_ptr()),
                      "codeps_upconv16_dgrad_gated")
                _handoff_blocks(gt, dx, part, blocks)
            else:
                check(lib().codeps_upconv16_dgrad(ptr(dz), ptr(wt), b, hs, ws_, ptr(dx), ptr(wsp), nb, stream_ptr()), "codeps_upconv16_dgrad")
            _prof_end(t0, "conv_dgrad", ctx.flops)
        if ctx.needs_input_grad[1]:
            sink = ctx.w_sink
            dw = sink if sink is not None else torch.empty(16, 16, 3, 3, dtype=torch.float32, device=dev)
            nbytes = lib().codeps_upconv16_wgrad_workspace_bytes()
            ws = _workspace(nbytes, dev, "upconv16")
            t0 = _prof_begin()
            check(lib().codeps_upconv16_wgrad(ptr(x), ptr(dz), b, hs, ws_, ptr(dw), 1 if sink is not None else 0, ptr(ws), nbytes, stream_ptr()),
                  "codeps_upconv16_wgrad")
            _prof_end(t0, "conv_wgrad", ctx.flops)
            if sink is not None:
                dw = None
        return dx, dw, dbias, None, None


def upconv16(x, weight, bias=None, act=None, exclusive=False):
    """act(conv3x3(nearest_x2(x), weight) + bias) for 16 -> 16 channels without a skip tensor (the decoders' last stage,
    models/depth_head.py:60-71): x [B,hs,ws,16] NHWC -> [B,2hs,2ws,16].  Same values as conv2d(..., up0=True) to fp32 rounding
    (the taps that read one source pixel are summed in the weights first).  exclusive: as in conv2d -- x's producer's activation backward
    then runs in this layer's data-gradient kernel; the result offers its own gate to ITS exclusive reader."""
    opts = _ConvOpts(_act_gate_of(x, exclusive))
    y = _UpConv16.apply(x, weight, bias, ACT[act], opts)
    if opts.gate is not None:
        y._codeps_gate = opts.gate
    return y


class _DispHead(torch.autograd.Function):
    """ReflectionPad2d(1) -> Conv2d(16, 1, 3) -> Sigmoid -> disp_to_depth in one kernel each way (C ABI: codeps_disp_head_*)."""

    @staticmethod
    def forward(ctx, x, weight, bias, min_depth, max_depth, logit_out, in_gate=None):
        require_cuda(x, weight)
        x = x.contiguous()
        b, h, w, c = x.shape
        wt = _f32c(weight.detach())
        bs = None if bias is None else _f32c(bias.detach())
        disp = torch.empty(b, 1, h, w, dtype=torch.float32, device=x.device)
        depth = torch.empty_like(disp)
        if logit_out is not None and (tuple(logit_out.shape) != (b, 1, h, w) or logit_out.dtype != torch.float32 or not logit_out.is_contiguous()):
            raise CodepsHipError("disp_head: logit_out must be a contiguous float32 [B,1,H,W] tensor")
        check(lib().codeps_disp_head_fwd(ptr(x), ptr(wt), ptr(bs), b, h, w, c, min_depth, max_depth, ptr(disp), ptr(depth), ptr(logit_out),
                                         stream_ptr()), "codeps_disp_head_fwd")
        ctx.save_for_backward(x, wt, disp)
        ctx.in_gate = in_gate if ctx.needs_input_grad[0] else None
        ctx.meta = (min_depth, max_depth, bias is not None)
        ctx.w_sink = grad_sink(weight) if weight.requires_grad else None
        ctx.b_sink = grad_sink(bias) if (bias is not None and bias.requires_grad) else None
        return depth, disp

    @staticmethod
    def backward(ctx, ddepth, ddisp):
        x, wt, disp = ctx.saved_tensors
        mn, mx, has_bias = ctx.meta
        b, h, w, c = x.shape
        dev = x.device
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        need_w, need_b = ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2]
        # weight and bias gradients leave through one launch: both accumulate into the flat buffer, or neither
        sinks = ctx.w_sink is not None and (ctx.b_sink is not None or not need_b)
        dw = db = None
        if need_w:
            dw = ctx.w_sink if sinks else torch.empty(1, c, 3, 3, dtype=torch.float32, device=dev)
        if need_b:
            db = ctx.b_sink if sinks else torch.empty(1, dtype=torch.float32, device=dev)
        nbytes = lib().codeps_disp_head_bwd_workspace_bytes(b, h, w)
        ws = _workspace(nbytes, dev, "disp_head")
        gt = ctx.in_gate if dx is not None else None
        blocks = lib().codeps_disp_head_dx_blocks(b, h, w) if gt is not None else 0
        part = torch.empty(blocks, 16, dtype=torch.float32, device=dev) if (gt is not None and gt.want_colsum) else None
        check(lib().codeps_disp_head_bwd_gated(ptr(x), ptr(wt), ptr(disp), ptr(None if ddisp is None else ddisp.contiguous()),
                                               ptr(None if ddepth is None else ddepth.contiguous()), b, h, w, c, mn, mx, ptr(dx), ptr(dw), ptr(db),
                                               1 if sinks else 0, gt.act if gt is not None else 0, ptr(part), ptr(ws), nbytes, stream_ptr()),
              "codeps_disp_head_bwd")
        if gt is not None:
            _handoff_blocks(gt, dx, part, blocks)
        if sinks:
            dw = db = None              # accumulated in place into the flat gradient buffer
        elif ctx.w_sink is not None and dw is not None:
            ctx.w_sink.add_(dw.view_as(ctx.w_sink))
            dw = None
        return dx, dw, db, None, None, None, None


def disp_head(x, weight, bias, min_depth=0.1, max_depth=100.0, logit_out=None, exclusive=False):
    """The depth decoder's output stage (models/depth_head.py:44-59,73-77) on NHWC features x [B,H,W,16]: reflection-padded
    3x3 convolution to one channel, sigmoid, disp_to_depth -> (depth, disp), both [B,1,H,W]; the 16-channel tensor is moved once
    each way.  logit_out (optional, [B,1,H,W]): receives the pre-sigmoid convolution output (parity checks in logit space)."""
    return _DispHead.apply(x, weight, bias, float(min_depth), float(max_depth), logit_out, _act_gate_of(x, exclusive))


# ---------------------------------------------------------------------------------------- pose
class _Pose(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, invert):
        require_cuda(feat)
        feat = feat.contiguous()
        b, h, w, cp = feat.shape
        vec = torch.empty(b, 6, dtype=torch.float32, device=feat.device)
        T = torch.empty(b, 4, 4, dtype=torch.float32, device=feat.device)
        check(lib().codeps_pose_fwd(ptr(feat), b, h * w, cp, invert, ptr(vec), ptr(T), stream_ptr()), "codeps_pose_fwd")
        ctx.save_for_backward(vec)
        ctx.meta = (b, h, w, cp, invert)
        return T

    @staticmethod
    def backward(ctx, dT):
        (vec,) = ctx.saved_tensors
        b, h, w, cp, inv = ctx.meta
        df = torch.empty(b, h, w, cp, dtype=torch.float32, device=vec.device)
        check(lib().codeps_pose_bwd(ptr(dT.contiguous()), ptr(vec), b, h * w, cp, inv, ptr(df), stream_ptr()), "codeps_pose_bwd")
        return df, None


def pose_matrix(feat, invert=False):
    """NHWC pose features [B,h,w,Cp] -> [B,4,4] (spatial mean * 0.01, frame 0, Rodrigues; models/pose_head.py:46-54).
    invert: bool for the whole batch, or a per-item sequence of bools (B <= 64): several reference-style calls in one launch."""
    if isinstance(invert, (list, tuple)):
        if len(invert) != feat.shape[0] or len(invert) > 64:
            raise CodepsHipError("pose_matrix: one invert flag per item, at most 64 items")
        mask = sum(1 << i for i, v in enumerate(invert) if v)
    else:
        mask = 0xFFFFFFFFFFFFFFFF if invert else 0
    return _Pose.apply(feat, mask)


class _SplitBatch(torch.autograd.Function):
    """x -> (x[b0:b1] for every span) as views; the backward writes the spans' gradients into ONE buffer (a single cat launch)
    instead of autograd's zeros + slice copy per span and an add per extra span."""

    @staticmethod
    def forward(ctx, x, bounds):
        ctx.bounds, ctx.shape = bounds, tuple(x.shape)
        ctx.set_materialize_grads(False)
        return tuple(x[b0:b1] for b0, b1 in bounds)

    @staticmethod
    def backward(ctx, *gs):
        if all(g is None for g in gs):
            return None, None
        like = next(g for g in gs if g is not None)
        covered, parts, pos = True, [], 0
        for (b0, b1), g in zip(ctx.bounds, gs):
            covered = covered and b0 == pos
            pos = b1
            parts.append(g if g is not None else torch.zeros((b1 - b0,) + ctx.shape[1:], dtype=like.dtype, device=like.device))
        if covered and pos == ctx.shape[0]:
            if like.dtype == torch.float32 and len(parts) <= 8:
                return concat_rows(parts), None
            return torch.cat(parts, 0), None
        out = torch.zeros(ctx.shape, dtype=like.dtype, device=like.device)       # overlapping / partial spans: accumulate
        for (b0, b1), g in zip(ctx.bounds, gs):
            if g is not None:
                out[b0:b1] += g
        return out, None


def split_batch(x, bounds):
    """Views x[b0:b1] for bounds = [(b0, b1), ...] along dim 0 with a single-launch backward when the spans tile the batch."""
    if not x.requires_grad or not torch.is_grad_enabled():
        return tuple(x[b0:b1] for b0, b1 in bounds)
    return _SplitBatch.apply(x, tuple((int(a), int(b)) for a, b in bounds))


# ---------------------------------------------------------------------------------------- losses
def _loss_ws(pixels, device):
    return torch.empty(lib().codeps_loss_workspace_bytes(pixels), dtype=torch.uint8, device=device)


class _Smoothness(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, disp):
        require_cuda(img, disp)
        img, disp = _f32c(img), _f32c(disp)
        b, _, h, w = img.shape
        ws = _loss_ws(b * h * w, img.device)
        mean = torch.empty(b, dtype=torch.float32, device=img.device)
        loss = torch.empty(1, dtype=torch.float32, device=img.device)
        check(lib().codeps_smoothness_fwd(ptr(img), ptr(disp), b, h, w, ptr(mean), ptr(loss), ptr(ws), ws.numel(), stream_ptr()),
              "codeps_smoothness_fwd")
        ctx.save_for_backward(img, disp, mean, ws)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        img, disp, mean, ws = ctx.saved_tensors
        b, _, h, w = img.shape
        dd = torch.empty_like(disp)
        check(lib().codeps_smoothness_bwd(ptr(img), ptr(disp), ptr(mean), ptr(_f32c(g).reshape(1)), b, h, w, ptr(dd), ptr(ws),
                                          ws.numel(), stream_ptr()), "codeps_smoothness_bwd")
        return None, dd


def smoothness_loss(img, disp):
    """Edge-aware smoothness (algos/depth.py:58-107); img [B,3,H,W], disp [B,1,H,W] NCHW."""
    return _Smoothness.apply(img, disp)


class _SemanticLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, probs, c, target, pixel_weights, class_weights, top_k, ignore_index, drop=0):
        require_cuda(probs, target, pixel_weights, class_weights)
        probs = probs.contiguous()
        cp = probs.shape[-1]
        n = probs.numel() // cp
        target = target.contiguous()
        if target.dtype != torch.uint8:
            target = target.to(torch.uint8)
        if pixel_weights is not None:
            pixel_weights = pixel_weights.contiguous().to(torch.uint8)
        cw = None if class_weights is None else _f32c(class_weights)
        ws = _loss_ws(n, probs.device)
        loss = torch.empty(1, dtype=torch.float32, device=probs.device)
        check(lib().codeps_semantic_loss_fwd(ptr(probs), c, cp, ptr(target), ptr(pixel_weights), ptr(cw), ignore_index, drop, n, top_k,
                                             ptr(loss), ptr(ws), ws.numel(), stream_ptr()), "codeps_semantic_loss_fwd")
        ctx.save_for_backward(probs, target, pixel_weights, cw, ws)
        ctx.meta = (c, cp, n, top_k, ignore_index, drop)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        probs, target, pw, cw, ws = ctx.saved_tensors
        c, cp, n, top_k, ign, drop = ctx.meta
        dp = torch.empty_like(probs)
        check(lib().codeps_semantic_loss_bwd(ptr(probs), c, cp, ptr(target), ptr(pw), ptr(cw), ign, drop, n, top_k,
                                             ptr(_f32c(g).reshape(1)), ptr(dp), ptr(ws), ws.numel(), stream_ptr()),
              "codeps_semantic_loss_bwd")
        return dp, None, None, None, None, None, None, None


def semantic_loss(probs_nhwc, num_classes, target, pixel_weights=None, class_weights=None, top_k=0.2, ignore_index=255,
                  ignore_labels=None):
    """Hard-pixel-mined CE on softmax probabilities (algos/semantic_seg.py:45-67); probs NHWC [B,H,W,Cp].
    ignore_labels: labels whose pixels are ignored and whose channels leave the CE (semantic_seg.py:50-55; the preserved
    channels are indexed with the original label values like the reference does; class_weights: one per preserved channel)."""
    drop = 0
    for label in (ignore_labels or ()):
        if not 0 <= int(label) < 64:
            raise CodepsHipError(f"semantic_loss: ignore label {label} outside [0, 64)")
        drop |= 1 << int(label)
    if class_weights is not None and class_weights.numel() != int(num_classes) - bin(drop & ((1 << int(num_classes)) - 1)).count("1"):
        raise CodepsHipError("semantic_loss: weight tensor should be defined either for all preserved classes or no classes")
    return _SemanticLoss.apply(probs_nhwc, int(num_classes), target, pixel_weights, class_weights, float(top_k), int(ignore_index), drop)


class _InstanceLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, c, target, kind, ignore_index):
        require_cuda(pred, target)
        pred = pred.contiguous()
        b, h, w, cp = pred.shape
        if kind == 2:
            tf, tu = None, target.contiguous().to(torch.uint8)
        else:
            tf, tu = _f32c(target), None
        ws = _workspace(4096, pred.device, "instloss")
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        check(lib().codeps_instance_loss_fwd(kind, ptr(pred), c, cp, ptr(tf), ptr(tu), ignore_index, b, h * w, ptr(loss), ptr(ws),
                                             ws.numel(), stream_ptr()), "codeps_instance_loss_fwd")
        ctx.save_for_backward(pred, tf, tu)
        ctx.meta = (c, cp, kind, ignore_index, b, h * w)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        pred, tf, tu = ctx.saved_tensors
        c, cp, kind, ign, b, hw = ctx.meta
        dp = torch.empty_like(pred)
        check(lib().codeps_instance_loss_bwd(kind, ptr(pred), c, cp, ptr(tf), ptr(tu), ign, b, hw, ptr(_f32c(g).reshape(1)), ptr(dp),
                                             stream_ptr()), "codeps_instance_loss_bwd")
        return dp, None, None, None, None


def center_loss(pred_nhwc, target):
    """mean((c - c_gt)^2) (algos/instance_seg.py:12-25); pred NHWC [B,H,W,Cp] (ch 0), target [B,1,H,W]."""
    return _InstanceLoss.apply(pred_nhwc, 1, target, 0, 255)


def offset_loss(pred_nhwc, target):
    """mean(|o - o_gt|) (algos/instance_seg.py:28-41); pred NHWC (ch 0,1), target [B,2,H,W]."""
    return _InstanceLoss.apply(pred_nhwc, 2, target, 1, 255)


def binary_mask_loss(probs_nhwc, target, ignore_index=255):
    """mean CE on thing probabilities (algos/instance_seg.py:44-51); target u8 [B,H,W]."""
    return _InstanceLoss.apply(probs_nhwc, 2, target, 2, int(ignore_index))


class _Ssim(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, c1, c2):
        require_cuda(x, y)
        x, y = _f32c(x), _f32c(y)
        if x.shape != y.shape or x.dim() != 4:
            raise CodepsHipError(f"ssim: expects two [B,C,H,W] tensors of one shape, got {tuple(x.shape)} and {tuple(y.shape)}")
        b, c, h, w = x.shape
        out = torch.empty_like(x)
        check(lib().codeps_ssim_fwd(ptr(x), ptr(y), b * c, h, w, c1, c2, ptr(out), stream_ptr()), "codeps_ssim_fwd")
        ctx.save_for_backward(x, y)
        ctx.c = (c1, c2)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        b, c, h, w = x.shape
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        nbytes = lib().codeps_ssim_bwd_workspace_bytes(b * c, h, w)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        check(lib().codeps_ssim_bwd(ptr(x), ptr(y), ptr(_f32c(g)), b * c, h, w, ctx.c[0], ctx.c[1], ptr(gx), ptr(gy), ptr(ws), nbytes,
                                    stream_ptr()), "codeps_ssim_bwd")
        return gx, gy, None, None


def ssim_loss_map(x, y, c1=1e-4, c2=9e-4):
    """SSIMLoss.__call__ (algos/depth.py:128-155): per-pixel, per-channel clamp((1 - SSIM) / 2, 0, 1) of two [B,C,H,W] images."""
    return _Ssim.apply(x, y, float(c1), float(c2))


# ---------------------------------------------------------------------------------------- step control
def depth_rule_state(device, length=300):
    """Zero-initialised device state of the rolling "car not moving" rule (algos/depth.py:537-558)."""
    return torch.zeros(lib().codeps_depth_rule_state_floats(int(length)), dtype=torch.float32, device=device)


def depth_key_weights(counts, device, target_index=-1, poses=None, rule_state=None, length=300, shard=None):
    """Image-count weights of DepthAlgo.adaptation's per-key losses (algos/depth.py:562-577) as a device vector
    [n_keys + 1] = (n_k * keep_k ..., the denominator), with the "car not moving" rule (depth.py:537-558) evaluated on the device
    for key `target_index` from that key's (T_t->t-1, T_t->t+1): no host synchronisation, so the rule stays live inside a
    replayed HIP graph.  `rule_state` (depth_rule_state) carries the rolling window across frames; it is updated in place.

    shard (batch-sharded step, world > 1): dict(world, global_total, global_target, rule: bool, owner: bool, group).  `counts`
    are this rank's; the denominator is the global image count / world.  With `rule`, the rank that owns the global batch's
    first target item (`owner`, rank 0 by construction) evaluates the rule and the keep flag is broadcast -- one 4-byte
    collective per step -- so that every rank drops / keeps the target key together."""
    n = len(counts)
    coef = torch.empty(n + 1, dtype=torch.float32, device=device)
    require_cuda(coef)
    kc = (ctypes.c_float * max(n, 1))(*[float(c) for c in counts])
    tp = tn = None
    if target_index >= 0 and poses is not None:
        tp, tn = _f32c(poses[0].detach()).reshape(-1, 16), _f32c(poses[1].detach()).reshape(-1, 16)
        require_cuda(tp, tn, rule_state)
    if shard is None:
        check(lib().codeps_depth_key_weights(ptr(tp), ptr(tn), ptr(rule_state), int(length), n, kc, int(target_index), 0.0, 0.0, 1,
                                             None, 0, ptr(coef), stream_ptr()), "codeps_depth_key_weights")
        return coef
    keep = None
    if shard.get("rule"):
        import torch.distributed as dist
        keep = torch.ones(1, dtype=torch.float32, device=device)
        if shard.get("owner"):
            check(lib().codeps_depth_key_weights(ptr(tp), ptr(tn), ptr(rule_state), int(length), n, kc, int(target_index), 0.0, 0.0, 1,
                                                 ptr(keep), 1, None, stream_ptr()), "codeps_depth_key_weights")
        dist.broadcast(keep, src=shard.get("owner_rank", 0), group=shard.get("group"))
    check(lib().codeps_depth_key_weights(None, None, None, int(length), n, kc, int(target_index), float(shard["global_total"]),
                                         float(shard["global_target"]), int(shard["world"]), ptr(keep), 2, ptr(coef), stream_ptr()),
          "codeps_depth_key_weights")
    coef._codeps_keep = keep
    return coef


class _WeightedSum(torch.autograd.Function):
    """out = sum_i w_i f_i loss_i / D in ONE launch (and one for all gradients) instead of ~3 scalar ATen kernels per term."""

    @staticmethod
    def forward(ctx, weights, factors, denom, *losses):
        require_cuda(*losses)
        n = len(losses)
        ls = [_f32c(l).reshape(1) for l in losses]
        dev = ls[0].device
        lp = (ctypes.c_void_p * n)(*[l.data_ptr() for l in ls])
        fp = None if factors is None else (ctypes.c_void_p * n)(*[None if f is None else f.data_ptr() for f in factors])
        w = (ctypes.c_float * n)(*[float(x) for x in weights])
        out = torch.empty(1, dtype=torch.float32, device=dev)
        check(lib().codeps_weighted_sum_fwd(lp, w, fp, n, ptr(denom), ptr(out), stream_ptr()), "codeps_weighted_sum_fwd")
        ctx.pack = (w, fp, factors, denom, n)          # `factors` / `denom` tensors are kept alive with the ctypes arrays
        return out[0]

    @staticmethod
    def backward(ctx, g):
        w, fp, _factors, denom, n = ctx.pack
        gv = torch.empty(n, dtype=torch.float32, device=g.device)
        gp = (ctypes.c_void_p * n)(*[gv.data_ptr() + 4 * i if ctx.needs_input_grad[3 + i] else None for i in range(n)])
        check(lib().codeps_weighted_sum_bwd(ptr(_f32c(g).reshape(1)), w, fp, n, ptr(denom), gp, stream_ptr()),
              "codeps_weighted_sum_bwd")
        return (None, None, None, *[gv[i] if ctx.needs_input_grad[3 + i] else None for i in range(n)])


def weighted_sum(losses, weights, factors=None, denom=None):
    """sum_i weights[i] * factors[i] * losses[i] / denom over 0-dim device losses (scripts/adapt_codeps.py:274-288;
    algos/depth.py:562-577).  weights: host floats; factors: optional per-term 0-dim DEVICE tensors (None = 1);
    denom: optional 0-dim DEVICE tensor (<= 0 -> result 0, zero gradients)."""
    if len(losses) != len(weights) or (factors is not None and len(factors) != len(losses)):
        raise CodepsHipError("weighted_sum: one weight (and one factor) per loss")
    return _WeightedSum.apply(list(weights), None if factors is None else list(factors), denom, *losses)


# ---------------------------------------------------------------------------------------- optimiser
def adam_ema_step(params, grads, exp_avg, exp_avg_sq, step, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0,
                  ema_params=None, ema_alpha=0.99):
    """One fused Adam (+EMA) step on flat fp32 buffers (C ABI: codeps_adam_ema_step)."""
    require_cuda(params, grads, exp_avg, exp_avg_sq, ema_params)
    check(lib().codeps_adam_ema_step(ptr(params), ptr(grads), ptr(exp_avg), ptr(exp_avg_sq), ptr(ema_params), params.numel(),
                                     int(step), lr, betas[0], betas[1], eps, grad_scale, ema_alpha, stream_ptr()),
          "codeps_adam_ema_step")
    invalidate_weight_cache()


# ---------------------------------------------------------------------------------------- panoptic fusion
def panoptic_fusion(semantic, center, offset, thing_list, label_divisor=1000, void_label=255, threshold=0.1, nms_kernel=7,
                    top_k=200):
    """Device-side panoptic fusion (algos/instance_seg.py:207-258).  semantic [B,H,W] uint8/int64, center
    [B,1,H,W], offset [B,2,H,W] -> (panoptic int16 [B,H,W], instance [B,H,W] (semantic's dtype),
    centers int32 [B,256,2], state int32 [B,4] = (#candidates, #centres, ., .)).  No host sync."""
    require_cuda(semantic, center, offset)
    b, h, w = semantic.shape
    # the uint8 form of an int64 label map made by channel_argmax travels with it (no conversion pass)
    sem8 = semantic
    if semantic.dtype != torch.uint8:
        twin = getattr(semantic, "_codeps_u8", None)
        # an in-place edit of the int64 map after channel_argmax (e.g. masking labels) bumps its version: the twin is stale then
        sem8 = twin[0] if (twin is not None and twin[1] == semantic._version and tuple(twin[0].shape) == (b, h, w)) else semantic.to(torch.uint8)
    sem8 = sem8.contiguous()
    center, offset = center.detach(), offset.detach()
    if center.dtype != torch.float32 or offset.dtype != torch.float32:
        center, offset = center.float(), offset.float()
    # the maps are read in place through their strides (NCHW planes, or channels of the instance head's NHWC outputs)
    cs, os_ = center.stride(), offset.stride()
    if not (cs[2] == w * cs[3] and os_[2] == w * os_[3] and cs[3] >= 1 and os_[3] >= 1 and tuple(center.shape) == (b, 1, h, w)
            and tuple(offset.shape) == (b, 2, h, w)):
        center, offset = center.contiguous(), offset.contiguous()
        cs, os_ = center.stride(), offset.stride()
    mask = 0
    for c in thing_list:
        mask |= 1 << int(c)
    dev = semantic.device
    pan = torch.empty(b, h, w, dtype=torch.int16, device=dev)
    inst = torch.empty(b, h, w, dtype=torch.uint8, device=dev)
    inst64 = torch.empty(b, h, w, dtype=torch.int64, device=dev) if semantic.dtype == torch.int64 else None
    centers = torch.empty(b, 256, 2, dtype=torch.int32, device=dev)
    state = torch.empty(b, 4, dtype=torch.int32, device=dev)
    nbytes = lib().codeps_panoptic_workspace_bytes(b, h, w)
    ws = _workspace(nbytes, dev, "panoptic")
    check(lib().codeps_panoptic_fusion_strided(ptr(sem8), ptr(center), cs[0], cs[3], ptr(offset), os_[0], os_[1], os_[3], b, h, w, mask,
                                               label_divisor, void_label, threshold, nms_kernel, top_k, ptr(pan), ptr(inst), ptr(inst64),
                                               ptr(centers), ptr(state), ptr(ws), ws.numel(), stream_ptr()),
          "codeps_panoptic_fusion_strided")
    if inst64 is not None:
        inst = inst64
    elif semantic.dtype != torch.uint8:
        inst = inst.to(semantic.dtype)
    return pan, inst, centers, state


# ------------------------------------------------------------------------------------------------ mix-up
_C2C_DTYPES = {torch.float32: 0, torch.uint8: 1, torch.int32: 2}


def warp_c2c(in_maps, intr_in, intr_out, out_hw, interp_mode="bilinear", padding_mode="border", depth_val=1.0):
    """Mixup.warp_c2c (datasets/mixup.py:211-229): in_maps [B,C,hs,ws] (or [B,hs,ws] -> C = 1) seen by cameras intr_in
    [B,4] -> [B,C,ho,wo] as seen by cameras intr_out; float64 sampling arithmetic, stored in in_maps' dtype (fp32 for
    images; uint8 / int32 / int64 label maps, nearest only)."""
    require_cuda(in_maps, intr_in, intr_out)
    if interp_mode not in ("bilinear", "nearest") or padding_mode not in ("zeros", "border"):
        raise ValueError(f"warp_c2c: interp_mode {interp_mode!r} / padding_mode {padding_mode!r} not supported")
    x = in_maps.unsqueeze(1) if in_maps.dim() == 3 else in_maps
    orig = x.dtype
    if orig == torch.int64 or orig == torch.int16:
        x = x.to(torch.int32)
    elif orig not in _C2C_DTYPES:
        x = x.float()
    if interp_mode == "bilinear" and x.dtype != torch.float32:
        raise TypeError("warp_c2c: bilinear sampling of an integer map")
    x = x.contiguous()
    b, c, hs, ws = x.shape
    ho, wo = int(out_hw[0]), int(out_hw[1])
    out = torch.empty(b, c, ho, wo, dtype=x.dtype, device=x.device)
    check(lib().codeps_warp_c2c(ptr(x), _C2C_DTYPES[x.dtype], ptr(_f32c(intr_in)), ptr(_f32c(intr_out)), b, c, hs, ws, ho, wo,
                                int(interp_mode == "nearest"), int(padding_mode == "border"), float(depth_val), ptr(out), stream_ptr()),
          "codeps_warp_c2c")
    return out if out.dtype == orig else out.to(orig)


def cut_mixup(img_src, img_tgt, sem_src, sem_tgt, inst_src, inst_tgt, rects, intr_src=None, intr_tgt=None, depth_val=1.0):
    """Mixup.cut_mixup (datasets/mixup.py:314-344) fused with do_mixup's camera-to-camera warps (mixup.py:431-441).
    img_* [B,3,.,.] fp32, label maps [B,.,.] int32, rects [B,nseg,4] int32 (y0, y1, x0, x1; end-exclusive)
    -> (rgb_mixup fp32 [B,3,H,W], semantic int32 [B,H,W], instance int32 [B,H,W])."""
    require_cuda(img_src, img_tgt, sem_src, sem_tgt, inst_src, inst_tgt, rects)
    b, _, h, w = img_src.shape
    ht, wt = img_tgt.shape[-2:]
    img_src, img_tgt = _f32c(img_src), _f32c(img_tgt)
    labels = [t.contiguous() if t.dtype == torch.int32 else t.to(torch.int32).contiguous() for t in (sem_src, sem_tgt, inst_src, inst_tgt)]
    rects = rects.to(torch.int32).contiguous()
    dev = img_src.device
    rgb = torch.empty(b, 3, h, w, dtype=torch.float32, device=dev)
    sem = torch.empty(b, h, w, dtype=torch.int32, device=dev)
    inst = torch.empty(b, h, w, dtype=torch.int32, device=dev)
    geom = intr_src is not None and intr_tgt is not None
    check(lib().codeps_cut_mixup(ptr(img_src), ptr(img_tgt), ptr(labels[0]), ptr(labels[1]), ptr(labels[2]), ptr(labels[3]),
                                 ptr(_f32c(intr_src)) if geom else None, ptr(_f32c(intr_tgt)) if geom else None, ptr(rects),
                                 rects.shape[1], b, h, w, ht, wt, float(depth_val), ptr(rgb), ptr(sem), ptr(inst), stream_ptr()),
          "codeps_cut_mixup")
    return rgb, sem, inst


def conf_instance_stats(instance):
    """Per instance id 1..255 of one label map [H,W] (the teacher's panoptic fusion): int64 [256,5] = (pixels, sum y, sum x,
    min x, max x) on the device (C ABI: codeps_conf_instance_stats; datasets/mixup.py:572-581,590-592)."""
    require_cuda(instance)
    inst = instance.contiguous() if instance.dtype == torch.int32 else instance.to(torch.int32).contiguous()
    h, w = inst.shape
    table = torch.empty(lib().codeps_conf_stats_bytes() // 8, dtype=torch.int64, device=inst.device)
    check(lib().codeps_conf_instance_stats(ptr(inst), h, w, ptr(table), stream_ptr()), "codeps_conf_instance_stats")
    return table.view(256, 5)


def conf_place(things, row, want_row=False):
    """kornia.contrib.distance_transform of the uint8 thing mask [H,W] along `row` and the first column of its maximum
    (datasets/mixup.py:583-587), closed form, no iteration (C ABI: codeps_conf_place).  -> (argmax int32 [1] on the device,
    dist_row fp32 [W] or None)."""
    require_cuda(things)
    assert things.dtype == torch.uint8 and things.is_contiguous()
    h, w = things.shape
    dev = things.device
    argmax = torch.empty(1, dtype=torch.int32, device=dev)
    dist = torch.empty(w, dtype=torch.float32, device=dev) if want_row else None
    nbytes = lib().codeps_conf_place_workspace_bytes(w)
    ws = _workspace(nbytes, dev, "conf_place")
    check(lib().codeps_conf_place(ptr(things), h, w, int(row), ptr(dist), ptr(argmax), ptr(ws), nbytes, stream_ptr()), "codeps_conf_place")
    return argmax, dist


def conf_apply(instance, cid, table, cx, argmax, shift, dst, things):
    """Place instance `cid` (datasets/mixup.py:586-645): shift[cid] from the placement column, its centre column and its column
    range; dst / things updated in place (C ABI: codeps_conf_apply)."""
    h, w = instance.shape
    check(lib().codeps_conf_apply(ptr(instance), h, w, int(cid), ptr(table), int(cx), ptr(argmax), ptr(shift), ptr(dst), ptr(things),
                                  stream_ptr()), "codeps_conf_apply")


def conf_paste(img_src, img_tgt, sem_src, sem_tgt, inst_src, inst_conf, dst, shift):
    """The mixed item of one image (datasets/mixup.py:366-381; C ABI: codeps_conf_paste).  img_* [C,H,W] fp32, label maps int32
    [H,W], dst uint8 [H,W], shift int32 [256] -> (rgb fp32 [C,H,W], semantic int32 [H,W], instance int32 [H,W])."""
    c, h, w = img_src.shape
    dev = img_src.device
    rgb = torch.empty(c, h, w, dtype=torch.float32, device=dev)
    sem = torch.empty(h, w, dtype=torch.int32, device=dev)
    inst = torch.empty(h, w, dtype=torch.int32, device=dev)
    check(lib().codeps_conf_paste(ptr(img_src), ptr(img_tgt), ptr(sem_src), ptr(sem_tgt), ptr(inst_src), ptr(inst_conf), ptr(dst), ptr(shift),
                                  c, h, w, ptr(rgb), ptr(sem), ptr(inst), stream_ptr()), "codeps_conf_paste")
    return rgb, sem, inst


def offset_center(instance, sigma=8.0):
    """Mixup.get_off_cen (datasets/mixup.py:243-257 -> datasets/dataset.py:126-177) on the device: instance [B,(1,)H,W]
    integer ids -> (center fp32 [B,1,H,W], offset fp32 [B,2,H,W], status int32 [B] = #instances or -1 on overflow).
    No host sync: read `status` when convenient (Mixup.check does)."""
    require_cuda(instance)
    inst = instance.reshape((instance.shape[0],) + tuple(instance.shape[-2:]))
    inst = inst.contiguous() if inst.dtype == torch.int32 else inst.to(torch.int32).contiguous()
    b, h, w = inst.shape
    dev = inst.device
    center = torch.empty(b, 1, h, w, dtype=torch.float32, device=dev)
    offset = torch.empty(b, 2, h, w, dtype=torch.float32, device=dev)
    status = torch.empty(b, dtype=torch.int32, device=dev)
    nbytes = lib().codeps_offset_center_workspace_bytes(b)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(lib().codeps_offset_center(ptr(inst), b, h, w, float(sigma), ptr(center), ptr(offset), ptr(status), ptr(ws), nbytes,
                                     stream_ptr()), "codeps_offset_center")
    return center, offset, status
