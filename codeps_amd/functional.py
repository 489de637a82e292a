"""torch.autograd.Function wrappers over the C ABI (include/codeps_hip.h).

PyTorch supplies device memory (caching allocator), the current HIP stream and the autograd tape;
all arithmetic happens in libcodeps_hip.so.  Every function raises on CPU tensors -- no fallback.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, lib, ptr, require_cuda, stream_ptr

_seed_counter = [0x5EED]


def _f32c(t):
    """fp32 + contiguous without copying when already so."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class _PhotometricLoss(torch.autograd.Function):
    """Fused multi-scale photometric loss (C ABI: codeps_photometric_{fwd,bwd})."""

    @staticmethod
    def forward(ctx, img_t, img_m, img_p, depth, pose_m, pose_p, intr, noise, num_scales, alpha, seed):
        require_cuda(img_t, img_m, img_p, depth, pose_m, pose_p, intr, noise)
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
        check(lib().codeps_photometric_fwd(ctypes.byref(desc), ptr(img_t), ptr(img_m), ptr(img_p), ptr(depth),
                                           ptr(pose_m), ptr(pose_p), ptr(intr), ptr(noise), ptr(loss), ptr(ws),
                                           nbytes, stream_ptr()), "codeps_photometric_fwd")
        ctx.save_for_backward(img_t, img_m, img_p, depth, pose_m, pose_p, intr, ws)
        ctx.desc = desc
        return loss[0]

    @staticmethod
    def backward(ctx, grad_loss):
        img_t, img_m, img_p, depth, pose_m, pose_p, intr, ws = ctx.saved_tensors
        g = _f32c(grad_loss).reshape(1)
        gd = torch.empty_like(depth)
        gpm, gpp = torch.empty_like(pose_m), torch.empty_like(pose_p)
        check(lib().codeps_photometric_bwd(ctypes.byref(ctx.desc), ptr(img_t), ptr(img_m), ptr(img_p), ptr(depth),
                                           ptr(pose_m), ptr(pose_p), ptr(intr), ptr(g), ptr(gd), ptr(gpm), ptr(gpp),
                                           ptr(ws), ws.numel(), stream_ptr()), "codeps_photometric_bwd")
        return None, None, None, gd, gpm, gpp, None, None, None, None, None


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
                                  int(seed))
