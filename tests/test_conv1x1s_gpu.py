"""Pre-split activations (csrc/conv1x1s.hip): codeps_split_pack_rows reproduces every fp32 value exactly as the sum of its three
bf16 pieces, in the layout of the conv engine's pre-split weight pack; codeps_conv1x1_split_fwd (both operands by LDS DMA) agrees
with the conv engine's 1x1 path and with a float64 reference at fp32 rounding level, ragged row counts included."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _split(x2d):
    from codeps_amd import _lib
    rows, c = x2d.shape
    out = torch.empty(rows * c * 3, dtype=torch.int16, device=DEV)
    _lib.check_diag(_lib.diag_lib().codeps_split_pack_rows(_lib.ptr(x2d), rows, c, x2d.stride(0), _lib.ptr(out), _lib.stream_ptr()), "codeps_split_pack_rows")
    return out


def test_split_pack_rows_is_exact_and_laid_out_like_the_weight_pack():
    from codeps_amd import functional as F
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(77, 96, generator=g) * torch.exp2(torch.randint(-20, 20, (77, 96), generator=g).float())).to(DEV)
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 1.0e30, -1.0e30, 1.0e-30, 65504.0])     # (exact while the third piece stays normal: |x| > 1e-33)
    s = _split(x).view(77, 3, 3, 32)                                    # [row][group][piece][32]
    pieces = (s.to(torch.int32) << 16).view(torch.float32)              # bf16 bits -> fp32
    back = pieces[:, :, 0] + pieces[:, :, 1] + pieces[:, :, 2]          # summed smallest-last is not needed: the sum is exact
    bad = (back.reshape(77, 96) != x).nonzero()
    assert bad.numel() == 0, (bad[:4], x[bad[0, 0], bad[0, 1]].item(), back.reshape(77, 96)[bad[0, 0], bad[0, 1]].item())
    # a strided view (first 64 of 96 channels)
    s2 = _split(x[:, :64]).view(77, 2, 3, 32)
    assert torch.equal(s2, s[:, :2])
    # same bytes as the weight pack's pre-split part for a 1x1 weight (rows = output channels)
    w = x[:64].contiguous().view(64, 96, 1, 1)
    from codeps_amd._lib import lib as _l
    prev = _l().codeps_conv2d_set_arith(3)                     # (the bf16 pieces are the pack's content under arithmetic 3)
    try:
        wf, _ = F.pack_weights(w, 64, 96, False)
    finally:
        _l().codeps_conv2d_set_arith(prev)
    assert torch.equal(wf.view(-1)[64 * 96:].view(torch.int16)[: 64 * 96 * 3], _split(x[:64].contiguous()))


@pytest.mark.parametrize("rows,ci,co,act", [(8 * 32 * 64, 1024, 256, 1), (5000, 256, 1024, 2), (130, 64, 64, 0), (4096, 2048, 512, 1)])
def test_conv1x1_split_fwd_matches_engine_and_float64(rows, ci, co, act):
    from codeps_amd import functional as F, _lib
    g = torch.Generator().manual_seed(rows + ci)
    x = torch.randn(rows, ci, generator=g)
    w = torch.randn(co, ci, generator=g) * 0.05
    scale, shift, res = torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g), torch.randn(rows, co, generator=g)
    pre = (x.double() @ w.double().t()) * scale.double() + shift.double() + res.double()
    ref = pre if act == 0 else (pre.clamp_min(0) if act == 1 else torch.where(pre > 0, pre, torch.expm1(pre)))
    xd, wd = x.to(DEV), w.to(DEV).view(co, ci, 1, 1)
    sc_d, sh_d, res_d = scale.to(DEV), shift.to(DEV), res.to(DEV)       # (held: raw pointers go to the library)
    prev0 = _lib.lib().codeps_conv2d_set_arith(3)            # the prototype reads the three-bf16-piece pack
    try:
        wf, _ = F.pack_weights(wd, co, ci, False)
        wf = wf.clone()
    finally:
        _lib.lib().codeps_conv2d_set_arith(prev0)
    xs, w_split = _split(xd), wf.view(-1)[co * ci:]
    y = torch.empty(rows, co, device=DEV)
    _lib.check_diag(_lib.diag_lib().codeps_conv1x1_split_fwd(_lib.ptr(xs), rows, ci, co, _lib.ptr(w_split), _lib.ptr(sc_d), _lib.ptr(sh_d), _lib.ptr(res_d), act,
                                                   _lib.ptr(y), _lib.stream_ptr()), "codeps_conv1x1_split_fwd")
    prev = _lib.lib().codeps_conv2d_set_arith(3)       # the prototype shares the three-bf16-piece arithmetic: compare it with the engine on that arithmetic
    try:
        eng = F.conv2d(xd.view(1, 1, rows, ci), wd, scale=sc_d, shift=sh_d, residual=res_d.view(1, 1, rows, co),
                       act=[None, "relu", "elu"][act]).view(rows, co)
    finally:
        _lib.lib().codeps_conv2d_set_arith(prev)
    err = lambda a: float((a.double().cpu() - ref).norm() / ref.norm())     # noqa: E731
    e_new, e_eng = err(y), err(eng)
    assert e_new <= max(2.0 * e_eng, 2e-7), (e_new, e_eng)
    assert float((y - eng).abs().max()) <= 1e-4 * float(ref.abs().max())
    # bad arguments are refused
    assert _lib.diag_lib().codeps_conv1x1_split_fwd(_lib.ptr(y), rows, 48, co, _lib.ptr(y), None, None, None, 0, _lib.ptr(y), _lib.stream_ptr()) != 0
