"""The one-launch weight pack of a step (codeps_conv2d_pack_weights_multi, slabs through LDS) writes exactly the bytes of the
one-by-one pack (codeps_conv2d_pack_weights): fp32 packs, data-gradient packs and the pre-split bf16 copies, for every kernel
shape of the networks (1x1, 3x3, 4x4 up-dgrad weights, the 7x4 stem form, channel counts that need padding).  Since both entry
points run the same slab code, both are also compared with the layout written out in torch: [row][tap][col] fp32, then
[row][K group of 32][piece][32] bf16 with the pieces of the exact 3-way split."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SHAPES = [(64, 64, 3, 3), (256, 128, 3, 3), (512, 512, 3, 3), (19, 32, 3, 3), (16, 48, 1, 1), (1024, 256, 1, 1), (128, 2048, 1, 1),
          (64, 16, 7, 4), (32, 96, 4, 4), (144, 64, 1, 1), (6, 256, 1, 1), (256, 1280, 3, 3)]


def test_multi_pack_equals_single_packs():
    from codeps_amd import functional as F
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(*s, device=DEV)) for s in SHAPES]
    F.invalidate_weight_cache(True)
    F._pack_cache_train.clear()
    F._pack_table.update(dev=None, n=0, groups=0, entries=[], epoch=-1)
    want = []
    for p in params:
        co, ci = p.shape[:2]
        wf, wd = F.pack_weights(p, F.pad16(co), F.pad16(ci), True)        # one-by-one path creates the persistent buffers
        want.append((wf.clone(), wd.clone()))
    assert F.build_pack_table() == 2 * len(params)
    with torch.no_grad():
        for p in params:
            p.mul_(1.5)                                                    # new weights: the packs are stale now
    F.invalidate_weight_cache()
    F.refresh_weight_packs()                                               # ONE launch re-fills every pack in place
    torch.cuda.synchronize()
    for p, (wf0, wd0) in zip(params, want):
        co, ci = p.shape[:2]
        ent = F._pack_cache_train[(id(p), F.pad16(co), F.pad16(ci))]
        got_f, got_d = ent[3].clone(), ent[4].clone()
        # reference: the single-parameter kernels on the NEW weights, into fresh buffers
        F._pack_cache_train.pop((id(p), F.pad16(co), F.pad16(ci)))
        ref_f, ref_d = F.pack_weights(p, F.pad16(co), F.pad16(ci), True)
        torch.cuda.synchronize()
        # the defined part of a pack: fp32, three bf16 pieces, two fp16 pieces, the scale slot (behind it: scratch of the pack's work groups)
        total = F.pad16(co) * F.pad16(ci) * p.shape[2] * p.shape[3]
        off16 = total + (total * 6 + 3) // 4
        from codeps_amd._lib import lib
        pieces = (off16, off16 + total + 4) if lib().codeps_conv2d_set_arith(-1) == 16 else (total, off16)    # the fp16 pieces + scale slot, or the bf16 pieces
        parts = [(0, total)] + ([pieces] if got_f.numel() > total else [])
        for lo, hi in parts:
            assert torch.equal(got_f[lo:hi].view(torch.int32), ref_f[lo:hi].view(torch.int32)), f"forward pack of {tuple(p.shape)} differs"
            assert torch.equal(got_d[lo:hi].view(torch.int32), ref_d[lo:hi].view(torch.int32)), f"data-gradient pack of {tuple(p.shape)} differs"
        assert not torch.equal(got_f, wf0), "the refresh did not see the new weights"
    F._pack_cache_train.clear()
    F._pack_table.update(dev=None, n=0, groups=0, entries=[], epoch=-1)


def _reference_pack(w, rows_pad, cols_pad, dgrad):
    """fp32 pack [rows_pad][taps][cols_pad] (+ split copy or None) of an OIHW tensor, in torch."""
    co, ci, kh, kw = w.shape
    t = w.reshape(co, ci, kh * kw)
    t = t.permute(1, 2, 0) if dgrad else t.permute(0, 2, 1)                 # [row][tap][col]
    out = torch.zeros(rows_pad, kh * kw, cols_pad, device=w.device)
    out[:t.shape[0], :, :t.shape[2]] = t
    flat = out.reshape(-1)
    taps = kh * kw
    if not (cols_pad % 32 == 0 or (cols_pad == 16 and (taps * cols_pad) % 32 == 0)):
        return flat, None
    p0 = flat.to(torch.bfloat16)
    r1 = flat - p0.float()
    p1 = r1.to(torch.bfloat16)
    p2 = (r1 - p1.float()).to(torch.bfloat16)
    pieces = torch.stack([p0, p1, p2], 0).view(3, -1, 32).permute(1, 0, 2).contiguous()      # [group of 32][piece][32]
    return flat, pieces.view(torch.int16).reshape(-1)


@pytest.mark.parametrize("shape", SHAPES + [(32, 32, 7, 1)])
def test_single_pack_matches_the_layout_written_out_in_torch(shape):
    from codeps_amd import functional as F
    torch.manual_seed(1)
    co, ci = shape[:2]
    p = torch.nn.Parameter(torch.randn(*shape, device=DEV) * torch.exp2(torch.randint(-8, 9, shape, device=DEV).float()))
    cop, cip = F.pad16(co), F.pad16(ci)
    from codeps_amd._lib import lib
    prev = lib().codeps_conv2d_set_arith(3)                    # the three bf16 pieces are written under arithmetic 3 ...
    try:
        wf3, wd3 = (t.clone() for t in F.pack_weights(p, cop, cip, True))
    finally:
        lib().codeps_conv2d_set_arith(prev)
    prev = lib().codeps_conv2d_set_arith(16)                   # ... the two fp16 pieces under 16 (the stamp of a pack carries the arithmetic)
    try:
        wf, wd = (t.clone() for t in F.pack_weights(p, cop, cip, True))
    finally:
        lib().codeps_conv2d_set_arith(prev)
    torch.cuda.synchronize()
    for got, got3, dgrad, rows, cols in ((wf, wf3, False, cop, cip), (wd, wd3, True, cip, cop)):
        ref, pieces = _reference_pack(p.detach(), rows, cols, dgrad)
        n = ref.numel()
        assert torch.equal(got[:n].view(torch.int32), ref.view(torch.int32)), f"fp32 pack (dgrad={dgrad}) of {shape}"
        assert torch.equal(got3[:n].view(torch.int32), ref.view(torch.int32)), f"fp32 pack under arithmetic 3 (dgrad={dgrad}) of {shape}"
        if pieces is not None:
            have = got3[n:].view(torch.int16)[:pieces.numel()]
            assert torch.equal(have, pieces), f"split copy (dgrad={dgrad}) of {shape}"
            # the two fp16 pieces of w 2^kb ([group of 32][hi | lo][32]) and the scale slot (2^-kb, 2^kb, max |w|, 0): csrc/conv.hip, split_f16
            off = n + (n * 6 + 3) // 4
            inv, sc, amax = (float(v) for v in got[off + n:off + n + 3])
            assert amax == float(ref.abs().max()) and inv * sc == 1.0 and 2.0 ** 14 <= amax * sc < 2.0 ** 15, (amax, sc)
            h = got[off:off + n].view(torch.float16).view(-1, 2, 32).float()
            hi, lo = h[:, 0].reshape(-1), h[:, 1].reshape(-1)
            want_hi = (ref * sc).to(torch.float16)
            assert torch.equal(hi, want_hi.float()), f"fp16 hi pieces (dgrad={dgrad}) of {shape}"
            assert torch.equal(lo, (ref * sc - want_hi.float()).to(torch.float16).float()), f"fp16 lo pieces (dgrad={dgrad}) of {shape}"
            err = ((hi.double() + lo.double()) * inv - ref.double()).abs()
            ok = (err <= 2.0 ** -21 * ref.double().abs()) | (err <= 2.0 ** -24 * inv)       # (lo in fp16's subnormal range: absolute 2^-25 of the scaled value)
            assert bool(ok.all()), "hi + lo reproduces w to 2^-22 (2^-25 / 2^kb absolute for elements 2^17 below the largest)"
    F._pack_cache_train.clear()
