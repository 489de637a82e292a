"""The one-launch weight pack of a step (codeps_conv2d_pack_weights_multi, slabs through LDS) writes exactly the bytes of the
one-by-one pack (codeps_conv2d_pack_weights): fp32 packs, data-gradient packs and the pre-split bf16 copies, for every kernel
shape of the networks (1x1, 3x3, 4x4 up-dgrad weights, the 7x4 stem form, channel counts that need padding)."""
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
        assert torch.equal(got_f.view(torch.int32), ref_f.view(torch.int32)), f"forward pack of {tuple(p.shape)} differs"
        assert torch.equal(got_d.view(torch.int32), ref_d.view(torch.int32)), f"data-gradient pack of {tuple(p.shape)} differs"
        assert not torch.equal(got_f, wf0), "the refresh did not see the new weights"
    F._pack_cache_train.clear()
    F._pack_table.update(dev=None, n=0, groups=0, entries=[], epoch=-1)
