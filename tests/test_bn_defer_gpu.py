"""Train-mode BatchNorm + ReLU applied on the consumer's operand load (F.batch_norm_train(defer=True) -> F.conv2d; C ABI
codeps_bn_in_affine / codeps_conv2d_fwd_in / codeps_conv2d_wgrad_in / codeps_in_affine_apply).  The kernels compute
relu((x - mean) invstd gamma + beta) with the apply pass's own expression where they stage their operand, so everything is compared BIT FOR BIT
with the path that writes the normalised tensor (CODEPS_BN_DEFER=0 / defer=False): forward, running statistics, every gradient.
Reference semantics: conv -> bn -> relu -> 1x1 conv of torchvision's Bottleneck behind models/resnet_encoder.py:115-126, and the depthwise
stage of models/instance_head.py:38-46."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(defer, b, h, w, c, co, groups, k_prod, seed, arith=None):
    """conv(k_prod) -> BN(train) -> ReLU -> 1x1 conv(c -> co); returns outputs, running stats and all gradients"""
    from codeps_amd import functional as F
    from codeps_amd._lib import lib
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, h, w, 32, generator=g).to(DEV).requires_grad_(True)
    w1 = (torch.randn(c, 32, k_prod, k_prod, generator=g) * 0.1).to(DEV).requires_grad_(True)
    w2 = (torch.randn(co, c, 1, 1, generator=g) * (2.0 / c) ** 0.5).to(DEV).requires_grad_(True)
    gamma = (0.5 + torch.rand(c, generator=g)).to(DEV).requires_grad_(True)
    beta = (0.3 * torch.randn(c, generator=g)).to(DEV).requires_grad_(True)
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    gy = torch.randn(b, h, w, F.pad16(co), generator=g).to(DEV)
    z = F.conv2d(x, w1, pad=k_prod // 2, bn_stats=True)
    y = F.batch_norm_train(z, gamma, beta, rm, rv, 0.1, 1e-5, act="relu", groups=groups, defer=defer)
    deferred = getattr(y, "_codeps_defer", None) is not None
    out = F.conv2d(y, w2, exclusive=True)
    fwd_kernel = lib().codeps_conv2d_last_kernel().decode()
    out.backward(gy)
    torch.cuda.synchronize()
    return dict(out=out.detach(), rm=rm, rv=rv, dx=x.grad, dw1=w1.grad, dw2=w2.grad, dgamma=gamma.grad, dbeta=beta.grad), deferred, fwd_kernel


#                                   b  h    w    c    co  groups            producer kernel size
CASES = {"bottleneck_l1": (2, 64, 128, 64, 256, None, 3),
         "bottleneck_l3": (2, 16, 32, 256, 1024, None, 3),
         "groups": (4, 32, 64, 128, 128, [(0, 1), (1, 3), (3, 4)], 1),
         "instance_320": (2, 16, 32, 320, 256, None, 1),
         "wide_out_swap": (2, 32, 64, 512, 64, None, 1),
         "f32_path_144": (2, 32, 64, 144, 128, None, 1),          # 144 channels: the f32 GEMM kernel (K chunks of 16)
         "head_128_32": (2, 64, 128, 128, 32, None, 1),            # <= 32 output channels: wgrad_small_kernel's 1x1 form
         "head_groups_32": (3, 32, 64, 128, 32, [(0, 2), (2, 3)], 3),
         "ragged_hw": (2, 24, 40, 64, 64, None, 1)}                # 960 pixels per image: the consumer cannot fuse -> it materialises


@pytest.mark.parametrize("name", list(CASES))
def test_deferred_apply_is_bit_identical_to_the_written_tensor(name):
    b, h, w, c, co, groups, kp = CASES[name]
    ref, d0, _k = _run(False, b, h, w, c, co, groups, kp, 5)
    got, d1, kern = _run(True, b, h, w, c, co, groups, kp, 5)
    assert not d0 and d1, "defer=True did not defer (or defer=False did)"
    for key in ref:
        assert torch.equal(ref[key], got[key]), f"{name}: {key} differs (max {float((ref[key] - got[key]).abs().max()):.3e}); forward kernel {kern}"


def test_deferred_path_runs_the_fused_kernels_and_skips_the_apply_pass(monkeypatch):
    """A served shape: the consumer never materialises (F.in_affine_apply is not called) and its weight gradient is the split-operand 1x1
    kernel reading BatchNorm's input; a shape it cannot serve (3x3) is refused by the query and materialised by F.conv2d."""
    from codeps_amd import functional as F
    from codeps_amd import _lib
    b, h, w, c, co, _groups, _kp = CASES["bottleneck_l1"]
    d = _lib.ConvDesc(b, h, w, c, 0, 0, co, h, w, 1, 1, 1, 0, 0, 0, 1, 0, 0, 0)
    assert _lib.lib().codeps_conv2d_in_affine_ok(ctypes.byref(d), 0) == 1 and _lib.lib().codeps_conv2d_in_affine_ok(ctypes.byref(d), 1) == 1
    d3 = _lib.ConvDesc(b, h, w, c, 0, 0, co, h, w, 3, 3, 1, 1, 0, 0, 1, 1, 0, 0)
    assert _lib.lib().codeps_conv2d_in_affine_ok(ctypes.byref(d3), 0) == 0, "3x3 consumers are not served"
    calls = []
    real = F.in_affine_apply
    monkeypatch.setattr(F, "in_affine_apply", lambda x, t: (calls.append(tuple(x.shape)), real(x, t))[1])
    g = torch.Generator().manual_seed(2)
    z = torch.randn(b, h, w, c, generator=g).to(DEV)
    gamma, beta = torch.ones(c, device=DEV, requires_grad=True), torch.zeros(c, device=DEV, requires_grad=True)
    for k, co, served in ((1, 256, True), (1, 32, True), (3, 256, False)):
        calls.clear()
        wt = (torch.randn(co, c, k, k, generator=g) * 0.05).to(DEV).requires_grad_(True)
        y = F.batch_norm_train(z, gamma, beta, torch.zeros(c, device=DEV), torch.ones(c, device=DEV), act="relu", defer=True)
        out = F.conv2d(y, wt, pad=k // 2, exclusive=True)
        fwd = _lib.lib().codeps_conv2d_last_kernel().decode()
        out.sum().backward()
        last = _lib.lib().codeps_conv2d_last_kernel().decode()
        if served:
            assert calls == [] and "conv_igemm_kernel" in fwd and ("wgrad_1x1s_kernel" if co > 32 else "wgrad_small_kernel") in last, (calls, fwd, last)
        else:
            assert len(calls) == 1, calls                    # made once in the forward and kept for the weight gradient


def test_in_affine_apply_equals_the_apply_pass():
    from codeps_amd import functional as F
    g = torch.Generator().manual_seed(3)
    b, h, w, c = 3, 16, 32, 48
    z = torch.randn(b, h, w, c, generator=g).to(DEV)
    gamma, beta = (0.5 + torch.rand(c, generator=g)).to(DEV), torch.randn(c, generator=g).to(DEV)
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    groups = [(0, 2), (2, 3)]
    y_ref = F.batch_norm_train(z, gamma, beta, rm.clone(), rv.clone(), 0.1, 1e-5, act="relu", groups=groups)
    rm2, rv2 = rm.clone(), rv.clone()
    y_def = F.batch_norm_train(z, gamma, beta, rm2, rv2, 0.1, 1e-5, act="relu", groups=groups, defer=True)
    assert y_def.data_ptr() == z.data_ptr()
    y = F.in_affine_apply(y_def, y_def._codeps_defer)
    assert torch.equal(y, y_ref)
    rm3, rv3 = rm.clone(), rv.clone()
    F.batch_norm_train(z, gamma, beta, rm3, rv3, 0.1, 1e-5, act="relu", groups=groups)
    assert torch.equal(rm2, rm3) and torch.equal(rv2, rv3)


class _OneRank:
    """SyncBatchNormReducer's interface on a 'world' of one rank: the cross-rank branch of F.batch_norm_train (all-gather packet ->
    codeps_bn_sync_combine -> apply), without a process group."""

    def gather(self, packet):
        return packet.contiguous().unsqueeze(0).clone()

    def sum(self, sums):
        return sums


@pytest.mark.parametrize("groups", [None, [(0, 1), (1, 3)]])
def test_deferred_apply_under_sync_batchnorm(groups):
    """SyncBatchNorm (misc/train_utils.py:50-51): the statistics are combined across ranks, the running statistics are updated by the combine
    kernel, and the apply pass -- deferred or not -- sees the combined (mean, invstd): same bits either way, forward and backward."""
    from codeps_amd import functional as F
    g = torch.Generator().manual_seed(12)
    b, h, w, c, co = 3, 32, 64, 128, 64
    res = []
    for defer in (False, True):
        x = torch.randn(b, h, w, 32, generator=torch.Generator().manual_seed(1)).to(DEV).requires_grad_(True)
        w1 = (torch.randn(c, 32, 1, 1, generator=torch.Generator().manual_seed(2)) * 0.2).to(DEV).requires_grad_(True)
        w2 = (torch.randn(co, c, 1, 1, generator=torch.Generator().manual_seed(3)) * 0.1).to(DEV).requires_grad_(True)
        gamma = torch.full((c,), 1.3, device=DEV, requires_grad=True)
        beta = torch.full((c,), -0.1, device=DEV, requires_grad=True)
        rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        z = F.conv2d(x, w1, bn_stats=True)
        y = F.batch_norm_train(z, gamma, beta, rm, rv, 0.1, 1e-5, act="relu", reduce_fn=_OneRank(), groups=groups, defer=defer)
        assert (getattr(y, "_codeps_defer", None) is not None) == defer
        out = F.conv2d(y, w2, exclusive=True)
        out.backward(torch.randn(out.shape, generator=torch.Generator().manual_seed(4)).to(DEV))
        torch.cuda.synchronize()
        res.append((out.detach(), rm, rv, x.grad, w1.grad, w2.grad, gamma.grad, beta.grad))
    for a, bb in zip(*res):
        assert torch.equal(a, bb), float((a - bb).abs().max())
    del g


def test_c_abi_refuses_shapes_it_does_not_serve():
    """codeps_conv2d_fwd_in / _wgrad_in on a shape codeps_conv2d_in_affine_ok rejects: an error code and a message, no launch."""
    from codeps_amd import _lib
    lib = _lib.lib()
    b, h, w, c, co = 1, 16, 16, 64, 64
    x = torch.zeros(b, h, w, c, device=DEV)
    y = torch.zeros(b, h, w, co, device=DEV)
    tab = torch.zeros(b, 4, c, device=DEV)
    wf = torch.zeros(lib.codeps_conv2d_packed_weight_floats(co, c, 3, 3, 0), device=DEV)
    d3 = _lib.ConvDesc(b, h, w, c, 0, 0, co, h, w, 3, 3, 1, 1, 0, 0, 1, 1, 0, 0)
    rc = lib.codeps_conv2d_fwd_in(ctypes.byref(d3), _lib.ptr(x), _lib.ptr(tab), _lib.ptr(wf), None, None, None, _lib.ptr(y), None, 0, None, _lib.stream_ptr())
    assert rc != 0 and b"in_affine_ok" in lib.codeps_last_error()
    dw = torch.zeros(co, c, 3, 3, device=DEV)
    ws = torch.zeros(1 << 20, device=DEV)
    rc = lib.codeps_conv2d_wgrad_in(ctypes.byref(d3), _lib.ptr(x), _lib.ptr(tab), _lib.ptr(y), _lib.ptr(dw), co, c, 0, _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr())
    assert rc != 0 and b"in_affine_ok" in lib.codeps_last_error()
    d1 = _lib.ConvDesc(b, h, w, c, 0, 0, co, h, w, 1, 1, 1, 0, 0, 0, 1, 0, 0, 0)          # 256 pixels per image: served
    assert lib.codeps_conv2d_in_affine_ok(ctypes.byref(d1), 0) == 1
    d1s = _lib.ConvDesc(b, 2 * h, 2 * w, c, 0, 0, co, h, w, 1, 1, 2, 0, 0, 0, 2, 0, 0, 0)  # stride 2 (a down-sampling shortcut): not served
    assert lib.codeps_conv2d_in_affine_ok(ctypes.byref(d1s), 0) == 0 and lib.codeps_conv2d_in_affine_ok(ctypes.byref(d1s), 1) == 0


def test_a_deferred_tensor_refuses_a_fork():
    from codeps_amd import functional as F
    from codeps_amd._lib import CodepsHipError
    z = torch.randn(2, 16, 32, 64, device=DEV)
    w = torch.randn(64, 64, 1, 1, device=DEV)
    ones, zeros = torch.ones(64, device=DEV), torch.zeros(64, device=DEV)
    y = F.batch_norm_train(z, ones, zeros, zeros.clone(), ones.clone(), act="relu", defer=True)
    with pytest.raises(CodepsHipError):
        F.conv2d(y, w, fork=True)


@pytest.mark.parametrize("arith", [16, 3])
def test_r50_bottleneck_stage_and_instance_decoder_with_and_without_deferral(arith):
    """A trainable ResNet-50 layer1 + the instance head's depthwise stages, forward + backward, CODEPS_BN_DEFER on / off: identical."""
    from codeps_amd import functional as F
    from codeps_amd._lib import lib
    from codeps_amd.models import InstanceHead, ResnetEncoder
    prev_arith = lib().codeps_conv2d_set_arith(-1)
    lib().codeps_conv2d_set_arith(arith)
    F.invalidate_weight_cache(True)
    try:
        res = []
        for on in (True, False):
            torch.manual_seed(4)
            enc = ResnetEncoder(50, False).to(DEV).train()
            head = InstanceHead(enc.num_ch_enc).to(DEV).train()
            x = torch.randn(2, 3, 128, 256, device=DEV)
            F._DEFER = on
            feats = enc(x)
            c, o, _t = head(feats)
            (c.square().mean() + o.square().mean() + sum(f.mean() for f in feats)).backward()
            torch.cuda.synchronize()
            res.append(([f.detach() for f in feats] + [c.detach(), o.detach()],
                        [p.grad.clone() for p in list(enc.parameters()) + list(head.parameters()) if p.grad is not None],
                        [m.running_var.clone() for m in list(enc.modules()) + list(head.modules()) if isinstance(m, torch.nn.BatchNorm2d)]))
        for part_on, part_off in zip(res[0], res[1]):
            assert len(part_on) == len(part_off)
            for a, b in zip(part_on, part_off):
                assert torch.equal(a, b), float((a - b).abs().max())
    finally:
        F._DEFER = True
        lib().codeps_conv2d_set_arith(prev_arith)
        F.invalidate_weight_cache(True)
