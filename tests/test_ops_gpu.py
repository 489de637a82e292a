"""HIP NHWC ops (conv engine, BN, pooling, resize, dwconv, softmax, disp->depth) against plain PyTorch
CPU references (float64 where cheap).  GPU box only."""
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _F():
    from codeps_amd import functional
    return functional


def nhwc(x, cp=None):
    """CPU NCHW -> device NHWC padded."""
    b, c, h, w = x.shape
    cp = cp or (c + 15) // 16 * 16
    y = torch.zeros(b, h, w, cp)
    y[..., :c] = x.permute(0, 2, 3, 1)
    return y.to(DEV)


def nchw(y, c):
    return y[..., :c].permute(0, 3, 1, 2).cpu()


def rel_err(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


CONV_CASES = [
    # (B, H, W, C0, C1, up0, Cout, k, stride, pad, pad_mode, act, bias, affine, residual)
    (2, 17, 23, 32, 0, 0, 64, 3, 1, 1, "zeros", "elu", True, False, False),
    (1, 16, 24, 64, 0, 0, 256, 1, 1, 0, "zeros", "relu", False, True, True),
    (2, 19, 31, 64, 0, 0, 128, 3, 2, 1, "zeros", None, False, False, False),
    (1, 32, 48, 3, 0, 0, 64, 7, 2, 3, "zeros", "relu", False, True, False),      # stem (3 -> pad 16)
    (2, 16, 24, 32, 64, 1, 32, 3, 1, 1, "zeros", "elu", True, False, False),     # up(x0) ++ skip
    (1, 12, 20, 256, 1024, 1, 256, 3, 1, 1, "zeros", "elu", True, False, False),
    (2, 15, 22, 16, 0, 0, 1, 3, 1, 1, "reflect", None, True, False, False),      # dispconv
    (1, 14, 18, 240, 0, 0, 14, 3, 1, 1, "zeros", None, True, False, False),      # GUDA conv
    (3, 9, 11, 512, 0, 0, 256, 3, 1, 1, "zeros", "relu", True, False, False),    # pose head
    (2, 20, 28, 128, 0, 0, 16, 3, 1, 1, "zeros", "elu", True, False, False),
    (1, 8, 8, 16, 0, 0, 16, 3, 1, 1, "zeros", "elu", True, False, False),
    (2, 10, 14, 64, 0, 0, 128, 1, 2, 0, "zeros", None, False, True, False),      # downsample 1x1/2
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_forward_backward(case):
    run_conv_case(case, 1000 + CONV_CASES.index(case))


def run_conv_case(case, seed):
    """One fused convolution (forward, data / weight / bias / residual gradients) against float64 on the CPU."""
    F = _F()
    b, h, w, c0, c1, up0, co, k, stride, pad, pad_mode, act, bias, affine, residual = case
    g = torch.Generator().manual_seed(seed)
    hs, ws_ = (h // 2, w // 2) if up0 else (h, w)
    if up0:
        h, w = hs * 2, ws_ * 2
    x0 = torch.randn(b, c0, hs, ws_, generator=g)
    x1 = torch.randn(b, c1, h, w, generator=g) if c1 else None
    wt = torch.randn(co, c0 + c1, k, k, generator=g) / (k * (c0 + c1) ** 0.5)
    bs = torch.randn(co, generator=g) * 0.1 if bias else None
    sc = torch.rand(co, generator=g) + 0.5 if affine else None
    sh = torch.randn(co, generator=g) * 0.1 if affine else None
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = torch.randn(b, co, ho, wo, generator=g) if residual else None

    # ---- reference (float64 CPU)
    rx0, rw = x0.double().requires_grad_(True), wt.double().requires_grad_(True)
    rx1 = x1.double().requires_grad_(True) if c1 else None
    rb = bs.double().requires_grad_(True) if bias else None
    rres = res.double().requires_grad_(True) if residual else None
    xin = TF.interpolate(rx0, scale_factor=2, mode="nearest") if up0 else rx0
    if c1:
        xin = torch.cat([xin, rx1], 1)
    if pad_mode == "reflect":
        y = TF.conv2d(TF.pad(xin, (pad,) * 4, mode="reflect"), rw, rb, stride)
    else:
        y = TF.conv2d(xin, rw, rb, stride, pad)
    if affine:
        y = y * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    if residual:
        y = y + rres
    pre = y.detach()
    y = TF.elu(y) if act == "elu" else (TF.relu(y) if act == "relu" else y)
    gy = torch.randn(y.shape, generator=g).double()
    if act == "relu":
        # an fp32 pre-activation within rounding of zero may land on the other side of the kink than the float64 one: one
        # flipped mask element moves a weight-gradient entry by 1/sqrt(pixels) of its size.  Such outputs get no gradient.
        gy = gy * (pre.abs() > 1e-5)
    (y * gy).sum().backward()

    # ---- HIP
    dx0 = nhwc(x0).requires_grad_(c0 % 16 == 0 and pad_mode == "zeros" and co % 16 == 0)
    dx1 = nhwc(x1).requires_grad_(True) if c1 else None
    dw = wt.to(DEV).requires_grad_(True)
    db = bs.to(DEV).requires_grad_(True) if bias else None
    dres = nhwc(res).requires_grad_(True) if residual else None
    out = F.conv2d(dx0, dw, db, x1=dx1, scale=None if sc is None else sc.to(DEV), shift=None if sh is None else sh.to(DEV),
                   residual=dres, stride=stride, pad=pad, pad_mode=pad_mode, act=act, up0=bool(up0))
    assert out.shape == (b, ho, wo, (co + 15) // 16 * 16)
    assert rel_err(nchw(out, co), y.detach()) < 2e-5
    if co % 16:
        assert out[..., co:].abs().max().item() == 0.0          # padding channels stay zero
    out.backward(nhwc(gy.float(), out.shape[-1]))
    assert rel_err(dw.grad.cpu(), rw.grad) < 5e-5
    if bias:
        assert rel_err(db.grad.cpu(), rb.grad) < 5e-5
    if dx0.requires_grad:
        assert rel_err(nchw(dx0.grad, c0), rx0.grad) < 5e-5
    if c1:
        assert rel_err(nchw(dx1.grad, c1), rx1.grad) < 5e-5
    if residual:
        assert rel_err(nchw(dres.grad, co), rres.grad) < 5e-5


@pytest.mark.parametrize("c,act,res", [(64, "relu", True), (256, None, False), (2048, "relu", False), (48, "relu", True)])
def test_batch_norm_train(c, act, res):
    run_bn_case(c, act, res, 3, 9, 13, c)


def run_bn_case(c, act, res, b, h, w, seed):
    F = _F()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, c, h, w, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1
    r = torch.randn(b, c, h, w, generator=g) if res else None
    rm, rv = torch.zeros(c), torch.ones(c)
    rx, rg, rb = x.double().requires_grad_(True), gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rr = r.double().requires_grad_(True) if res else None
    rrm, rrv = rm.double(), rv.double()
    y = TF.batch_norm(rx, rrm, rrv, rg, rb, True, 0.1, 1e-5)
    if res:
        y = y + rr
    if act:
        y = TF.relu(y)
    gy = torch.randn(y.shape, generator=g).double()
    if act:
        gy = gy * (y.detach().abs() > 1e-5)                 # outputs on the ReLU kink take no gradient (see run_conv_case)
    (y * gy).sum().backward()
    dx, dg, dbt = nhwc(x).requires_grad_(True), gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    dr = nhwc(r).requires_grad_(True) if res else None
    drm, drv = rm.to(DEV), rv.to(DEV)
    out = F.batch_norm_train(dx, dg, dbt, drm, drv, 0.1, 1e-5, residual=dr, act=act)
    assert rel_err(nchw(out, c), y.detach()) < 2e-5
    out.backward(nhwc(gy.float()))
    assert rel_err(nchw(dx.grad, c), rx.grad) < 2e-4
    assert rel_err(dg.grad.cpu(), rg.grad) < 1e-4 and rel_err(dbt.grad.cpu(), rb.grad) < 1e-4
    assert rel_err(drm.cpu(), rrm) < 1e-5 and rel_err(drv.cpu(), rrv) < 1e-5
    if res:
        assert rel_err(nchw(dr.grad, c), rr.grad) < 1e-5


def test_maxpool():
    run_maxpool_case(2, 64, 17, 22, 3)


def run_maxpool_case(b, c, h, w, seed):
    F = _F()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, c, h, w, generator=g)
    rx = x.double().requires_grad_(True)
    y = TF.max_pool2d(rx, 3, 2, 1)
    gy = torch.randn(y.shape, generator=g).double()
    (y * gy).sum().backward()
    dx = nhwc(x).requires_grad_(True)
    out = F.max_pool3x3s2(dx)
    assert torch.equal(nchw(out, c), y.detach().float())
    out.backward(nhwc(gy.float()))
    assert rel_err(nchw(dx.grad, c), rx.grad) < 1e-6


@pytest.mark.parametrize("align,h,w,oh,ow", [(False, 8, 12, 64, 96), (False, 16, 24, 32, 48), (True, 5, 7, 10, 14),
                                             (True, 4, 8, 8, 16), (True, 6, 9, 11, 20)])
def test_bilinear(align, h, w, oh, ow):
    run_bilinear_case(align, 2, 32, h, w, oh, ow, h)


def run_bilinear_case(align, b, c, h, w, oh, ow, seed):
    F = _F()
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, c, h, w, generator=g)
    rx = x.double().requires_grad_(True)
    y = TF.interpolate(rx, (oh, ow), mode="bilinear", align_corners=align)
    gy = torch.randn(y.shape, generator=g).double()
    (y * gy).sum().backward()
    dx = nhwc(x).requires_grad_(True)
    out = F.bilinear_resize(dx, oh, ow, align)
    assert rel_err(nchw(out, c), y.detach()) < 1e-5
    out.backward(nhwc(gy.float()))
    assert rel_err(nchw(dx.grad, c), rx.grad) < 1e-5


def test_bilinear_concat_guda():
    F = _F()
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(1, c, 32 // s, 48 // s, generator=g) for c, s in ((128, 8), (64, 4), (32, 2), (16, 1))]
    rxs = [x.double().requires_grad_(True) for x in xs]
    y = torch.cat([TF.interpolate(r, (32, 48), mode="bilinear", align_corners=False) for r in rxs], 1)
    gy = torch.randn(y.shape, generator=g).double()
    (y * gy).sum().backward()
    dxs = [nhwc(x).requires_grad_(True) for x in xs]
    out = F.bilinear_concat(dxs, 32, 48, False)
    assert rel_err(nchw(out, 240), y.detach()) < 1e-5
    out.backward(nhwc(gy.float(), 240))
    for d, r, x in zip(dxs, rxs, xs):
        assert rel_err(nchw(d.grad, x.shape[1]), r.grad) < 1e-5


@pytest.mark.parametrize("ci,h,w,b", [(3, 32, 64, 2), (6, 38, 52, 3)])
def test_stem_conv_double_pixel_layout(ci, h, w, b):
    """ResNet conv1 (7x7 / 2 / 3) through the double-pixel stem layout (nchw_to_nhwc_xpad + 7x4 conv of stride (2,1))."""
    F = _F()
    g = torch.Generator().manual_seed(ci * 7 + h)
    x = torch.randn(b, ci, h, w, generator=g)
    wt = torch.randn(64, ci, 7, 7, generator=g) * 0.05
    rw = wt.double().requires_grad_(True)
    y = TF.conv2d(x.double(), rw, None, 2, 3)
    gy = torch.randn(y.shape, generator=g).double()
    (y * gy).sum().backward()
    halves = [x[:, :3].to(DEV)] + ([x[:, 3:].to(DEV)] if ci == 6 else [])
    xs = F.images_to_nhwc(halves, 8, F.STEM_XPAD)
    assert xs.shape == (b, h, w + 6, 8)
    assert float(xs[:, :, :4].abs().max()) == 0.0 and float(xs[:, :, -2:].abs().max()) == 0.0 and float(xs[..., ci:].abs().max()) == 0.0
    dw = wt.to(DEV).requires_grad_(True)
    out = F.stem_conv2d(xs, dw)
    assert rel_err(nchw(out, 64), y) < 1e-5
    out.backward(nhwc(gy.float()))
    assert rel_err(dw.grad.cpu(), rw.grad) < 1e-5


@pytest.mark.parametrize("h,w,b", [(32, 48, 2), (24, 40, 1), (37, 150, 1), (16, 200, 2)])      # the last two: several 8 x 64 tiles, ragged edges, non-integer ratios
def test_guda_fusion_matches_concat_conv(h, w, b):
    """conv3x3(cat(up(d_k))) evaluated as per-tap 1x1 convs at low resolution + tap gather (semantic_head.py:64-72)."""
    F = _F()
    g = torch.Generator().manual_seed(h)
    xs = [torch.randn(b, c, h // s, w // s, generator=g) for c, s in ((128, 8), (64, 4), (32, 2), (16, 1))]
    wt, bias = torch.randn(14, 240, 3, 3, generator=g) * 0.05, torch.randn(14, generator=g)
    rxs = [x.double().requires_grad_(True) for x in xs]
    rw, rb = wt.double().requires_grad_(True), bias.double().requires_grad_(True)
    cat = torch.cat([TF.interpolate(r, (h, w), mode="bilinear", align_corners=False) for r in rxs[:3]] + [rxs[3]], 1)
    y = TF.conv2d(cat, rw, rb, padding=1)
    gy = torch.randn(y.shape, generator=g).double()
    (y * gy).sum().backward()
    dxs = [nhwc(x).requires_grad_(True) for x in xs]
    dw, db = wt.to(DEV).requires_grad_(True), bias.to(DEV).requires_grad_(True)
    out = F.guda_fusion(dxs[:3], dxs[3], dw, db)
    assert rel_err(nchw(out, 14), y.detach()) < 1e-5
    assert float(out[..., 14:].abs().max()) == 0.0
    out.backward(nhwc(gy.float(), 16))
    for d, r, x in zip(dxs, rxs, xs):
        assert rel_err(nchw(d.grad, x.shape[1]), r.grad) < 1e-5
    assert rel_err(dw.grad.cpu(), rw.grad) < 1e-5
    assert rel_err(db.grad.cpu(), rb.grad) < 1e-5


@pytest.mark.parametrize("b,c,h,w", [(2, 144, 13, 17), (2, 128, 40, 70), (1, 320, 33, 16), (3, 32, 16, 16), (1, 36, 48, 50), (5, 64, 5, 37)])
def test_dwconv5x5(b, c, h, w):
    """Depthwise 5x5 (models/instance_head.py:38-46) on the LDS-tile kernels: one tile, several tiles with ragged edges, channel counts that
    end inside a 32-channel block, more tiles than one block of the weight gradient takes."""
    F = _F()
    g = torch.Generator().manual_seed(9 + c + h)
    x = torch.randn(b, c, h, w, generator=g)
    wt = torch.randn(c, 1, 5, 5, generator=g) * 0.2
    sc, sh = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1
    rx, rw = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    y = TF.relu(TF.conv2d(rx, rw, None, 1, 2, 1, c) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    gy = torch.randn(y.shape, generator=g).double()
    (y * gy).sum().backward()
    dx, dw = nhwc(x).requires_grad_(True), wt.to(DEV).requires_grad_(True)
    out = F.dwconv5x5(dx, dw, sc.to(DEV), sh.to(DEV), "relu")
    assert rel_err(nchw(out, c), y.detach()) < 1e-5
    out.backward(nhwc(gy.float()))
    assert rel_err(nchw(dx.grad, c), rx.grad) < 1e-5
    assert rel_err(dw.grad.cpu(), rw.grad) < 1e-4


def test_softmax_and_disp_depth():
    F = _F()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 14, 9, 12, generator=g) * 3
    rx = x.double().requires_grad_(True)
    y = torch.softmax(rx, 1)
    gy = torch.randn(y.shape, generator=g).double()
    (y * gy).sum().backward()
    dx = nhwc(x).requires_grad_(True)
    out = F.channel_softmax(dx, 14)
    assert rel_err(nchw(out, 14), y.detach()) < 1e-6
    out.backward(nhwc(gy.float()))
    assert rel_err(nchw(dx.grad, 14), rx.grad) < 1e-5
    # sigmoid + disp_to_depth
    lg = torch.randn(2, 1, 9, 12, generator=g) * 3
    rl = lg.double().requires_grad_(True)
    disp = torch.sigmoid(rl)
    depth = 1 / (0.01 + (10 - 0.01) * disp)
    g1, g2 = torch.randn(depth.shape, generator=g).double(), torch.randn(depth.shape, generator=g).double()
    ((depth * g1).sum() + (disp * g2).sum()).backward()
    dl = nhwc(lg).requires_grad_(True)
    d_depth, d_disp = F.disp_depth(dl)
    assert rel_err(d_depth.cpu(), depth.detach()) < 1e-5 and rel_err(d_disp.cpu(), disp.detach()) < 1e-6
    ((d_depth * g1.float().to(DEV)).sum() + (d_disp * g2.float().to(DEV)).sum()).backward()
    assert rel_err(dl.grad[..., 0].cpu(), rl.grad[:, 0]) < 1e-4


def test_layout_roundtrip_zero_copy():
    F = _F()
    x = torch.randn(2, 3, 8, 10)
    y = F.from_nchw(x.to(DEV))
    assert y.shape == (2, 8, 10, 16) and torch.equal(y[..., :3].cpu(), x.permute(0, 2, 3, 1)) and y[..., 3:].abs().max() == 0
    v = F.to_nchw(y, 3)
    assert v.shape == (2, 3, 8, 10) and torch.equal(v.cpu(), x)
    assert F.from_nchw(v).data_ptr() == y.data_ptr()            # zero-copy round trip
    two = F.images_to_nhwc([x.to(DEV), (x + 1).to(DEV)])
    assert torch.equal(two[..., 3:6].cpu(), (x + 1).permute(0, 2, 3, 1)) and two[..., 6:].abs().max() == 0


def test_sync_bn_combine_matches_parallel_variance():
    """codeps_bn_sync_combine (the merge after the SyncBatchNorm all-gather) against the parallel-variance formula, and
    the whole SyncBN forward/backward with a 3-rank reducer simulated on one GPU (every rank holds a different shard)."""
    import ctypes
    F = _F()
    from codeps_amd._lib import lib, ptr, stream_ptr, check
    g = torch.Generator().manual_seed(3)
    world, ng, c = 3, 2, 32
    shards = [torch.randn(4, 6, 5, c, generator=g) * (1 + r) + r for r in range(world)]          # NHWC, groups = batch halves
    groups = [(0, 2), (2, 4)]
    per = 2 * 6 * 5
    # reference: statistics of the concatenation of all ranks' shards, per group
    ref_mean, ref_var = [], []
    for (b0, b1) in groups:
        allx = torch.cat([s_[b0:b1].reshape(-1, c) for s_ in shards]).double()
        ref_mean.append(allx.mean(0)); ref_var.append(allx.var(0, unbiased=False))
    ref_mean, ref_var = torch.stack(ref_mean), torch.stack(ref_var)
    packets = []
    for s_ in shards:
        m = torch.stack([s_[b0:b1].reshape(-1, c).double().mean(0) for b0, b1 in groups])
        v = torch.stack([s_[b0:b1].reshape(-1, c).double().var(0, unbiased=False) for b0, b1 in groups])
        packets.append(torch.stack([m, v]).float())
    gathered = torch.stack(packets).to(DEV)                                                      # [world, 2, ng, c]
    mean, var, invstd = (torch.empty(ng, c, device=DEV) for _ in range(3))
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    counts = (ctypes.c_float * ng)(*[float(per * world)] * ng)
    check(lib().codeps_bn_sync_combine(ptr(gathered), world, ng, counts, None, c, c, 1e-5, 0.1, ptr(rm), ptr(rv), ptr(mean), ptr(var), ptr(invstd),
                                       stream_ptr()), "codeps_bn_sync_combine")
    assert rel_err(mean.cpu(), ref_mean) < 1e-5 and rel_err(var.cpu(), ref_var) < 1e-5
    # unequal shards (BASELINE configs[3]): rank r holds r + 1 items of group 0 and (2, 0, 1) items of group 1
    items = [[1, 2], [2, 0], [3, 1]]
    u_shards = [[torch.randn(items[r][gi] * 7, c, generator=g) * (1 + r) + gi for gi in range(ng)] for r in range(world)]
    pk = torch.zeros(world, 2, ng, c)
    for r in range(world):
        for gi in range(ng):
            if items[r][gi]:
                pk[r, 0, gi], pk[r, 1, gi] = u_shards[r][gi].double().mean(0).float(), u_shards[r][gi].double().var(0, unbiased=False).float()
            else:
                pk[r, :, gi] = float("nan")                  # an empty slot must be ignored, whatever it holds
    allg = [torch.cat([u_shards[r][gi] for r in range(world)]).double() for gi in range(ng)]
    rc = (ctypes.c_float * (world * ng))(*[float(items[r][gi] * 7) for r in range(world) for gi in range(ng)])
    gc = (ctypes.c_float * ng)(*[float(a.shape[0]) for a in allg])
    m2, v2, i2 = (torch.empty(ng, c, device=DEV) for _ in range(3))
    check(lib().codeps_bn_sync_combine(ptr(pk.to(DEV)), world, ng, gc, rc, c, c, 1e-5, 0.1, None, None, ptr(m2), ptr(v2), ptr(i2), stream_ptr()),
          "codeps_bn_sync_combine")
    assert rel_err(m2.cpu(), torch.stack([a.mean(0) for a in allg])) < 1e-5
    assert rel_err(v2.cpu(), torch.stack([a.var(0, unbiased=False) for a in allg])) < 1e-5
    assert rel_err(invstd.cpu(), 1.0 / torch.sqrt(ref_var + 1e-5)) < 1e-5
    erm, erv = torch.zeros(c).double(), torch.ones(c).double()
    n = per * world
    for gi in range(ng):
        erm = erm * 0.9 + 0.1 * ref_mean[gi]
        erv = erv * 0.9 + 0.1 * ref_var[gi] * n / (n - 1)
    assert rel_err(rm.cpu(), erm) < 1e-5 and rel_err(rv.cpu(), erv) < 1e-5

    # end to end: rank 0's SyncBN output / input gradient with a reducer that "gathers" the three shards' statistics
    class FakeReducer:
        def gather(self, pk):
            return gathered
        def sum(self, sums):                      # backward: sum over ranks of (sum g, sum g xhat); ranks 1, 2 contribute zero here
            return sums
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    x = shards[0].to(DEV).requires_grad_(True)
    y = F.batch_norm_train(x, gamma.to(DEV), beta.to(DEV), None, None, 0.1, 1e-5, None, "relu", FakeReducer(), groups)
    want = []
    for gi, (b0, b1) in enumerate(groups):
        xh = (shards[0][b0:b1].double() - ref_mean[gi]) / torch.sqrt(ref_var[gi] + 1e-5)
        want.append(torch.relu(xh * gamma.double() + beta.double()))
    assert rel_err(y.detach().cpu(), torch.cat(want)) < 1e-5


def test_conv_error_is_at_fp32_rounding_level():
    """The default arithmetic of the main conv kernel (fp32 operands as three bf16 pieces, six products on the bf16 MFMA, fp32
    accumulate) must be indistinguishable from an fp32 convolution: its error against float64 stays within 3x the error of
    torch's own fp32 CPU convolution of the same tensors (measured: 2.5e-7 vs 1.7e-7 relative L2 forward; the f32 MFMA path
    gives 2.2e-7, a 2-piece split 4.5e-6 = 26x)."""
    F = _F()
    g = torch.Generator().manual_seed(77)
    x = torch.relu(torch.randn(2, 256, 24, 40, generator=g))
    wt = torch.randn(256, 256, 3, 3, generator=g) * 0.02
    gy = torch.randn(2, 256, 24, 40, generator=g)
    l2 = lambda a, b: float((a.double() - b).norm() / b.norm())
    r64 = x.double().requires_grad_(True)
    ref = TF.conv2d(r64, wt.double(), None, 1, 1)
    (ref * gy.double()).sum().backward()
    r32 = x.clone().requires_grad_(True)
    y32 = TF.conv2d(r32, wt, None, 1, 1)
    (y32 * gy).sum().backward()
    dx = nhwc(x).requires_grad_(True)
    y = F.conv2d(dx, wt.to(DEV), pad=1)
    y.backward(nhwc(gy))
    assert l2(nchw(y, 256), ref.detach()) < 3 * l2(y32.detach(), ref.detach())
    # data gradient: a sequential fp32 chain of 2304 random-sign terms (the f32 MFMA path measures the same 3.3x)
    assert l2(nchw(dx.grad, 256), r64.grad) < 5 * l2(r32.grad, r64.grad)
