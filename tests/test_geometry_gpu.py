"""The reference's public geometry API on the HIP path: ImageWarper / CoordinateWarper (misc/image_warper.py:92-184,
csrc/warp.hip) and SSIMLoss (algos/depth.py:110-155, csrc/ssim.hip) against outputs AND gradients of the real reference
(tests/golden/geometry.npz, geometry_grad.npz)."""
import pytest
import torch

import inputs as gi
from helpers import close, load

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cams(intr, w, h, as_tensor):
    from codeps_amd.misc import CameraModel
    return intr.to(DEV) if as_tensor else [CameraModel.from_tensor(w, h, intr[i].to(DEV)) for i in range(intr.shape[0])]


@pytest.mark.parametrize("as_tensor", [False, True])
def test_warpers_forward_against_reference(as_tensor):
    from codeps_amd.misc import ImageWarper
    from codeps_amd.misc.image_warper import CoordinateWarper
    z, meta = load("geometry")
    b, h, w, seed = meta["b"], meta["h"], meta["w"], meta["seed"]
    cm = _cams(gi.intrinsics(b, h, w, seed), w, h, as_tensor)
    depth, T, src = gi.depth_map(b, h, w, seed).to(DEV), gi.pose(b, seed).to(DEV), gi.smooth_image(b, h, w, seed + 1).to(DEV)
    grid = CoordinateWarper(w, h, DEV)(cm, depth, T)
    close(grid, z["grid"], 2e-6, 2e-6, "grid")                 # normalised coordinates of O(1); a few ulp of the projection
    close(ImageWarper(w, h, DEV)(cm, src, depth, T), z["warped"], 2e-5, 0, "warped")
    eye = torch.eye(4, device=DEV).repeat(b, 1, 1)
    ident = ImageWarper(w, h, DEV)(cm, src, depth, eye)
    close(ident, z["warped_identity"], 2e-5, 0, "identity warp")
    close(ident, src, 2e-4, 0, "identity warp reproduces the source")


def _leaves(b, h, w, seed):
    return (gi.smooth_image(b, h, w, seed + 1).to(DEV).requires_grad_(True), gi.depth_map(b, h, w, seed).to(DEV).requires_grad_(True),
            gi.pose(b, seed, rot=0.05, trans=0.6).to(DEV).requires_grad_(True))


def _rel(a, ref, what, tol):
    ref = torch.as_tensor(ref).to(a.device)
    err = float((a.double() - ref.double()).norm() / (ref.double().norm() + 1e-30))
    assert err < tol, f"{what}: relative L2 error {err:.3e} >= {tol}"


def test_warpers_are_differentiable_like_the_reference():
    from codeps_amd.misc import ImageWarper
    from codeps_amd.misc.image_warper import CoordinateWarper
    z, meta = load("geometry_grad")
    b, h, w, seed = meta["b"], meta["h"], meta["w"], meta["seed"]
    cm = _cams(gi.intrinsics(b, h, w, seed), w, h, False)
    src, depth, T = _leaves(b, h, w, seed)
    out = ImageWarper(w, h, DEV)(cm, src, depth, T)
    close(out, z["warped"], 2e-5, 0, "warped")
    (out * gi.normal(tuple(out.shape), seed + 5).to(DEV)).sum().backward()
    # depth / pose gradients go through image differences over one pixel: a pixel whose sample point sits within rounding of
    # an integer coordinate may pick the neighbouring cell in either implementation -> compare in relative L2
    _rel(src.grad, z["warped_dsrc"], "d warped / d src", 1e-5)
    _rel(depth.grad, z["warped_ddepth"], "d warped / d depth", 2e-4)
    _rel(T.grad, z["warped_dT"], "d warped / d T", 2e-4)

    src, depth, T = _leaves(b, h, w, seed)
    grid = CoordinateWarper(w, h, DEV)(cm, depth, T)
    close(grid, z["grid"], 2e-6, 2e-6, "grid")
    (grid * gi.normal(tuple(grid.shape), seed + 6).to(DEV)).sum().backward()
    _rel(depth.grad, z["grid_ddepth"], "d grid / d depth", 1e-5)
    _rel(T.grad, z["grid_dT"], "d grid / d T", 1e-5)

    src, depth, T = _leaves(b, h, w, seed)
    out = ImageWarper(w, h, DEV)(cm, src, depth, T, interp_mode="nearest")
    bad = (out.detach().cpu() != torch.as_tensor(z["nearest"])).float().mean()
    assert bad < 2e-3, f"nearest warp: {float(bad):.2%} of the pixels picked another neighbour"
    (out * gi.normal(tuple(out.shape), seed + 5).to(DEV)).sum().backward()
    _rel(src.grad, z["nearest_dsrc"], "d nearest / d src", 5e-2 if bad > 0 else 1e-6)
    assert depth.grad is None or float(depth.grad.abs().max()) == 0.0

    src, depth, T = _leaves(b, h, w, seed)
    motion = gi.normal((b, 3, h, w), seed + 7, 0.05).to(DEV).requires_grad_(True)
    out = ImageWarper(w, h, DEV)(cm, src, depth, T, object_motion_map=motion)
    close(out, z["motion_warped"], 2e-5, 0, "warped with object motion")
    (out * gi.normal(tuple(out.shape), seed + 5).to(DEV)).sum().backward()
    _rel(motion.grad, z["motion_dmotion"], "d warped / d motion", 2e-4)
    _rel(depth.grad, z["motion_ddepth"], "d warped / d depth (motion)", 2e-4)
    _rel(T.grad, z["motion_dT"], "d warped / d T (motion)", 2e-4)


def test_ssim_loss_is_callable_and_differentiable():
    from codeps_amd.algos import SSIMLoss
    z, meta = load("geometry")
    b, h, w, seed = meta["b"], meta["h"], meta["w"], meta["seed"]
    a, c = gi.smooth_image(b, h, w, seed + 2).to(DEV), gi.smooth_image(b, h, w, seed + 3).to(DEV)
    close(SSIMLoss()(a, c), z["ssim"], 2e-6, 1e-5, "ssim")
    close(SSIMLoss()(a, a), z["ssim_self"], 2e-6, 0, "ssim(a, a)")
    zg, mg = load("geometry_grad")
    seed = mg["seed"]
    a = gi.smooth_image(b, h, w, seed + 2).to(DEV).requires_grad_(True)
    c = gi.smooth_image(b, h, w, seed + 3).to(DEV).requires_grad_(True)
    sm = SSIMLoss()(a, c)
    close(sm, zg["ssim"], 2e-6, 1e-5, "ssim (grad case)")
    (sm * gi.normal(tuple(sm.shape), seed + 8).to(DEV)).sum().backward()
    _rel(a.grad, zg["ssim_da"], "d ssim / d src", 2e-5)
    _rel(c.grad, zg["ssim_dc"], "d ssim / d target", 2e-5)
    with pytest.raises(ValueError):
        SSIMLoss(5)


def test_ssim_full_size_against_torch_composite():
    """512x1024: the kernel against the reference's own expression written with torch ops on the device."""
    import torch.nn.functional as tf
    from codeps_amd.algos import SSIMLoss
    gen = torch.Generator(device=DEV).manual_seed(5)
    x = torch.rand(2, 3, 512, 1024, generator=gen, device=DEV, requires_grad=True)
    y = (x.detach() + 0.1 * torch.randn(x.shape, generator=gen, device=DEV)).requires_grad_(True)
    g = torch.randn(x.shape, generator=gen, device=DEV)

    def ref(a, b):
        a, b = tf.pad(a, (1, 1, 1, 1), mode="reflect"), tf.pad(b, (1, 1, 1, 1), mode="reflect")
        pool = lambda t: tf.avg_pool2d(t, 3, 1)                        # noqa: E731
        mx, my = pool(a), pool(b)
        sx, sy, sxy = pool(a * a) - mx * mx, pool(b * b) - my * my, pool(a * b) - mx * my
        n = (2 * mx * my + 1e-4) * (2 * sxy + 9e-4)
        d = (mx * mx + my * my + 1e-4) * (sx + sy + 9e-4)
        return torch.clamp((1 - n / d) / 2, 0, 1)

    want = ref(x, y)
    gx, gy = torch.autograd.grad(want, (x, y), g)
    got = SSIMLoss()(x, y)
    hx, hy = torch.autograd.grad(got, (x, y), g)
    close(got, want, 5e-6, 1e-5, "ssim 512x1024")
    _rel(hx, gx, "d ssim / d x", 5e-5)
    _rel(hy, gy, "d ssim / d y", 5e-5)
