"""The halo-tile ring kernels (weights by LDS DMA, hand-counted vmcnt waits, fragment reads under the MFMAs) -- conv_halo3p_kernel
(default: four consumer waves + two producer waves) and conv_halo3r_kernel (every wave stages and multiplies; CODEPS_HALO_WS=0 /
codeps_conv2d_set_halo_ws(0)) -- against the register-staged kernel they replace: the products of every accumulator are issued in the
same order, so all three must agree BIT FOR BIT, and a missed wait would show up as a launch-to-launch difference -- 300 launches per
shape must return the same bits."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHAPES = {                       # name: (B, H, W, C0, C1, Cout, up0)
    "layer3": (8, 32, 64, 256, 0, 256, False),
    "decoder": (4, 64, 128, 128, 128, 256, True),
    "wide": (2, 64, 128, 128, 0, 512, False),
    "two_chunks_edge": (8, 36, 96, 32, 32, 128, False),        # both concat sources, tiles on every image border
    "deep": (2, 32, 64, 1024, 256, 256, True),                 # 40 chunks, source switch in the middle, fused nearest x2
}

_CHILD = r"""
import sys, torch
sys.path.insert(0, {root!r})
sys.path.insert(0, {tests!r})
from test_halo_ring_gpu import run_case
torch.save({{k: run_case(k, 1)[0].cpu() for k in {names!r}}}, {out!r})
"""


def run_case(name, launches):
    from codeps_amd import functional as F, _lib
    import ctypes
    lib = _lib.lib()
    lib.codeps_conv2d_last_kernel.restype = ctypes.c_char_p
    b, h, w, c0, c1, co, up = SHAPES[name]
    g = torch.Generator().manual_seed(1234 + len(name))
    x0 = torch.randn(b, h // 2 if up else h, w // 2 if up else w, c0, generator=g).to(DEV)
    x1 = torch.randn(b, h, w, c1, generator=g).to(DEV) if c1 else None
    wt = (torch.randn(co, c0 + c1, 3, 3, generator=g) * 0.05).to(DEV)
    bias = torch.randn(co, generator=g).to(DEV)
    outs = []
    for _ in range(launches):
        outs.append(F.conv2d(x0, wt, bias=bias, x1=x1, pad=1, up0=up, act="elu"))
    torch.cuda.synchronize()
    return outs[0], outs, lib.codeps_conv2d_last_kernel().decode()


def _set_ws(on):
    from codeps_amd import _lib
    return _lib.lib().codeps_conv2d_set_halo_ws(on)


def _set_arith(a):
    from codeps_amd import _lib
    return _lib.lib().codeps_conv2d_set_arith(a)


def _family(name):
    from helpers import kernel_family
    return kernel_family(name)


@pytest.mark.parametrize("ws", [1, 0])
@pytest.mark.parametrize("name", list(SHAPES))
def test_ring_kernel_is_repeatable(name, ws):
    # the producer / consumer form exists on the three-bf16-piece arithmetic only; the 256-thread form runs the default (two fp16 pieces)
    prev, prev_a = _set_ws(ws), _set_arith(3 if ws else -1)
    try:
        first, outs, kernel = run_case(name, 300)
    finally:
        _set_ws(prev)
        _set_arith(prev_a)
    assert _family(kernel).replace("<splitk>", "") == ("conv_halo3p_kernel" if ws else "conv_halo3r_kernel"), kernel
    bad = [i for i, o in enumerate(outs) if not torch.equal(o, first)]
    assert not bad, f"{name}: launches {bad[:5]} differ from launch 0"


@pytest.mark.parametrize("name", list(SHAPES))
def test_producer_consumer_form_equals_the_256_thread_form_bit_for_bit(name):
    prev, prev_a = _set_ws(1), _set_arith(3)
    try:
        a, _, ka = run_case(name, 1)
        _set_ws(0)
        b, _, kb = run_case(name, 1)
    finally:
        _set_ws(prev)
        _set_arith(prev_a)
    assert "halo3p" in ka and "halo3r" in kb, (ka, kb)
    assert torch.equal(a, b), f"{name}: max |diff| {(a - b).abs().max().item():.3e}"


def test_producer_consumer_data_gradient_equals_the_256_thread_form_bit_for_bit():
    """The MODE_DGRAD instantiation (mirrored taps) and its K-sliced form."""
    from codeps_amd import functional as F, _lib
    import ctypes
    lib = _lib.lib()
    lib.codeps_conv2d_last_kernel.restype = ctypes.c_char_p
    for b, h, w, ci, co in ((6, 32, 64, 256, 256), (4, 32, 64, 256, 512), (2, 64, 128, 256, 256)):
        g = torch.Generator().manual_seed(b * 100 + co)
        x = torch.randn(b, h, w, ci, generator=g).to(DEV)
        wt = (torch.randn(co, ci, 3, 3, generator=g) * 0.05).to(DEV)
        gy = torch.randn(b, h, w, co, generator=g).to(DEV)
        got = {}
        for ws in (1, 0):
            prev, prev_a = _set_ws(ws), _set_arith(3)
            try:
                xx = x.clone().requires_grad_(True)
                F.conv2d(xx, wt, pad=1).backward(gy)
                torch.cuda.synchronize()
                got[ws] = (xx.grad.clone(), lib.codeps_conv2d_last_kernel().decode())
            finally:
                _set_ws(prev)
                _set_arith(prev_a)
        assert "halo3p" in got[1][1] and "halo3r" in got[0][1], (got[1][1], got[0][1])
        assert torch.equal(got[1][0], got[0][0]), f"B{b} {ci}->{co}: data gradients differ"


def test_ring_kernel_equals_the_register_staged_kernel_bit_for_bit(tmp_path):
    out = str(tmp_path / "ref.pt")
    env = dict(os.environ, CODEPS_HALO_RING="0")
    code = _CHILD.format(root=ROOT, tests=os.path.join(ROOT, "tests"), names=list(SHAPES), out=out)
    subprocess.run([sys.executable, "-c", code], check=True, env=env, cwd=ROOT, timeout=600)
    ref = torch.load(out)
    for name in SHAPES:
        got, _, kernel = run_case(name, 1)
        assert _family(kernel).replace("<splitk>", "") in ("conv_halo3p_kernel", "conv_halo3r_kernel"), kernel
        assert torch.equal(got.cpu(), ref[name]), f"{name}: max |diff| {(got.cpu() - ref[name]).abs().max().item():.3e}"
