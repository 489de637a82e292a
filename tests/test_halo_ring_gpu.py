"""conv_halo3r_kernel (weights by LDS DMA, hand-counted vmcnt waits, fragment reads under the MFMAs) against the kernel it replaces:
the products of every accumulator are issued in the same order, so the two must agree BIT FOR BIT, and a missed wait would show up
as a launch-to-launch difference -- 300 launches per shape must return the same bits."""
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


@pytest.mark.parametrize("name", list(SHAPES))
def test_ring_kernel_is_repeatable(name):
    first, outs, kernel = run_case(name, 300)
    assert kernel == "conv_halo3r_kernel", kernel
    bad = [i for i, o in enumerate(outs) if not torch.equal(o, first)]
    assert not bad, f"{name}: launches {bad[:5]} differ from launch 0"


def test_ring_kernel_equals_the_register_staged_kernel_bit_for_bit(tmp_path):
    out = str(tmp_path / "ref.pt")
    env = dict(os.environ, CODEPS_HALO_RING="0")
    code = _CHILD.format(root=ROOT, tests=os.path.join(ROOT, "tests"), names=list(SHAPES), out=out)
    subprocess.run([sys.executable, "-c", code], check=True, env=env, cwd=ROOT, timeout=600)
    ref = torch.load(out)
    for name in SHAPES:
        got, _, kernel = run_case(name, 1)
        assert kernel == "conv_halo3r_kernel", kernel
        assert torch.equal(got.cpu(), ref[name]), f"{name}: max |diff| {(got.cpu() - ref[name]).abs().max().item():.3e}"
