"""Randomised shapes through the conv engine (forward, data / weight / bias / residual gradients) against float64 on the CPU:
sizes that are not multiples of any tile, single rows / columns, every kernel family the dispatcher can pick (halo tile,
wave-specialised, 256-thread implicit GEMM with and without split-K, 16-column, the weight-gradient variants), fused nearest
x2 + concat, reflection padding, stride 2, bias / affine / residual / activation epilogues.  Seeded: the cases are the same
on every run."""
import random
import zlib

import pytest
import torch

from test_ops_gpu import run_conv_case

pytestmark = pytest.mark.gpu


def _cases(n, seed):
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        k = rng.choice([1, 1, 3, 3, 3])
        stride = rng.choice([1, 1, 1, 2])
        up0 = rng.random() < 0.2 and stride == 1
        c0 = rng.choice([16, 32, 48, 64, 96, 128, 256, 512])
        c1 = rng.choice([16, 32, 64, 128]) if (up0 or rng.random() < 0.1) else 0
        co = rng.choice([1, 2, 14, 16, 19, 32, 48, 64, 128, 144, 256, 512])
        shape = rng.choice(["tiny", "odd", "halo", "wide"])
        if shape == "tiny":
            b, h, w = rng.randint(1, 3), rng.randint(1, 6), rng.randint(1, 9)
        elif shape == "odd":
            b, h, w = rng.randint(1, 3), rng.randint(5, 37), rng.randint(5, 67)
        elif shape == "halo":
            b, h, w = rng.randint(2, 6), 4 * rng.randint(2, 10), 32 * rng.randint(1, 3)
        else:
            b, h, w = rng.randint(1, 2), rng.randint(8, 24), rng.randint(64, 200)
        if up0:
            h, w = max(2, h // 2 * 2), max(2, w // 2 * 2)
        if (c0 + c1) * k * k * b * h * w * co > 6e9:          # keep the float64 reference to a second or two
            continue
        pad = k // 2
        reflect = k == 3 and co <= 2 and not up0 and c1 == 0 and min(h, w) >= 2 and stride == 1 and rng.random() < 0.7
        if stride == 2 and (h < 2 or w < 2):
            continue
        act = rng.choice([None, "relu", "elu"])
        bias = rng.random() < 0.5
        affine = not bias and rng.random() < 0.5
        residual = rng.random() < 0.25
        out.append((b, h, w, c0, c1, int(up0), co, k, stride, pad, "reflect" if reflect else "zeros", act, bias, affine, residual))
    return out


@pytest.mark.parametrize("case", _cases(64, 2024), ids=lambda c: "-".join(str(v) for v in c))
def test_random_conv_shapes(case):
    run_conv_case(case, zlib.crc32(repr(case).encode()) & 0xFFFF)
