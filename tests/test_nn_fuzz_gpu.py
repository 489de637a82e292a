"""Randomised shapes through the non-conv layer kernels (train-mode BatchNorm with residual / ReLU, 3x3/2 max pooling, bilinear
up-sampling) against float64 on the CPU: single pixels, odd sizes, channel counts from one 16-channel group to 2048.  Seeded."""
import random

import pytest

from test_ops_gpu import run_bilinear_case, run_bn_case, run_maxpool_case

pytestmark = pytest.mark.gpu


def _bn_cases(n, seed):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        c = rng.choice([16, 32, 48, 64, 128, 256, 512, 1024, 2048])
        b, h, w = rng.randint(1, 5), rng.randint(1, 40), rng.randint(1, 50)
        if b * h * w < 4 or b * h * w * c > 4e6:
            b, h, w = 2, 6, 7
        out.append((c, rng.choice([None, "relu"]), rng.random() < 0.5, b, h, w))
    return out


@pytest.mark.parametrize("case", _bn_cases(24, 7), ids=lambda c: "-".join(str(v) for v in c))
def test_random_batch_norm_shapes(case):
    c, act, res, b, h, w = case
    run_bn_case(c, act, res, b, h, w, 100 + c + h * w)


@pytest.mark.parametrize("i", range(16))
def test_random_maxpool_shapes(i):
    rng = random.Random(900 + i)
    run_maxpool_case(rng.randint(1, 4), rng.choice([16, 64, 128, 256]), rng.randint(1, 40), rng.randint(1, 60), i)


@pytest.mark.parametrize("i", range(16))
def test_random_bilinear_shapes(i):
    rng = random.Random(1900 + i)
    h, w = rng.randint(1, 24), rng.randint(1, 30)
    oh, ow = h * rng.randint(1, 4) + rng.randint(0, 3), w * rng.randint(1, 4) + rng.randint(0, 3)
    run_bilinear_case(rng.random() < 0.5, rng.randint(1, 3), rng.choice([16, 32, 64, 144]), h, w, oh, ow, i)
