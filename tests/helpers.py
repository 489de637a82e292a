"""Shared helpers for the parity tests."""
import json
import os
from collections import OrderedDict

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"])) if "meta" in z.files else {}
    return z, meta


def keys_of(z, name):
    return OrderedDict((k, tuple(s)) for k, s in json.loads(str(z[name + "_keys"])))


def t(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, atol, rtol=0.0, what=""):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    err = np.abs(a.astype(np.float64) - b.astype(np.float64))
    tol = atol + rtol * np.abs(b.astype(np.float64))
    assert (err <= tol).all(), f"{what}: max err {err.max():.3e} (tol {atol:g}+{rtol:g}*|ref|), at {np.unravel_index(err.argmax(), err.shape)}"


def grad_stats(named_grads):
    """(random projection, l2) per gradient tensor -- same definition as tests/golden/make_golden.py."""
    import inputs as gi
    rows = []
    for i, g in enumerate(named_grads):
        if g is None:
            rows.append([0.0, 0.0])
            continue
        g = g.detach().double().cpu()
        rows.append([(g * gi.normal(tuple(g.shape), 4242 + i).double()).sum().item(), g.norm().item()])
    return np.array(rows)


_HEAD_KEYS = {"depth_head": "depth", "pose_head": "pose", "semantic_head": "sem", "semantic_head_noguda": "semng",
              "instance_head": "inst", "instance_head_nomask": "instnomask"}


def head_shapes(name, tag="r18"):
    """State-dict key -> shape of the REFERENCE module (recorded by make_golden.py)."""
    z, _ = load("heads_" + tag)
    return keys_of(z, _HEAD_KEYS[name])
