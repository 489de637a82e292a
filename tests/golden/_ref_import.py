"""Import shim for the *real* CoDEPS reference (development container only).

Used ONLY by tests/golden/make_golden.py to generate fixtures.  /root/reference does not exist on
the GPU box, nothing in tests/, bench.py or __graft_entry__ imports this module at run time.
Missing third-party packages of the reference (torchvision, cv2, yacs, ...) are replaced by inert
placeholders so that the reference's own `models`, `algos`, `misc`, `codeps` packages import and run
on CPU (recipe: SURVEY.md Appendix B).
"""
import copy
import sys
import types
from unittest.mock import MagicMock

from torch import nn

REF = "/root/reference"


def _faiss_stand_in():
    """faiss (requirements.txt: faiss-gpu==1.7.2) is absent.  The reference's ReplayBuffer.add / load_state / save_state
    (datasets/replay_buffer.py:387-499) use exactly: IndexIDMap(index_factory(d, "Flat", METRIC_INNER_PRODUCT)) with ntotal /
    search / add_with_ids / remove_ids / id_map.at / index.reconstruct(_n), normalize_L2 and vector_to_array.  A flat
    inner-product index is EXACT brute force, so a numpy brute force reproduces it up to the fp32 summation order of the dot
    products (faiss: SIMD / BLAS sgemm; here: numpy's float32 matmul) -- the stand-in exists so that the reference's OWN `add`
    (threshold test, two-phase similarity-matrix bookkeeping, eviction rule, prev_target handling) runs and produces
    tests/golden/replay.npz.  Written independently of oracle/replay.py (which it pins)."""
    import numpy as np
    mod = types.ModuleType("faiss")
    mod.METRIC_INNER_PRODUCT = 0

    class _IdVec:
        def __init__(self, owner):
            self._o = owner

        def at(self, i):
            return int(self._o._ids[i])

    class _Flat:
        def __init__(self, d):
            self.d, self._x = d, np.zeros((0, d), np.float32)

        def reconstruct(self, i):
            return self._x[i].copy()

        def reconstruct_n(self, i0, n):
            return self._x[i0:i0 + n].copy()

    class IndexIDMap:
        def __init__(self, index):
            self.index, self._ids = index, np.zeros(0, np.int64)
            self.id_map = _IdVec(self)
            self.top1_log = []

        @property
        def ntotal(self):
            return len(self._ids)

        def add_with_ids(self, x, ids):
            x = np.ascontiguousarray(x, np.float32).reshape(-1, self.index.d)
            self.index._x = np.vstack([self.index._x, x])
            self._ids = np.append(self._ids, np.asarray(ids, np.int64))

        def search(self, x, k):
            x = np.ascontiguousarray(x, np.float32).reshape(-1, self.index.d)
            ip = x @ self.index._x.T                                   # float32 brute force
            D = np.full((len(x), k), -3.4028235e38, np.float32)
            I = np.full((len(x), k), -1, np.int64)
            for r in range(len(x)):
                o = np.argsort(-ip[r], kind="stable")[:k]              # decreasing score, earlier-stored first among ties
                D[r, :len(o)], I[r, :len(o)] = ip[r, o], self._ids[o]
            if k == 1:
                self.top1_log.append(float(D[0, 0]))
            return D, I

        def remove_ids(self, ids):
            keep = ~np.isin(self._ids, np.asarray(ids, np.int64))     # IndexFlat / IndexIDMap compact, order of the rest kept
            self.index._x, self._ids = self.index._x[keep], self._ids[keep]
            return int((~keep).sum())

    def index_factory(d, desc, metric=0):
        assert desc == "Flat" and metric == mod.METRIC_INNER_PRODUCT
        return _Flat(int(d))

    def normalize_L2(x):                                              # in place, fvec_renorm_L2: x *= 1 / sqrt(<x, x>) when > 0
        assert x.dtype == np.float32 and x.ndim == 2
        for r in range(x.shape[0]):
            nr = np.float32(np.dot(x[r], x[r]))
            if nr > 0:
                x[r] *= np.float32(1.0) / np.sqrt(nr, dtype=np.float32)

    def vector_to_array(v):
        return v._o._ids.copy()

    mod.IndexIDMap, mod.index_factory, mod.normalize_L2, mod.vector_to_array = IndexIDMap, index_factory, normalize_L2, vector_to_array
    return mod


def install():
    if REF in sys.path:
        return
    sys.path.insert(0, REF)
    tv, tvm, tvr = (types.ModuleType(n) for n in
                    ("torchvision", "torchvision.models", "torchvision.models.resnet"))
    tv.__version__ = "0.13.1"

    class _R(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    tvm.ResNet, tvr.BasicBlock, tvr.Bottleneck, tvm.resnet, tv.models = _R, object, object, tvr, tvm
    for n in ("resnet18", "resnet34", "resnet50", "resnet101", "resnet152"):
        setattr(tvm, n, None)
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm,
                        "torchvision.models.resnet": tvr,
                        "torchvision.transforms": MagicMock(),
                        "torchvision.transforms.functional": MagicMock()})
    for m in ("cv2", "kornia", "kornia.contrib", "skimage", "skimage.exposure", "mmcv",
              "mmcv.parallel", "mmcv.parallel.data_container", "wandb", "seaborn"):
        sys.modules[m] = MagicMock()
    sys.modules["faiss"] = _faiss_stand_in()

    def _distance_transform(image, kernel_size=3, h=0.35):
        """kornia is absent: the reference's `conf_instance_mixup` runs on the oracle's restatement of
        kornia.contrib.distance_transform (oracle/mixup.py; parity unpinned for that function, pinned for everything the
        reference builds around it)."""
        import numpy as np
        import torch
        from oracle.mixup import kornia_distance_transform
        out = np.stack([np.stack([kornia_distance_transform(c.numpy(), kernel_size, h) for c in b]) for b in image])
        return torch.from_numpy(out)

    sys.modules["kornia.contrib"].distance_transform = _distance_transform

    class CfgNode(dict):
        def __getattr__(self, k):
            if k in self:
                return self[k]
            raise AttributeError(k)
        __setattr__ = dict.__setitem__

        def clone(self):
            return copy.deepcopy(self)

    y, yc = types.ModuleType("yacs"), types.ModuleType("yacs.config")
    yc.CfgNode = CfgNode
    y.config = yc
    sys.modules.update({"yacs": y, "yacs.config": yc})
