"""DiversityReplayBuffer: the maximize_diversity branch of the reference's ReplayBuffer (datasets/replay_buffer.py:387-455,
462-499) with its faiss index, similarity matrix and bookkeeping on the device (kernel K18).

The reference's `add(index, image_features)` is called once per adapted frame (scripts/adapt_codeps.py:307) and blocks on a
device->host copy of the pooled feature.  Here `add` only enqueues three small kernels; the outcome (stored / rejected, evicted
id) lands in pinned host memory behind an event and is folded into `buffer_indices` by `sync()` -- which the data loader side
calls when it is about to draw replay samples (replay_buffer.py:205-230), i.e. a step later, when the work has long finished.
The dataset / sampling half of the reference class (file loading, source sampling, mixup pairing) is host-side data plumbing
and stays the reference's.
"""
from typing import Any, Dict, List, Optional

import numpy as np
import torch
from torch import Tensor

from .. import functional as F
from .._lib import CodepsHipError, check, lib, ptr, require_cuda, stream_ptr


class DiversityReplayBuffer:
    def __init__(self, num_target_samples: int, target_buffer_size: int, similarity_threshold: float, device, feature_dim: int = 2048):
        self.target_buffer_size = int(target_buffer_size)
        self.similarity_threshold = float(similarity_threshold)
        self.feature_dim = int(feature_dim)
        self.device = torch.device(device)
        self.buffer_indices: Dict[str, Any] = {"target": torch.zeros(num_target_samples, dtype=torch.int32)}
        nbytes = lib().codeps_replay_state_bytes(self.target_buffer_size, self.feature_dim)
        self._state = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        check(lib().codeps_replay_reset(ptr(self._state), nbytes, self.target_buffer_size, self.feature_dim, stream_ptr()), "codeps_replay_reset")
        self._pending: List[Any] = []           # (index, pinned result, event)
        self.log: List[Any] = []                # (index, stored, evicted id, similarity) in call order, filled by sync()

    # ------------------------------------------------------------------ reference API
    def add(self, index: int, image_features: Optional[Tensor] = None):
        """replay_buffer.py:387-455.  image_features: the shared encoder's last feature map of the frame, [1,C,h,w]
        (CodepsNet.adapt's result["image_features"]; channels-last storage is read in place)."""
        assert image_features is not None
        self._enqueue(int(index), image_features, force=False)

    def sync(self):
        """Fold finished decisions into buffer_indices (waits for the newest one)."""
        for index, host, done in self._pending:
            done.synchronize()
            stored, evicted, bits = (int(v) for v in host.tolist())
            if stored < 0:
                raise CodepsHipError("codeps_replay_add: no free slot in the device index")
            sim = float(np.array([bits], np.uint32).view(np.float32)[0])
            self.log.append((index, bool(stored), evicted, sim))
            if stored and index < 10000:
                self.buffer_indices["target"][index] = 1
            if evicted >= 10000:                                            # "hardcoded continual learning", replay_buffer.py:447-451
                self.buffer_indices["prev_target"].remove(evicted)
            elif evicted >= 0:
                self.buffer_indices["target"][evicted] = 0
        self._pending = []

    def state_dict(self):
        """The picklable part of save_state (replay_buffer.py:462-473): ids and stored vectors in slot order."""
        self.sync()
        cap, dim = self.target_buffer_size, self.feature_dim
        ids = torch.empty(cap + 1, dtype=torch.int64, device=self.device)
        feats = torch.empty(cap + 1, dim, dtype=torch.float32, device=self.device)
        check(lib().codeps_replay_export(ptr(self._state), self._state.numel(), cap, dim, ptr(ids), ptr(feats), stream_ptr()), "codeps_replay_export")
        ids, feats = ids.cpu(), feats.cpu()
        keep = ids >= 0
        return {"target_samples_indices": ids[keep].tolist(), "faiss_features": [f.numpy() for f in feats[keep]]}

    def load_state_dict(self, state):
        """load_state (replay_buffer.py:475-499): the previous domain's samples enter the index under id 10000 + idx."""
        idxs, feats = state["target_samples_indices"], state["faiss_features"]
        assert len(idxs) <= self.target_buffer_size, "more stored samples than the buffer holds"
        self.buffer_indices["prev_target"] = [10000 + int(i) for i in idxs]
        for i, f in zip(idxs, feats):
            v = torch.as_tensor(np.asarray(f, np.float32).reshape(1, -1, 1, 1)).to(self.device)
            self._enqueue(10000 + int(i), v, force=True)
        self.sync()

    # ------------------------------------------------------------------ plumbing
    def _enqueue(self, index, image_features, force):
        require_cuda(image_features)
        b, c, h, w = image_features.shape
        if b != 1 or c != self.feature_dim:
            raise ValueError(f"expected one [1,{self.feature_dim},h,w] feature map, got {tuple(image_features.shape)}")
        nhwc = F.from_nchw(image_features.detach())                        # zero-copy for the encoder's own outputs
        result = torch.empty(3, dtype=torch.int64, device=self.device)
        check(lib().codeps_replay_add(ptr(self._state), self._state.numel(), self.target_buffer_size, self.feature_dim, ptr(nhwc), h, w,
                                      nhwc.shape[3], index, self.similarity_threshold, int(force), ptr(result), stream_ptr()),
              "codeps_replay_add")
        host = torch.empty(3, dtype=torch.int64, pin_memory=True)
        host.copy_(result, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        self._pending.append((index, host, done))
