"""CPU oracle: replay-buffer diversity sampler.  TEST INFRA.

numpy restatement of the maximize_diversity branch of ReplayBuffer.add / load_state (datasets/replay_buffer.py:387-455,
475-499) as called per frame from scripts/adapt_codeps.py:307, INCLUDING its two-phase bookkeeping (similarity matrix built at
the first overflow, one row / column replaced per later insertion) -- the device kernel maintains the matrix incrementally, so
the tests also check that the two formulations agree.

PARITY UNPINNED: the algorithm's search runs inside faiss (requirements.txt: faiss-gpu==1.7.2), which is neither vendored
under the reference nor installed here, so the reference's `add` cannot be executed to produce golden vectors.  What this
file restates of faiss is its published behaviour for the three calls the reference makes:
  * faiss.normalize_L2(x): x *= 1 / sqrt(<x, x>) per row when the norm is positive (fvec_renorm_L2);
  * IndexIDMap(IndexFlatIP): exact inner products against every stored vector, results sorted by decreasing score;
    add_with_ids appends, remove_ids deletes and keeps the order of the rest, id_map / reconstruct_n read in stored order.
Anchored on the reference's call sites only (no golden vectors exist for this path in the reference's repository).
"""
import numpy as np


class _FlatIP:
    """What the reference uses of faiss.IndexIDMap(faiss.index_factory(d, "Flat", METRIC_INNER_PRODUCT))."""

    def __init__(self, dim):
        self.vecs = np.zeros((0, dim), np.float32)
        self.ids = np.zeros((0,), np.int64)

    @property
    def ntotal(self):
        return self.vecs.shape[0]

    def add_with_ids(self, x, ids):
        self.vecs = np.concatenate([self.vecs, x.astype(np.float32)])
        self.ids = np.concatenate([self.ids, np.asarray(ids, np.int64)])

    def search(self, x, k):
        scores = (x.astype(np.float32) @ self.vecs.T).astype(np.float32)
        order = np.argsort(-scores, axis=1, kind="stable")[:, :k]
        return np.take_along_axis(scores, order, 1), self.ids[order]

    def remove_ids(self, ids):
        keep = ~np.isin(self.ids, ids)
        self.vecs, self.ids = self.vecs[keep], self.ids[keep]


def normalize_l2(x):
    nr = (x * x).sum(1, dtype=np.float32)
    inv = np.where(nr > 0, np.float32(1) / np.sqrt(nr, dtype=np.float32), np.float32(1)).astype(np.float32)
    return x * inv[:, None]


def pool(image_features):
    """image_features.mean(-1).mean(-1) (replay_buffer.py:392) for [1,C,h,w] float32."""
    return image_features.mean(-1, dtype=np.float32).mean(-1, dtype=np.float32)


class DiversityBuffer:
    def __init__(self, num_target_samples, target_buffer_size, similarity_threshold):
        self.target_buffer_size, self.similarity_threshold = target_buffer_size, similarity_threshold
        self.buffer_indices = {"target": np.zeros(num_target_samples, np.int32)}
        self.index = None
        self.distance_matrix = None
        self.distance_matrix_indices = None

    def add(self, index, image_features):
        """-> (stored?, evicted id or -1, largest similarity)."""
        feats = normalize_l2(pool(image_features).astype(np.float32))
        if self.index is None:
            self.index = _FlatIP(feats.shape[1])
        extra = len(self.buffer_indices.get("prev_target", []))
        similarity = np.float32(0) if self.index.ntotal == 0 else self.index.search(feats, 1)[0][0, 0]
        if not similarity < self.similarity_threshold:
            return False, -1, float(similarity)
        self.index.add_with_ids(feats, [index])
        self.buffer_indices["target"][index] = 1
        removed = -1
        if self.buffer_indices["target"].sum() + extra > self.target_buffer_size:
            n = self.index.ntotal
            if self.distance_matrix is None:                                   # replay_buffer.py:415-423
                dist, match = self.index.search(self.index.vecs, n)
                pos = {int(i): k for k, i in enumerate(self.index.ids)}
                mat = np.empty((n, n), np.float32)
                for i in range(n):
                    mat[i, [pos[int(m)] for m in match[i]]] = dist[i]
                self.distance_matrix, self.distance_matrix_indices = mat, self.index.ids.copy()
            else:                                                              # replay_buffer.py:425-437
                fill = int(np.argwhere(self.distance_matrix_indices < 0)[0, 0])
                a, b = self.index.search(feats, n)
                self.distance_matrix_indices[fill] = index
                by_id = dict(zip(b[0].tolist(), a[0].tolist()))
                row = np.array([by_id[int(i)] for i in self.distance_matrix_indices], np.float32)
                self.distance_matrix[fill, :] = self.distance_matrix[:, fill] = row
            tmp = int(np.argmax(self.distance_matrix.sum(0) - self.distance_matrix.diagonal()))
            self.distance_matrix[:, tmp] = self.distance_matrix[tmp, :] = -1
            removed = int(self.distance_matrix_indices[tmp])
            self.distance_matrix_indices[tmp] = -1
            self.index.remove_ids([removed])
            if removed >= 10000:                                               # replay_buffer.py:447-451
                self.buffer_indices["prev_target"].remove(removed)
            else:
                self.buffer_indices["target"][removed] = 0
        return True, removed, float(similarity)

    def eviction_margin(self):
        """Gap between the two largest eviction scores of the current matrix (None before the first overflow): decisions
        with a gap at fp32-noise level may legitimately differ between summation orders."""
        if self.distance_matrix is None:
            return None
        valid = self.distance_matrix_indices >= 0
        m = self.distance_matrix[np.ix_(valid, valid)]
        s = np.sort(m.sum(0) - m.diagonal())
        return float(s[-1] - s[-2]) if s.size > 1 else None
