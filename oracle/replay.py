"""CPU oracle: replay-buffer diversity sampler.  TEST INFRA.

numpy restatement of the maximize_diversity branch of ReplayBuffer.add / load_state (datasets/replay_buffer.py:387-455,
475-499) as called per frame from scripts/adapt_codeps.py:307, INCLUDING its two-phase bookkeeping (similarity matrix built at
the first overflow, one row / column replaced per later insertion) -- the device kernel maintains the matrix incrementally, so
the tests also check that the two formulations agree.

PINNED (round 4) by tests/golden/replay.npz: the reference's OWN ReplayBuffer.add / save_state / load_state run frame by frame
(tests/golden/make_golden.py::golden_replay) -- threshold test, two-phase similarity-matrix bookkeeping, eviction rule,
previous-domain ids -- on a brute-force stand-in for the faiss calls it makes (tests/golden/_ref_import.py; faiss-gpu==1.7.2
of requirements.txt is neither vendored nor installed).  A flat inner-product index IS exact brute force, so what stays
unpinned of faiss is only the fp32 summation order inside its dot products; its published behaviour as restated here:
  * faiss.normalize_L2(x): x *= 1 / sqrt(<x, x>) per row when the norm is positive (fvec_renorm_L2);
  * IndexIDMap(IndexFlatIP): exact inner products against every stored vector, results sorted by decreasing score;
    add_with_ids appends, remove_ids deletes and keeps the order of the rest, id_map / reconstruct_n read in stored order.
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
                self.distance_matrix = self._pairwise_by_id()
                self.distance_matrix_indices = self.index.ids.copy()
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

    def _pairwise_by_id(self):
        """replay_buffer.py:416-420 / 491-494: all pairwise similarities, the columns of every row put in ASCENDING-ID order
        (`dist_mat[i, matching[i].argsort()]`) while the rows -- and distance_matrix_indices -- stay in stored order.  The two orders
        coincide whenever ids were stored in increasing order (every run that starts empty, and a loaded FULL state); they do not
        when a partially filled previous-domain state (ids 10000 + idx) is topped up with new frames before the first overflow."""
        n = self.index.ntotal
        dist, match = self.index.search(self.index.vecs, n)
        mat = np.empty((n, n), np.float32)
        for i in range(n):
            mat[i, :] = dist[i, np.argsort(match[i], kind="stable")]
        return mat

    def ids_stored_in_increasing_order(self):
        return bool(np.all(np.diff(self.index.ids) > 0)) if self.index is not None else True

    def load_state(self, target_samples_indices, features):
        """replay_buffer.py:475-499: the previous domain's samples enter a fresh index under id 10000 + idx, vectors as stored;
        a FULL loaded state gets its similarity matrix right away -- float64, with one spare row / column of -1 for the next frame."""
        self.buffer_indices["prev_target"] = [10000 + int(i) for i in target_samples_indices]
        self.index = _FlatIP(int(np.asarray(features[0]).size))
        for idx, f in zip(target_samples_indices, features):
            self.index.add_with_ids(np.asarray(f, np.float32).reshape(1, -1), [10000 + int(idx)])
        if self.index.ntotal == self.target_buffer_size:
            n = self.index.ntotal
            self.distance_matrix_indices = np.append(self.index.ids.copy(), [-1])
            self.distance_matrix = -np.ones((n + 1, n + 1))
            self.distance_matrix[:n, :n] = self._pairwise_by_id()

    def eviction_margin(self):
        """Gap between the two largest eviction scores of the current matrix (None before the first overflow): decisions
        with a gap at fp32-noise level may legitimately differ between summation orders."""
        if self.distance_matrix is None:
            return None
        valid = self.distance_matrix_indices >= 0
        m = self.distance_matrix[np.ix_(valid, valid)]
        s = np.sort(m.sum(0) - m.diagonal())
        return float(s[-1] - s[-2]) if s.size > 1 else None
