"""oracle/replay.py self-consistency (parity unpinned, see its header): the reference's two-phase similarity-matrix
bookkeeping must select the same evictions as recomputing all pairwise cosine similarities from scratch."""
import numpy as np

from oracle import replay as orp


def test_two_phase_bookkeeping_equals_full_recompute():
    g = np.random.default_rng(1)
    cap, c = 5, 40
    buf = orp.DiversityBuffer(200, cap, 2.0)
    kept = {}
    for i in range(40):
        f = np.abs(g.standard_normal((1, c, 3, 4))).astype(np.float32)
        stored, evicted, sim = buf.add(i, f)
        assert stored
        kept[i] = orp.normalize_l2(orp.pool(f))[0]
        if len(kept) > cap:
            ids = sorted(kept)
            m = np.stack([kept[k] for k in ids]) @ np.stack([kept[k] for k in ids]).T
            score = m.sum(0) - m.diagonal()
            top = np.sort(score)
            want = ids[int(np.argmax(score))]
            if top[-1] - top[-2] > 1e-5:
                assert evicted == want, i
            del kept[evicted]
        else:
            assert evicted == -1
        assert sorted(buf.index.ids.tolist()) == sorted(kept)
    assert int(buf.buffer_indices["target"].sum()) == cap


def test_threshold_and_empty_index():
    buf = orp.DiversityBuffer(10, 3, 0.95)
    f = np.ones((1, 8, 2, 2), np.float32)
    assert buf.add(0, f) == (True, -1, 0.0)                     # empty index: similarity 0 (replay_buffer.py:404-405)
    stored, evicted, sim = buf.add(1, 2 * f)                    # same direction: cosine 1 -> rejected
    assert not stored and evicted == -1 and abs(sim - 1) < 1e-6
    assert buf.buffer_indices["target"].tolist()[:2] == [1, 0]
