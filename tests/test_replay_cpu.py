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


# ------------------------------------------------------------------------------------------ pinned against the REFERENCE (round 4)
def _golden():
    import json
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "replay.npz"), allow_pickle=False)
    return z, json.loads(str(z["meta"]))


def _check_run(z, tag, buf, frames, first_index=0):
    """Frame-by-frame decisions of `buf` (oracle) against the reference's recorded ones.  A decision may legitimately differ only
    where the reference's own margin is at fp32-noise level (similarity within 1e-5 of the threshold, or the two largest
    eviction scores within 1e-5 of each other); after such a frame the two buffers hold different sets and the comparison stops."""
    for i, f in enumerate(frames):
        stored, evicted, sim = buf.add(first_index + i, f.numpy())
        w_stored, w_evicted, w_sim, margin = bool(z[tag + "_stored"][i]), int(z[tag + "_evicted"][i]), float(z[tag + "_sim"][i]), float(z[tag + "_margin"][i])
        assert abs(sim - w_sim) < 2e-6, (tag, i, sim, w_sim)
        if abs(w_sim - buf.similarity_threshold) <= 1e-5 and stored != w_stored:
            return i
        assert stored == w_stored, (tag, i)
        if w_evicted >= 0 and margin <= 1e-5 and evicted != w_evicted:
            return i
        assert evicted == w_evicted, f"{tag}: frame {i}: oracle evicts {evicted}, the reference {w_evicted} (margin {margin:.2e})"
    return len(frames)


def test_oracle_add_equals_the_reference_frame_by_frame():
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import inputs as gi
    z, meta = _golden()
    for tag, c in meta["cases"].items():
        buf = orp.DiversityBuffer(1000, c["capacity"], c["threshold"])
        frames = gi.replay_features(c["n"], c["c"], c["h"], c["w"], c["seed"])
        done = _check_run(z, tag, buf, frames)
        assert done >= min(c["n"], 30), f"{tag}: comparison stopped at frame {done} on a near-tie"
        if done == c["n"]:
            assert np.array_equal(buf.index.ids, z[tag + "_final_ids"])
            np.testing.assert_allclose(buf.index.vecs, z[tag + "_final_vecs"], atol=2e-7)
            assert np.array_equal(buf.buffer_indices["target"], z[tag + "_final_target"])
        assert (z[tag + "_evicted"] >= 0).sum() >= 10


def test_oracle_load_state_and_second_domain_equal_the_reference():
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import inputs as gi
    z, meta = _golden()
    c = meta["cases"][meta["domain2"]["from_case"]]
    first = orp.DiversityBuffer(1000, c["capacity"], c["threshold"])
    assert _check_run(z, "small", first, gi.replay_features(c["n"], c["c"], c["h"], c["w"], c["seed"])) == c["n"]
    second = orp.DiversityBuffer(1000, c["capacity"], c["threshold"])
    second.load_state(first.index.ids.tolist(), list(first.index.vecs))              # = save_state's pickle (replay_buffer.py:462-473)
    assert np.array_equal(second.index.ids, z["domain2_loaded_ids"])
    assert second.buffer_indices["prev_target"] == z["domain2_prev_target"].tolist()
    n2 = meta["domain2"]["n"]
    assert _check_run(z, "domain2", second, gi.replay_features(n2, c["c"], c["h"], c["w"], meta["domain2"]["seed"])) == n2
    assert np.array_equal(second.index.ids, z["domain2_final_ids"])
    assert second.buffer_indices["prev_target"] == z["domain2_final_prev_target"].tolist()
    assert np.array_equal(second.buffer_indices["target"], z["domain2_final_target"])
    assert (z["domain2_evicted"] >= 10000).sum() >= 1, "the fixture must evict previous-domain samples too"
