"""Device-side replay-buffer diversity sampler (codeps_replay_*, codeps_amd.datasets.DiversityReplayBuffer) against the
oracle's restatement of ReplayBuffer.add (datasets/replay_buffer.py:387-455) and, since round 4, against
tests/golden/replay.npz: the reference's OWN add / save_state / load_state run frame by frame (faiss replaced by an exact
brute-force stand-in, tests/golden/_ref_import.py).  Decisions are compared exactly wherever the margins exceed fp32 noise."""
import numpy as np
import pytest
import torch

from oracle import replay as orp

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _features(g, n, c, h, w, clusters=5):
    """Post-ReLU-like maps around a few cluster centres: some frames are near-duplicates (rejected), some are new."""
    centres = np.abs(g.standard_normal((clusters, c, 1, 1))).astype(np.float32)
    for i in range(n):
        k = g.integers(clusters)
        spread = g.choice([0.05, 0.5])
        yield np.maximum(centres[k] * (1 + spread * g.standard_normal((1, c, 1, 1))) + 0.3 * g.standard_normal((1, c, h, w)), 0).astype(np.float32)


@pytest.mark.parametrize("cap,c,h,w,thr,n", [(6, 64, 4, 8, 0.97, 60), (16, 2048, 16, 32, 0.95, 40), (5, 48, 3, 5, 2.0, 25)])
def test_add_matches_oracle(cap, c, h, w, thr, n):
    from codeps_amd import functional as F
    from codeps_amd.datasets import DiversityReplayBuffer
    g = np.random.default_rng(cap)
    ref = orp.DiversityBuffer(1000, cap, thr)
    buf = DiversityReplayBuffer(1000, cap, thr, DEV, feature_dim=c)
    want, margins = [], []
    for i, f in enumerate(_features(g, n, c, h, w)):
        want.append(ref.add(i, f))
        margins.append(ref.eviction_margin() if want[-1][1] >= 0 else None)
        x = torch.from_numpy(f).to(DEV)
        if i % 2:                                   # channels-last storage (what the encoder hands over) and plain NCHW
            x = F.to_nchw(F.from_nchw(x), c)
        buf.add(i, x)
    buf.sync()
    assert len(buf.log) == n
    evictions = 0
    for i, ((idx, stored, evicted, sim), (w_stored, w_evicted, w_sim)) in enumerate(zip(buf.log, want)):
        assert idx == i and abs(sim - w_sim) < 2e-6, (i, sim, w_sim)
        if abs(w_sim - thr) > 1e-5:
            assert stored == w_stored, i
        if w_evicted >= 0:
            evictions += 1
        assert evicted == w_evicted, f"frame {i}: evicted {evicted}, oracle {w_evicted}"
    assert evictions >= 3 and any(not s for _, s, _, _ in buf.log) == (thr < 1.5)
    assert np.array_equal(buf.buffer_indices["target"].numpy(), ref.buffer_indices["target"])
    assert int(buf.buffer_indices["target"].sum()) == cap
    state = buf.state_dict()
    assert sorted(state["target_samples_indices"]) == sorted(ref.index.ids.tolist())
    by_id = dict(zip(ref.index.ids.tolist(), ref.index.vecs))
    for i, v in zip(state["target_samples_indices"], state["faiss_features"]):
        np.testing.assert_allclose(v, by_id[i], atol=2e-6)
        assert abs(float(np.dot(v, v)) - 1.0) < 1e-5


def test_load_state_and_continual_eviction():
    from codeps_amd.datasets import DiversityReplayBuffer
    g = np.random.default_rng(7)
    c, cap = 32, 4
    first = DiversityReplayBuffer(100, cap, 2.0, DEV, feature_dim=c)
    feats = list(_features(g, 4, c, 2, 3))
    for i, f in enumerate(feats):
        first.add(i, torch.from_numpy(f).to(DEV))
    state = first.state_dict()
    assert state["target_samples_indices"] == [0, 1, 2, 3]
    second = DiversityReplayBuffer(100, cap, 2.0, DEV, feature_dim=c)
    second.load_state_dict(state)
    assert second.buffer_indices["prev_target"] == [10000, 10001, 10002, 10003]
    st2 = second.state_dict()
    for a, b in zip(state["faiss_features"], st2["faiss_features"]):
        assert np.array_equal(a, b)                                   # stored as they were, not renormalised
    second.add(5, torch.from_numpy(next(_features(g, 1, c, 2, 3))).to(DEV))
    second.sync()
    idx, stored, evicted, _ = second.log[-1]
    assert stored and (evicted >= 10000 or evicted == 5)
    if evicted >= 10000:
        assert evicted not in second.buffer_indices["prev_target"] and int(second.buffer_indices["target"][5]) == 1
    with pytest.raises(ValueError):
        second.add(6, torch.zeros(2, c, 2, 3, device=DEV))


# ------------------------------------------------------------------------------------------ pinned against the REFERENCE (round 4)
def _golden():
    import json
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "replay.npz"), allow_pickle=False)
    return z, json.loads(str(z["meta"]))


def _inputs():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import inputs as gi
    return gi


def _compare_log(z, tag, log, thr, first=0):
    """Device decisions (call order) against the reference's ReplayBuffer.add; a difference is admissible only on a frame whose
    margin in the reference is fp32 noise (<= 1e-5), after which the two buffers differ and the comparison stops."""
    for i, (idx, stored, evicted, sim) in enumerate(log):
        w_stored, w_evicted, w_sim, margin = bool(z[tag + "_stored"][i]), int(z[tag + "_evicted"][i]), float(z[tag + "_sim"][i]), float(z[tag + "_margin"][i])
        assert idx == first + i and abs(sim - w_sim) < 2e-6, (tag, i, sim, w_sim)
        if abs(w_sim - thr) <= 1e-5 and stored != w_stored:
            return i
        assert stored == w_stored, (tag, i)
        if w_evicted >= 0 and margin <= 1e-5 and evicted != w_evicted:
            return i
        assert evicted == w_evicted, f"{tag}: frame {i}: device evicts {evicted}, the reference {w_evicted} (margin {margin:.2e})"
    return len(log)


@pytest.mark.parametrize("tag", ["small", "full", "keepall"])
def test_add_matches_the_reference_frame_by_frame(tag):
    """tests/golden/replay.npz = the reference's own ReplayBuffer.add (datasets/replay_buffer.py:387-455) over the same frames."""
    from codeps_amd.datasets import DiversityReplayBuffer
    gi = _inputs()
    z, meta = _golden()
    c = meta["cases"][tag]
    buf = DiversityReplayBuffer(1000, c["capacity"], c["threshold"], DEV, feature_dim=c["c"])
    for i, f in enumerate(gi.replay_features(c["n"], c["c"], c["h"], c["w"], c["seed"])):
        buf.add(i, f.to(DEV))
    buf.sync()
    done = _compare_log(z, tag, buf.log, c["threshold"])
    assert done >= min(c["n"], 30), f"{tag}: comparison stopped at frame {done} on a near-tie"
    if done == c["n"]:
        state = buf.state_dict()
        assert sorted(state["target_samples_indices"]) == sorted(z[tag + "_final_ids"].tolist())
        by_id = dict(zip(z[tag + "_final_ids"].tolist(), z[tag + "_final_vecs"]))
        for i, v in zip(state["target_samples_indices"], state["faiss_features"]):
            np.testing.assert_allclose(v, by_id[i], atol=2e-6)
        assert np.array_equal(buf.buffer_indices["target"].numpy(), z[tag + "_final_target"])


def test_second_domain_from_a_loaded_state_matches_the_reference():
    """save_state -> load_state -> add (replay_buffer.py:462-499, 447-451): previous-domain samples under id 10000 + idx compete
    with the new domain's frames for the same slots."""
    from codeps_amd.datasets import DiversityReplayBuffer
    gi = _inputs()
    z, meta = _golden()
    c = meta["cases"][meta["domain2"]["from_case"]]
    first = DiversityReplayBuffer(1000, c["capacity"], c["threshold"], DEV, feature_dim=c["c"])
    for i, f in enumerate(gi.replay_features(c["n"], c["c"], c["h"], c["w"], c["seed"])):
        first.add(i, f.to(DEV))
    first.sync()
    assert _compare_log(z, "small", first.log, c["threshold"]) == c["n"]
    state = first.state_dict()
    order = np.argsort(state["target_samples_indices"])          # the reference saves in stored order = increasing id
    state = {"target_samples_indices": [state["target_samples_indices"][k] for k in order],
             "faiss_features": [state["faiss_features"][k] for k in order]}
    second = DiversityReplayBuffer(1000, c["capacity"], c["threshold"], DEV, feature_dim=c["c"])
    second.load_state_dict(state)
    assert second.buffer_indices["prev_target"] == z["domain2_prev_target"].tolist()
    second.log.clear()
    n2 = meta["domain2"]["n"]
    for i, f in enumerate(gi.replay_features(n2, c["c"], c["h"], c["w"], meta["domain2"]["seed"])):
        second.add(i, f.to(DEV))
    second.sync()
    assert _compare_log(z, "domain2", second.log, c["threshold"]) == n2
    assert sorted(second.buffer_indices["prev_target"]) == sorted(z["domain2_final_prev_target"].tolist())
    assert np.array_equal(second.buffer_indices["target"].numpy(), z["domain2_final_target"])
    assert sorted(second.state_dict()["target_samples_indices"]) == sorted(z["domain2_final_ids"].tolist())
