"""Tracking / motion boxes / label propagation against vectors produced by the REFERENCE's own classes
(tests/golden/track_golden.pkl <- tests/golden/make_golden.py track: src/vilgod/tracker.py, src/dataclass/objects.py,
src/utils/tracking_utils.py run unchanged on a seeded scene; `filterpy` and the pcdet IoU op, both absent here, replaced by
stand-ins written from their published formulas -- parity unpinned for those two dependencies)."""
import os
import pickle

import numpy as np
import pytest

from oracle import segment_oracle as so
from vilgod_amd import synthetic, tracking


@pytest.fixture(scope='module')
def gold(golden_dir):
    with open(os.path.join(golden_dir, 'track_golden.pkl'), 'rb') as f:
        return pickle.load(f)


@pytest.fixture(scope='module')
def scene():
    frames, poses = synthetic.make_sequence(seed=11, n_frames=18, n_points=5000, n_objects=9, moving_frac=0.6)
    X = []
    for f, p in zip(frames, poses):
        pr = so.apply_transform(f, np.linalg.inv(poses[0]) @ p)
        X.append(np.ascontiguousarray(pr[pr[:, 2] > 0.25]))
    return frames, poses, X


def run_tracker(gold, X):
    tr = tracking.Tracker(mode='cluster_center', max_distance=1.0, min_length=5, max_missed=3)
    med, cnt = {}, {}
    for fnr, dets in enumerate(gold['clusters']):
        keys = [(fnr, cid) for cid, _ in dets]
        for (cid, idx) in dets:
            med[(fnr, cid)] = np.median(X[fnr][idx], axis=0)
            cnt[(fnr, cid)] = len(idx)
        centers = np.array([med[k] for k in keys]) if keys else np.zeros((0, 5), np.float32)
        tr.next(fnr, keys, centers, [cnt[k] for k in keys], lambda k: (med[k], cnt[k]))
    tr.finish()
    return tr


def test_tracker_matches_reference(gold, scene):
    tr = run_tracker(gold, scene[2])
    assert len(tr.tracks) == len(gold['tracks'])
    for t, g in zip(tr.tracks, gold['tracks']):
        assert t.frames == g['frames']
        assert [(bool(p), k[0], k[1]) for p, k in zip(t.prediction, t.source)] == g['entries']
        assert np.allclose(t.kf.x, g['kf_x'], rtol=0, atol=1e-9) and np.allclose(t.kf.P, g['kf_P'], rtol=0, atol=1e-9)
    assert sum(sum(t.prediction) for t in tr.tracks) > 0


def test_greedy_assignment_properties():
    rng = np.random.default_rng(0)
    d, t = rng.uniform(-5, 5, (7, 2)), rng.uniform(-5, 5, (4, 2))
    m, near = tracking.assign_greedy(d, t, 1.0)
    assert len(m) == 4 and len(set(m[:, 0])) == 4 and len(set(m[:, 1])) == 4            # every track is paired, whatever the distance
    dist = np.linalg.norm(d[m[:, 0]] - t[m[:, 1]], axis=1)
    assert np.array_equal(near[m[:, 0]], dist < 1.0) and not near[[i for i in range(7) if i not in m[:, 0]]].any()
    m0, n0 = tracking.assign_greedy(np.zeros((0, 2)), t, 1.0)
    assert m0.shape == (0, 2) and len(n0) == 0
