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


def test_array_form_of_the_track_boxes_equals_the_line_by_line_form():
    """vilgod_amd/tracking.py evaluates the motion vectors and the motion-aligned boxes of a track with array operations; the
    line-by-line restatement of zero_shot_detector.py:491-659 (oracle/tracking_oracle.py, the form pinned against the reference's
    own run by track_golden.pkl) must come out bit for bit: directions (values and dtype) and boxes, on 200 seeded tracks that
    cover standing, creeping, walking and driving objects, short tracks, and both sources of the cluster medians."""
    from oracle import tracking_oracle as to
    rng = np.random.default_rng(0)
    n_vec = n_box = 0
    for trial in range(200):
        n = int(rng.integers(1, 40))
        speed = rng.choice([0.0, 0.02, 0.2, 0.6, 1.5])
        head = rng.uniform(0, 6.28)
        c = np.cumsum(np.c_[np.cos(head + rng.normal(0, 0.3, n)), np.sin(head + rng.normal(0, 0.3, n))] * speed * rng.uniform(0.5, 1.5, (n, 1)), axis=0)
        c = (c + rng.normal(0, 0.05, (n, 2)) + rng.uniform(-50, 50, 2)).astype(np.float32)
        a, b = to.motion_vectors(c), tracking.motion_vectors(c)
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert x.dtype == y.dtype and np.array_equal(x, y), trial
        if not a:
            continue
        n_vec += 1
        pts = [(rng.normal(size=(int(rng.integers(12, 300)), 5)) * [1.5, 0.7, 0.6, 1, 1] + [c[i, 0], c[i, 1], 0.8, 0, 0]).astype(np.float32) for i in range(n)]
        T = [np.linalg.inv(p) for p in synthetic.make_poses(n, seed=trial)]
        for c3 in (None, [np.median(p[:, :3], axis=0) for p in pts]):
            assert np.array_equal(to.moving_boxes(pts, a, T, centers3=c3), tracking.moving_boxes(pts, a, T, centers3=c3)), trial
            n_box += 1
    assert n_vec > 50 and n_box > 100


def build_table(gold, X):
    tab = tracking.DetectionTable()
    pts, stat = {}, {}
    for fnr, (dets, flags) in enumerate(zip(gold['clusters'], gold['static'])):
        for (cid, idx), st in zip(dets, flags):
            pts[(fnr, cid)] = X[fnr][idx]
            stat[(fnr, cid)] = st
            tab.valid[(fnr, cid)] = True
    return tab, pts, stat


def entry_boxes(tab, t):
    return np.array([t.clone_box[i] if t.prediction[i] else tab.box[t.source[i]] for i in range(len(t))], dtype=np.float64)


def test_track_boxes_match_reference(gold, scene):
    """The track branch of fit_bounding_boxes_simple with the REFERENCE's rectangle routine plugged in (the product plugs in the
    GPU kernel's all-edges rectangle): motion vectors, motion-aligned boxes, top-3 median size, closest-corner shift."""
    frames, poses, X = scene
    tr = run_tracker(gold, X)
    tab, pts, stat = build_table(gold, X)
    to_ego = lambda f: np.linalg.inv(poses[f]) @ poses[0]
    tracking.fit_track_boxes(tr, tab, pts.__getitem__, stat.__getitem__, to_ego, lambda xy: so.minimum_bounding_rectangle(xy, all_edges=False))
    n_moving = 0
    for t, g in zip(tr.tracks, gold['tracks']):
        assert t.static == g['track_static_fit']
        n_moving += not t.static
        got = entry_boxes(tab, t)
        assert got.shape == g['boxes_fit'].shape
        # float32 medians / arctan2 upstream: agreement to ~1e-5 (identical code path, numpy's float32 arctan2 is <= 1 ulp)
        assert np.allclose(got, g['boxes_fit'], rtol=0, atol=2e-5), np.abs(got - g['boxes_fit']).max()
        flags = [t.clone_static_track[i] if t.prediction[i] else tab.static_track.get(t.source[i]) for i in range(len(t))]
        assert flags == g['static_track_fit']
    assert n_moving > 0


def test_propagate_labels_matches_reference(gold, scene):
    frames, poses, X = scene
    tr = run_tracker(gold, X)
    tab, pts, stat = build_table(gold, X)
    to_ego = lambda f: np.linalg.inv(poses[f]) @ poses[0]
    tracking.fit_track_boxes(tr, tab, pts.__getitem__, stat.__getitem__, to_ego, lambda xy: so.minimum_bounding_rectangle(xy, all_edges=False))
    for fnr, (dets, cls) in enumerate(zip(gold['clusters'], gold['cls_in'])):
        for (cid, _), (name, score) in zip(dets, cls):
            tab.name[(fnr, cid)], tab.score[(fnr, cid)] = name, np.float32(score)
    tracking.propagate_labels(tr, tab, lambda k: len(pts[k]), ['Vehicle', 'Pedestrian', 'Cyclist'], min_length=5)
    for t, g in zip(tr.tracks, gold['tracks']):
        assert (t.static, t.valid, t.class_label, t.class_label_corrected, t.class_label_corrected_by_size) == \
               (g['track_static'], g['track_valid'], g['class_label'], g['corrected'], g['corrected_by_size'])
        assert np.allclose(entry_boxes(tab, t), g['boxes_final'], rtol=0, atol=2e-5)
    for fnr, (dets, fin) in enumerate(zip(gold['clusters'], gold['final'])):
        for (cid, _), f in zip(dets, fin):
            k = (fnr, cid)
            assert tab.valid[k] == f['valid'] and tab.name[k] == f['name'] and abs(float(tab.score[k]) - f['score']) < 1e-6
            assert tab.static_track.get(k) == f['static_track']
            assert (f['box'] is None and k not in tab.box) or np.allclose(tab.box[k], f['box'], rtol=0, atol=2e-5)


def test_small_pieces():
    assert tracking.size_prior_class(np.array([0, 0, 0, 0.8, 0.7, 1.7, 0])) == 'Pedestrian'
    assert tracking.size_prior_class(np.array([0, 0, 0, 1.8, 0.7, 1.7, 0])) == 'Cyclist'
    assert tracking.size_prior_class(np.array([0, 0, 0, 4.5, 1.9, 1.6, 0])) == 'Vehicle'
    assert tracking.size_prior_class(np.array([0, 0, 0, 12.0, 3.5, 4.0, 0])) == 'Background'
    a = np.array([0, 0, 0, 4, 2, 1, 0.0]); b = np.array([2.5, 0, 0, 4, 2, 1, np.pi / 2]); c = np.array([4.1, 0, 0, 4, 2, 1, np.pi / 2])
    assert tracking.rectangles_overlap(a, b) and not tracking.rectangles_overlap(a, c)
    assert not tracking.rectangles_overlap(a, np.array([0, 0, 5, 4, 2, 1, 0.0]))          # apart in z
    ang = tracking.dominant_angles([0.10, 0.11, 0.12 + np.pi, 1.5, 3.0 + 2 * np.pi])
    assert len(ang) == 3 and abs(np.mean(ang) - 0.11) < 1e-9
    assert tracking.motion_vectors(np.zeros((6, 2), np.float32)) == []                        # a cluster that never moves: static path


@pytest.mark.gpu
def test_hip_tracked_stages_match_reference(gold, scene):
    """N2 through the PRODUCT's GPU pieces: the cluster medians the tracker associates on come from vg_cluster_medians, the static
    boxes of tracked clusters from PseudoLabelPipeline.fit_boxes (box_mode='reference': z extent kernel + qhull vertex order in a
    helper process), then tracking.fit_track_boxes / propagate_labels as zero_shot_detector wires them -- against the vectors the
    reference's own Tracker / Track / fit_bounding_boxes_simple / propagate_labels produced (track_golden.pkl)."""
    import torch
    from vilgod_amd.pipeline import PseudoLabelPipeline
    frames, poses, X = scene
    cuda = torch.device('cuda:0')
    pipe = PseudoLabelPipeline(device=cuda, max_points=max(len(x) for x in X) + 16, clip_model_path='/nonexistent', box_mode='reference')
    med, cnt, sbox, pts, stat = {}, {}, {}, {}, {}
    tab = tracking.DetectionTable()
    tr = tracking.Tracker(mode='cluster_center', max_distance=1.0, min_length=5, max_missed=3)
    for fnr, (dets, flags) in enumerate(zip(gold['clusters'], gold['static'])):
        keys = [(fnr, cid) for cid, _ in dets]
        if dets:
            index = np.concatenate([np.asarray(idx) for _, idx in dets]).astype(np.int32)
            seg = np.r_[0, np.cumsum([len(idx) for _, idx in dets])].astype(np.int32)
            d_X = torch.from_numpy(X[fnr]).to(cuda)
            m = pipe.cluster_medians(d_X, torch.from_numpy(index).to(cuda), torch.from_numpy(seg).to(cuda)).cpu().numpy()
            boxes = pipe.fit_boxes(d_X, index, seg)
            for j, ((cid, idx), st) in enumerate(zip(dets, flags)):
                k = (fnr, cid)
                assert np.array_equal(m[j], np.median(X[fnr][idx], axis=0))
                med[k], cnt[k], sbox[k], pts[k], stat[k] = m[j], len(idx), boxes[j], X[fnr][idx], st
                tab.valid[k] = True
        centers = np.array([med[k] for k in keys]) if keys else np.zeros((0, 5), np.float32)
        tr.next(fnr, keys, centers, [cnt[k] for k in keys], lambda k: (med[k], cnt[k]))
    tr.finish()
    assert len(tr.tracks) == len(gold['tracks'])
    for t, g in zip(tr.tracks, gold['tracks']):
        assert t.frames == g['frames'] and [(bool(p), k[0], k[1]) for p, k in zip(t.prediction, t.source)] == g['entries']
    to_ego = lambda f: np.linalg.inv(poses[f]) @ poses[0]
    tracking.fit_track_boxes(tr, tab, pts.__getitem__, stat.__getitem__, to_ego, static_box_of=sbox.__getitem__, median_of=med.__getitem__)
    for t, g in zip(tr.tracks, gold['tracks']):
        assert t.static == g['track_static_fit']
        assert np.allclose(entry_boxes(tab, t), g['boxes_fit'], rtol=0, atol=2e-5)
    for fnr, (dets, cls) in enumerate(zip(gold['clusters'], gold['cls_in'])):
        for (cid, _), (name, score) in zip(dets, cls):
            tab.name[(fnr, cid)], tab.score[(fnr, cid)] = name, np.float32(score)
    tracking.propagate_labels(tr, tab, lambda k: cnt[k], ['Vehicle', 'Pedestrian', 'Cyclist'], min_length=5)
    for t, g in zip(tr.tracks, gold['tracks']):
        assert (t.static, t.valid, t.class_label) == (g['track_static'], g['track_valid'], g['class_label'])
        assert np.allclose(entry_boxes(tab, t), g['boxes_final'], rtol=0, atol=2e-5)
    for fnr, (dets, fin) in enumerate(zip(gold['clusters'], gold['final'])):
        for (cid, _), f in zip(dets, fin):
            k = (fnr, cid)
            assert tab.valid[k] == f['valid'] and tab.name[k] == f['name'] and tab.static_track.get(k) == f['static_track']
            assert (f['box'] is None and k not in tab.box) or np.allclose(tab.box[k], f['box'], rtol=0, atol=2e-5)
