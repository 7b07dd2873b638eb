"""SURVEY §8f rows N3 (real-data adapters) and N4 (evaluation): vilgod_amd.sequence_datasets / vilgod_amd.evaluation against
tests/golden/dataset_golden.pkl = the reference's WaymoDataset / Argo2Dataset run on the same seeded OpenPCDet-layout trees
(tests/golden/make_golden.py::make_dataset), plus properties of the restated Waymo metric (parity unpinned, see evaluation.py)."""
import os
import pickle

import numpy as np
import pytest

from vilgod_amd import evaluation as ev
from vilgod_amd import fixture_data as fx
from vilgod_amd.sequence_datasets import Argo2Dataset, WaymoDataset

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'dataset_golden.pkl')
CLASSES = ['Vehicle', 'Pedestrian', 'Cyclist']
EVAL_VARIANTS = {          # = make_golden.EVAL_VARIANTS
    'plain': dict(class_agnostic=False, bev=False, moving=False, static=False, score_thresh=0.0, sampling_rate=1),
    'moving': dict(class_agnostic=False, bev=False, moving=True, static=False, score_thresh=0.0, sampling_rate=1),
    'static': dict(class_agnostic=False, bev=False, moving=False, static=True, score_thresh=0.3, sampling_rate=1),
    'agnostic_bev': dict(class_agnostic=True, bev=True, moving=False, static=False, score_thresh=0.0, sampling_rate=2),
}


def _detections(ds, seed=0):       # = make_golden.dataset_detections
    rng = np.random.default_rng(seed)
    out = []
    for f in range(ds.sequence_length):
        a = ds.get_annos(f)
        b = np.asarray(a['gt_boxes'], np.float64)[:, :7].copy()
        keep = rng.uniform(size=len(b)) < 0.8
        b = b[keep] + rng.normal(0, 0.05, size=(int(keep.sum()), 7))
        names = np.asarray(a['gt_names'])[keep]
        extra = np.c_[rng.uniform(-45, 45, 3), rng.uniform(-18, 18, 3), rng.uniform(0.5, 1.5, 3), rng.uniform(0.5, 5, (3, 3)), rng.uniform(-3, 3, 3)]
        out.append({'boxes_lidar': np.concatenate([b, extra]), 'name': np.concatenate([names, np.array(['Vehicle', 'Pedestrian', 'Cyclist'])]),
                    'score': rng.uniform(0.05, 1.0, size=len(b) + 3), 'moving': rng.uniform(size=len(b) + 3) < 0.5})
    return out


def _same(a, b, tol=0.0):
    if isinstance(a, dict):
        assert set(a) == set(b), (set(a), set(b))
        for k in a:
            _same(a[k], b[k], tol)
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            _same(x, y, tol)
    elif isinstance(a, np.ndarray) and a.dtype.kind in 'fc':
        assert a.shape == np.shape(b)
        np.testing.assert_allclose(a, b, rtol=0, atol=tol)
    elif isinstance(a, np.ndarray):
        assert a.shape == np.shape(b) and (a == np.asarray(b)).all(), (a, b)
    else:
        assert a == b, (a, b)


@pytest.mark.parametrize('kind', ['waymo', 'argo2'])
def test_adapter_matches_reference(kind, tmp_path):
    gold = pickle.load(open(GOLD, 'rb'))[kind]
    Cls, cfg, writer = {'waymo': (WaymoDataset, fx.WAYMO_CFG, fx.write_waymo), 'argo2': (Argo2Dataset, fx.ARGO2_CFG, fx.write_argo2)}[kind]
    writer(str(tmp_path), n_sequences=3, n_frames=6, n_points=3000, n_objects=10, seed=5)
    ds = Cls(dict(cfg, DATA_PATH=str(tmp_path)), CLASSES, training=True, start_sequence=0, end_sequence=2)
    ds.training = False
    _same(gold['mapping'], ds.sequence_mapping)
    assert (gold['start'], gold['end'], gold['names']) == (ds.start_sequence, ds.end_sequence, list(ds.sequence_names))
    for g, name in zip(gold['sequences'], ds.next_sequence()):
        assert g['name'] == name and g['indices'] == ds.sequence_indices
        assert g['moving_ids'] == sorted(ds._moving_track_ids)
        _same(g['annos'], [ds.get_annos(f) for f in range(ds.sequence_length)])
        T = np.linalg.inv(ds.sequence_infos[0]['pose']) @ ds.sequence_infos[1]['pose']
        _same(g['annos_T'], ds.get_annos(1, transformation=T), tol=1e-12)
        _same(g['poses'], [np.array(i['pose']) for i in ds.sequence_infos])
        _same(g['points0'], ds.get_lidar_points(0))
        _same(g['points1_T'], ds.get_lidar_points(1, transformation=ds.sequence_infos[1]['pose'])[:50], tol=1e-12)
        dets = _detections(ds, seed=11)
        for vname, kw in list(EVAL_VARIANTS.items()) + [('sequence', dict(EVAL_VARIANTS['plain'], sequence=True))]:
            want = g['eval'][vname]
            d, t = ev.filter_for_evaluation(ds, dets, CLASSES, indices=ds.sequence_indices, eval_range=[-50., -20., 50., 20.], **kw)
            pd_ = ev.waymo_type_results(d, CLASSES, is_gt=False)
            gt_ = ev.waymo_type_results(t, CLASSES, is_gt=True, fake_gt_infos=False)
            pd_m = ev.mask_by_distance(1000, pd_[1], pd_[0], pd_[2], pd_[3], pd_[4])
            gt_m = ev.mask_by_distance(1000, gt_[1], gt_[0], gt_[2], gt_[3], gt_[5])
            _same(want['pd'], list(pd_m), tol=1e-12)
            _same(want['gt'], list(gt_m), tol=1e-12)


def test_iou_against_clipping():
    """vectorised rotated IoU vs a scalar Sutherland-Hodgman clip."""
    rng = np.random.default_rng(3)
    A = np.c_[rng.uniform(-3, 3, (40, 3)), rng.uniform(0.5, 5, (40, 3)), rng.uniform(-4, 4, 40)]
    B = np.c_[rng.uniform(-3, 3, (40, 3)), rng.uniform(0.5, 5, (40, 3)), rng.uniform(-4, 4, 40)]
    B[:5] = A[:5]                                            # identical boxes -> 1
    B[5, :] = A[5, :]; B[5, 6] += np.pi                       # half-turn -> 1
    M = ev.iou3d_matrix(A, B)

    def poly(b):
        c, s = np.cos(b[6]), np.sin(b[6])
        loc = np.array([[b[3] / 2, b[4] / 2], [-b[3] / 2, b[4] / 2], [-b[3] / 2, -b[4] / 2], [b[3] / 2, -b[4] / 2]])
        return loc @ np.array([[c, s], [-s, c]]) + b[:2]

    def clip(pa, pb):
        out = [tuple(p) for p in pa]
        for i in range(4):
            a, b = pb[i], pb[(i + 1) % 4]
            inp, out = out, []
            side = lambda q: (b[0] - a[0]) * (q[1] - a[1]) - (b[1] - a[1]) * (q[0] - a[0])
            for j in range(len(inp)):
                cur, prv = inp[j], inp[j - 1]
                sc, sp = side(cur), side(prv)
                if (sc >= 0) != (sp >= 0):
                    t = sp / (sp - sc)
                    out.append((prv[0] + t * (cur[0] - prv[0]), prv[1] + t * (cur[1] - prv[1])))
                if sc >= 0:
                    out.append(cur)
        if len(out) < 3:
            return 0.0
        x, y = np.array(out).T
        return 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))
    for i in range(40):
        for j in range(40):
            zo = min(A[i, 2] + A[i, 5] / 2, B[j, 2] + B[j, 5] / 2) - max(A[i, 2] - A[i, 5] / 2, B[j, 2] - B[j, 5] / 2)
            inter = clip(poly(A[i]), poly(B[j])) * max(zo, 0)
            want = inter / (np.prod(A[i, 3:6]) + np.prod(B[j, 3:6]) - inter)
            assert abs(M[i, j] - want) < 1e-9, (i, j, M[i, j], want)
    np.testing.assert_allclose(np.diag(M)[:6], 1.0, atol=1e-9)


def _arrays(frames):
    f = np.concatenate([np.full(len(b), i) for i, (b, *_) in enumerate(frames)])
    return (f, np.concatenate([x[0] for x in frames]), np.concatenate([x[1] for x in frames]), np.concatenate([x[2] for x in frames]))


def test_metric_properties():
    cfg = ev.build_config(difficulties=[1, 2], breakdown_range=True, iou_thresholds=[0.4, 0.4, 0.4, 0.4])
    assert cfg['score_cutoffs'][0] == 0.0 and cfg['score_cutoffs'][-1] == 1.0 and len(cfg['score_cutoffs']) == 101
    rng = np.random.default_rng(0)
    n = 12
    gt_box = np.c_[rng.uniform(-40, 40, (n, 2)), np.ones(n), np.tile([4.5, 1.9, 1.6], (n, 1)), rng.uniform(-3, 3, n)]
    gt_type = np.ones(n, int); gt_frame = np.repeat(np.arange(3), 4); gt_diff = np.r_[np.ones(8, int), 2 * np.ones(4, int)]
    # perfect detections -> AP = APH = 1 at both levels
    out = ev.detection_metrics(gt_frame, gt_box, gt_type, rng.uniform(0.3, 0.9, n), gt_frame, gt_box, gt_type, gt_diff, cfg)
    for lvl in (1, 2):
        assert abs(out[f'OBJECT_TYPE_TYPE_VEHICLE_LEVEL_{lvl}/AP'][0] - 1) < 1e-9 and abs(out[f'OBJECT_TYPE_TYPE_VEHICLE_LEVEL_{lvl}/APH'][0] - 1) < 1e-9
    assert out['OBJECT_TYPE_TYPE_PEDESTRIAN_LEVEL_2/AP'][0] == 0
    # heading flipped by pi: AP 1, APH 0
    flipped = gt_box.copy(); flipped[:, 6] += np.pi
    out = ev.detection_metrics(gt_frame, flipped, gt_type, rng.uniform(0.3, 0.9, n), gt_frame, gt_box, gt_type, gt_diff, cfg)
    assert abs(out['OBJECT_TYPE_TYPE_VEHICLE_LEVEL_2/AP'][0] - 1) < 1e-9 and out['OBJECT_TYPE_TYPE_VEHICLE_LEVEL_2/APH'][0] < 1e-9
    # wrong class or far away: nothing matches
    out = ev.detection_metrics(gt_frame, gt_box + np.r_[30, 30, 0, 0, 0, 0, 0], gt_type, np.full(n, 0.5), gt_frame, gt_box, gt_type, gt_diff, cfg)
    assert out['OBJECT_TYPE_TYPE_VEHICLE_LEVEL_2/AP'][0] == 0
    # a hand-computed curve: 4 ground truths in one frame; detections by falling score: hit, miss, hit, (two never found)
    g = gt_box[:4]; fr = np.zeros(4, int)
    pd_box = np.concatenate([g[:1], g[:1] + np.r_[60, 0, 0, 0, 0, 0, 0], g[1:2]])
    out = ev.detection_metrics(np.zeros(3, int), pd_box, np.ones(3, int), np.array([0.9, 0.6, 0.3]), fr, g, np.ones(4, int), np.ones(4, int),
                               ev.build_config(difficulties=[1], breakdown_range=False, iou_thresholds=[0.4] * 4))
    # recall 0.25 at precision 1; recall 0.5 at precision 2/3 -> area 0.25 * 1 + 0.25 * 2/3
    assert abs(out['OBJECT_TYPE_TYPE_VEHICLE_LEVEL_1/AP'][0] - (0.25 + 0.25 * 2 / 3)) < 1e-9
    # level-2-only ground truth matched at level 1 is neither TP nor FP: level-1 AP stays 1 with level-1 recall complete
    out = ev.detection_metrics(gt_frame, gt_box, gt_type, np.full(n, 0.5), gt_frame, gt_box, gt_type, gt_diff,
                               ev.build_config(difficulties=[1, 2], breakdown_range=False))
    assert abs(out['OBJECT_TYPE_TYPE_VEHICLE_LEVEL_1/AP'][0] - 1) < 1e-9
    lines = ev.eval_log_lines(out)
    assert lines[0].startswith('Vehicle AP  L1: 100.00')


def test_evaluation_end_to_end(tmp_path):
    """adapter.evaluation on detections = ground truth: AP 1 for every class present, through the CLI's call signature."""
    fx.write_waymo(str(tmp_path), n_sequences=1, n_frames=5, n_points=2000, n_objects=12, seed=2)
    ds = WaymoDataset(dict(fx.WAYMO_CFG, DATA_PATH=str(tmp_path)), CLASSES, training=True, start_sequence=0, end_sequence=1)
    ds.training = False
    dets, idx = [], []
    for _ in ds.next_sequence():
        for f in range(ds.sequence_length):
            a = ds.get_annos(f)
            dets.append({'boxes_lidar': np.asarray(a['gt_boxes'], np.float64)[:, :7], 'name': np.asarray(a['gt_names']),
                         'score': np.full(len(a['gt_names']), 0.7), 'moving': np.asarray(a['moving'])})
        idx += ds.sequence_indices
    ap = ds.evaluation(dets, CLASSES, indices=idx, eval_cfg=dict(difficulties=[2], breakdown_range=False, iou_thresholds=[0.4] * 4),
                       class_agnostic=False, eval_range=[-75., -75., 75., 75.], bev=False, moving=False, static=False, score_thresh=0.0,
                       sampling_rate=1)
    present = {n for d in dets for n in d['name']}
    assert present
    for c in present:
        assert abs(ap[f'OBJECT_TYPE_TYPE_{c.upper()}_LEVEL_2/AP'][0] - 1) < 1e-9


def test_adapter_split_interval_and_sequence_window(tmp_path):
    """set_split re-reads the infos of the other split; SAMPLED_INTERVAL thins the frames; start/end_sequence pick the sequences
    (waymo_dataset.py:57-86; OpenPCDet's ImageSets/<split>.txt + SAMPLED_INTERVAL)."""
    root = str(tmp_path)
    fx.write_waymo(root, n_sequences=3, n_frames=6, n_points=500, n_objects=6, seed=1, split='train')
    names_val = fx.write_waymo(root, n_sequences=1, n_frames=4, n_points=500, n_objects=6, seed=9, split='val')
    ds = WaymoDataset(dict(fx.WAYMO_CFG, DATA_PATH=root), CLASSES, training=True, start_sequence=1, end_sequence=3)
    assert len(ds.infos) == 18 and len(ds.sequence_mapping) == 3
    assert len(ds.sequence_names) == 2 and ds.sequence_names == list(ds.sequence_mapping)[1:3]
    ds.set_split('val')
    assert len(ds.infos) == 4 and list(ds.sequence_mapping) == names_val
    assert ds.start_sequence == 0 and ds.end_sequence == 1          # 1 < 1 is false -> start falls back to 0 (create_sequence_mapping)
    seqs = list(ds.next_sequence())
    assert seqs == names_val and ds.sequence_length == 4 and ds.get_lidar_points(0).shape[1] == 5
    thin = WaymoDataset(dict(fx.WAYMO_CFG, DATA_PATH=root, SAMPLED_INTERVAL={'train': 2, 'test': 1}), CLASSES, training=True)
    assert len(thin.infos) == 9
    # NLZ rows are dropped unless DISABLE_NLZ_FLAG_ON_POINTS; intensity goes through tanh
    keep = WaymoDataset(dict(fx.WAYMO_CFG, DATA_PATH=root, DISABLE_NLZ_FLAG_ON_POINTS=False), CLASSES, training=True)
    next(iter(ds.next_sequence())); next(iter(keep.next_sequence()))
    full = WaymoDataset(dict(fx.WAYMO_CFG, DATA_PATH=root), CLASSES, training=True)
    next(iter(full.next_sequence()))
    a, b = full.get_lidar_points(0), keep.get_lidar_points(0)
    assert len(b) < len(a) and 0.9 < len(b) / len(a) < 1.0 and (a[:, 3] <= 1.0).all() and (a[:, 3] >= 0.0).all()
