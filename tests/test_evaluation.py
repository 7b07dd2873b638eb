"""the restated Waymo detection metric (vilgod_amd/evaluation.py) against HAND-COMPUTED AP / APH values
(tests/golden/eval_golden.json: every case carries its derivation).  The metric library itself (`waymo_open_dataset`, a TensorFlow
custom op) is not installable here, so this pins the restatement to the published definition case by case: perfect detections,
a yaw offset, a false positive above / between the true positives, a false negative, the IoU threshold from both sides, the
Hungarian matcher (waymo_eval.py:111) against a greedy one, level-2 ground truth, and the level assignment from point counts
(waymo_eval.py:44-49)."""
import json

import numpy as np
import pytest

from conftest import GOLDEN
from vilgod_amd import evaluation as ev

G = json.load(open(f'{GOLDEN}/eval_golden.json'))


def _cfg():
    return ev.build_config(difficulties=[1, 2], breakdown_range=False, iou_thresholds=[G['iou_threshold']] * 4)


def _check(out, case):
    tol = case.get('tol', 1e-9)
    for k, want in case['expect'].items():
        got = out[f'OBJECT_TYPE_TYPE_VEHICLE_LEVEL_{k}'][0]
        assert abs(got - want) <= tol, (case['name'], k, got, want)
    for t in ('PEDESTRIAN', 'CYCLIST', 'SIGN'):
        assert out[f'OBJECT_TYPE_TYPE_{t}_LEVEL_2/AP'][0] == 0


@pytest.mark.parametrize('case', G['cases'], ids=[c['name'] for c in G['cases']])
def test_metric_reproduces_hand_computed_case(case):
    gt, pd = np.array(case['gt'], np.float64), np.array(case['pd'], np.float64)
    out = ev.detection_metrics(np.zeros(len(pd), np.int64), pd, np.ones(len(pd), int), np.array(case['score']),
                               np.zeros(len(gt), np.int64), gt, np.ones(len(gt), int), np.array(case['gt_level'], np.int8), _cfg())
    _check(out, case)
    # the same objects spread over several frames (one ground truth per frame where possible) give the same curve only when the
    # detections travel with their ground truth; a permutation of the input order must never matter
    rng = np.random.default_rng(0)
    pp, gp = rng.permutation(len(pd)), rng.permutation(len(gt))
    out2 = ev.detection_metrics(np.zeros(len(pd), np.int64), pd[pp], np.ones(len(pd), int), np.array(case['score'])[pp],
                                np.zeros(len(gt), np.int64), gt[gp], np.ones(len(gt), int), np.array(case['gt_level'], np.int8)[gp], _cfg())
    _check(out2, case)


@pytest.mark.parametrize('case', G['through_infos'], ids=[c['name'] for c in G['through_infos']])
def test_waymo_evaluation_levels_from_point_counts(case):
    gt = dict(name=np.array(['Vehicle'] * len(case['gt'])), gt_boxes_lidar=np.array(case['gt'], np.float32),
              num_points_in_gt=np.array(case['num_points_in_gt']), difficulty=np.zeros(len(case['gt']), np.int32))
    pd = dict(name=np.array(['Vehicle'] * len(case['pd'])), boxes_lidar=np.array(case['pd'], np.float64), score=np.array(case['score']))
    out = ev.waymo_evaluation([pd], [gt], class_name=['Vehicle', 'Pedestrian', 'Cyclist'], distance_thresh=1000, fake_gt_infos=False,
                              cfg=dict(difficulties=[1, 2], breakdown_range=False, iou_thresholds=[G['iou_threshold']] * 4))
    _check(out, case)


def test_iou_of_the_golden_geometry():
    """The IoU values the derivations quote, from the evaluation's own polygon clipping."""
    B = lambda x, l=4.0, w=2.0, yaw=0.0: np.array([[x, 0, 0, l, w, 1.5, yaw]])
    assert abs(ev.iou3d_matrix(B(1.70), B(0))[0, 0] - 2.30 / 5.70) < 1e-12
    assert abs(ev.iou3d_matrix(B(1.73), B(0))[0, 0] - 2.27 / 5.73) < 1e-12
    assert abs(ev.iou3d_matrix(B(2.1, l=6.0), B(0))[0, 0] - 2.9 / 7.1) < 1e-12
    assert abs(ev.iou3d_matrix(B(2.1, l=6.0), B(4))[0, 0] - 3.1 / 6.9) < 1e-12
    a = 2 / (1 + np.sin(0.3) + np.cos(0.3))
    assert abs(ev.iou3d_matrix(B(0, l=1, w=1, yaw=0.3), B(0, l=1, w=1))[0, 0] - a / (2 - a)) < 1e-12


# ---- a second, independent check: brute force over ALL assignments on small random scenes -----------------------
def _aabb_iou3d(a, b):
    """IoU of two boxes whose yaw is a multiple of pi/2 (axis-aligned footprints), written from the definition -- no polygon clipping."""
    def ext(x):
        l, w = (x[3], x[4]) if abs(np.cos(x[6])) > 0.5 else (x[4], x[3])
        return x[0] - l / 2, x[0] + l / 2, x[1] - w / 2, x[1] + w / 2, x[2] - x[5] / 2, x[2] + x[5] / 2
    ax0, ax1, ay0, ay1, az0, az1 = ext(a)
    bx0, bx1, by0, by1, bz0, bz1 = ext(b)
    ix, iy, iz = min(ax1, bx1) - max(ax0, bx0), min(ay1, by1) - max(ay0, by0), min(az1, bz1) - max(az0, bz0)
    if ix <= 0 or iy <= 0 or iz <= 0:
        return 0.0
    inter = ix * iy * iz
    return inter / (a[3] * a[4] * a[5] + b[3] * b[4] * b[5] - inter)


def _brute_force_ap(pd, score, gt, gt_level, thr, level, cutoffs):
    """AP / APH of one type in one frame: at every score cutoff the assignment of kept predictions to ground truth that maximises the
    summed IoU of the pairs at or above the threshold is found by ENUMERATING every injective partial assignment (the metric uses the
    Hungarian algorithm, waymo_eval.py:111); precision / recall / heading-weighted versions from the counts; the area under the
    precision envelope by an explicit loop."""
    import itertools
    P, G = len(pd), len(gt)
    iou = np.array([[_aabb_iou3d(p, g) for g in gt] for p in pd]).reshape(P, G)
    w = np.where((iou >= thr) & (iou > 0), iou, 0.0)
    prec, rec, prec_h, rec_h = [], [], [], []
    for cut in cutoffs:
        keep = [i for i in range(P) if score[i] >= cut]
        best, best_pairs = -1.0, []
        # every way to give each kept prediction a distinct ground truth or none
        for assign in itertools.product(range(-1, G), repeat=len(keep)):
            used = [g for g in assign if g >= 0]
            if len(used) != len(set(used)):
                continue
            pairs = [(keep[k], g) for k, g in enumerate(assign) if g >= 0 and w[keep[k], g] > 0]
            tot = sum(w[p, g] for p, g in pairs)
            if tot > best + 1e-12:
                best, best_pairs = tot, pairs
        matched_g = {g for _, g in best_pairs}
        tp = sum(1 for _, g in best_pairs if gt_level[g] <= level)
        ha = 0.0
        for p, g in best_pairs:
            if gt_level[g] <= level:
                d = abs(pd[p][6] - gt[g][6]) % (2 * np.pi)
                ha += 1.0 - min(d, 2 * np.pi - d) / np.pi
        fp = len(keep) - len(best_pairs)
        fn = sum(1 for g in range(G) if g not in matched_g and gt_level[g] <= level)
        prec.append(tp / (tp + fp) if tp + fp else 0.0); rec.append(tp / (tp + fn) if tp + fn else 0.0)
        prec_h.append(ha / (tp + fp) if tp + fp else 0.0); rec_h.append(ha / (tp + fn) if tp + fn else 0.0)

    def area(p, r):
        pts = sorted(zip(r, p), key=lambda t: -t[0])             # recall falling
        a, best_p = 0.0, 0.0
        for i, (ri, pi_) in enumerate(pts):
            best_p = max(best_p, pi_)                            # best precision at any recall >= ri
            r_next = pts[i + 1][0] if i + 1 < len(pts) else 0.0
            a += (ri - r_next) * best_p
        return a
    return area(prec, rec), area(prec_h, rec_h)


@pytest.mark.parametrize('seed', range(40))
def test_metric_equals_brute_force_over_all_assignments(seed):
    """Random scenes of <= 5 predictions and <= 5 ground-truth vehicles with axis-aligned footprints (yaw in {0, pi/2, pi}), clustered so
    that predictions compete for ground truth: detection_metrics (Hungarian matcher, polygon-clipping IoU) against an exhaustive
    matcher with an analytic IoU.  Wherever the optimal assignment is unique the two agree to rounding; scenes with an exact tie between
    two optimal assignments of different true-positive sets are re-drawn (probability ~0 with continuous random offsets)."""
    rng = np.random.default_rng(1000 + seed)
    G_, P_ = int(rng.integers(1, 6)), int(rng.integers(1, 6))
    yaws = np.array([0.0, np.pi / 2, np.pi])
    gt = np.c_[rng.uniform(-3, 3, G_), rng.uniform(-2, 2, G_), np.zeros(G_), rng.uniform(3.5, 5, G_), rng.uniform(1.6, 2.2, G_),
               rng.uniform(1.4, 1.8, G_), yaws[rng.integers(0, 3, G_)]]
    src = rng.integers(0, G_, P_)
    pd = gt[src] + np.c_[rng.normal(0, 0.6, P_), rng.normal(0, 0.3, P_), np.zeros(P_), rng.normal(0, 0.2, P_), rng.normal(0, 0.1, P_),
                         np.zeros(P_), np.zeros(P_)]
    pd[:, 6] = yaws[rng.integers(0, 3, P_)]
    score = np.round(rng.uniform(0.05, 0.99, P_), 3)
    level = rng.integers(1, 3, G_).astype(np.int8)
    thr = float(rng.choice([0.3, 0.5, 0.7]))
    cfg = ev.build_config(difficulties=[1, 2], breakdown_range=False, iou_thresholds=[thr] * 4)
    out = ev.detection_metrics(np.zeros(P_, np.int64), pd, np.ones(P_, int), score, np.zeros(G_, np.int64), gt, np.ones(G_, int), level, cfg)
    # the analytic IoU agrees with the polygon clipper on these boxes
    got_iou = ev.iou3d_matrix(pd, gt)
    want_iou = np.array([[_aabb_iou3d(p, g) for g in gt] for p in pd])
    assert np.abs(got_iou - want_iou).max() < 1e-9
    for lvl in (1, 2):
        ap, aph = _brute_force_ap(pd, score, gt, level, thr, lvl, cfg['score_cutoffs'])
        assert abs(out[f'OBJECT_TYPE_TYPE_VEHICLE_LEVEL_{lvl}/AP'][0] - ap) < 1e-9, (seed, lvl)
        assert abs(out[f'OBJECT_TYPE_TYPE_VEHICLE_LEVEL_{lvl}/APH'][0] - aph) < 1e-9, (seed, lvl)
