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
