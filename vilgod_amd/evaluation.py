"""Detection evaluation -- SURVEY §8f row N4: the range / moving / static / class-agnostic filters of the dataset adapters'
`evaluation`, the OpenPCDet -> Waymo array conversion, and a restatement of the Waymo Open Dataset detection metric (AP / APH per
object type and difficulty level) that upstream reaches through TensorFlow.

Mirrors
  WaymoDataset.evaluation / Argo2Dataset.evaluation            src/datasets/waymo_dataset.py:202-329, argo2_dataset.py:217-377
  OpenPCDetWaymoDetectionMetricsEstimator                      src/datasets/waymo_eval.py:27-231
      generate_waymo_type_results :30-92, build_config :94-124, mask_by_distance :187-195, waymo_evaluation :197-231
  print_eval_log                                               src/utils/eval_utils.py:132-140

PARITY.  The filters, the array conversion and the config are the reference's own Python and are pinned against it
(tests/golden/make_golden.py::make_dataset).  The metric itself lives in `waymo_open_dataset.metrics` (a TensorFlow custom op
around the C++ library; neither package is installed, README.md:83) -- **parity unpinned**; what is restated from the published
library:
  * boxes are 7-DOF, IoU = BEV polygon intersection x z overlap / union volume (Label.Box.TYPE_3D);
  * per frame and score cutoff (the 101 cutoffs of build_config) the predictions with score >= cutoff and the ground truths of
    one object type are matched one to one by the Hungarian method maximising the total IoU, a pair being admissible when its
    IoU reaches the type's threshold;
  * at difficulty level L a matched prediction is a true positive when its ground truth has level <= L (contributing its
    heading accuracy 1 - min(|d|, 2 pi - |d|) / pi to APH) and is ignored otherwise; an unmatched prediction is a false
    positive; an unmatched ground truth of level <= L is a false negative;
  * AP / APH = area under the precision (heading-weighted precision) envelope over recall.
  NOT reproduced: the library's extra samples on sparse stretches of the curve (`desired_recall_delta`), its integer
  quantisation of the IoU weights, and tie-breaking among equally good assignments.  The RANGE breakdown assigns predictions
  and ground truths to [0,30) / [30,50) / [50,inf) by the distance of their own box centre.
Host logic on small data (hundreds of boxes per frame), like upstream's CPU op.
"""
from copy import deepcopy

import numpy as np
from scipy.optimize import linear_sum_assignment

from .sequence_datasets import boxes_to_corners_3d, boxes3d_kitti_fakelidar_to_lidar, drop_info_with_name, _get

WAYMO_CLASSES = ['unknown', 'Vehicle', 'Pedestrian', 'Sign', 'Cyclist']              # waymo_eval.py:28
_TYPE_KEY = {1: 'TYPE_VEHICLE', 2: 'TYPE_PEDESTRIAN', 3: 'TYPE_SIGN', 4: 'TYPE_CYCLIST'}
_RANGES = [('[0, 30)', 0.0, 30.0), ('[30, 50)', 30.0, 50.0), ('[50, +inf)', 50.0, np.inf)]


# ---- rotated-box IoU ---------------------------------------------------------------------------------------------------------
def _bev_corners(b):
    c, s = np.cos(b[:, 6]), np.sin(b[:, 6])
    hx, hy = b[:, 3] / 2, b[:, 4] / 2
    lx = np.stack([hx, -hx, -hx, hx], axis=1)
    ly = np.stack([hy, hy, -hy, -hy], axis=1)
    return np.stack([b[:, None, 0] + lx * c[:, None] - ly * s[:, None], b[:, None, 1] + lx * s[:, None] + ly * c[:, None]], axis=2)


def _inside(pts, b, eps=1e-9):
    """pts [p,k,2] inside the rectangles b [p,7] (closed)."""
    d = pts - b[:, None, :2]
    c, s = np.cos(b[:, 6])[:, None], np.sin(b[:, 6])[:, None]
    lx = d[..., 0] * c + d[..., 1] * s
    ly = -d[..., 0] * s + d[..., 1] * c
    return (np.abs(lx) <= b[:, None, 3] / 2 + eps) & (np.abs(ly) <= b[:, None, 4] / 2 + eps)


def bev_intersection(a, b):
    """Area shared by the rotated rectangles a[i] and b[i] ([p,7] each, float64): the convex polygon spanned by the corners of
    one inside the other and the edge crossings, ordered by angle around their mean."""
    p = len(a)
    if p == 0:
        return np.zeros(0)
    ca, cb = _bev_corners(a), _bev_corners(b)
    a0, a1 = ca[:, :, None, :], np.roll(ca, -1, axis=1)[:, :, None, :]                  # [p,4,1,2]
    b0, b1 = cb[:, None, :, :], np.roll(cb, -1, axis=1)[:, None, :, :]                  # [p,1,4,2]
    r, s_ = a1 - a0, b1 - b0
    den = r[..., 0] * s_[..., 1] - r[..., 1] * s_[..., 0]
    qp = b0 - a0
    ok = np.abs(den) > 1e-12
    den_s = np.where(ok, den, 1.0)
    t = (qp[..., 0] * s_[..., 1] - qp[..., 1] * s_[..., 0]) / den_s
    u = (qp[..., 0] * r[..., 1] - qp[..., 1] * r[..., 0]) / den_s
    hit = ok & (t >= 0) & (t <= 1) & (u >= 0) & (u <= 1)
    cross = (a0 + t[..., None] * r).reshape(p, 16, 2)
    pts = np.concatenate([ca, cb, cross], axis=1)                                         # [p,24,2]
    valid = np.concatenate([_inside(ca, b), _inside(cb, a), hit.reshape(p, 16)], axis=1)
    n = valid.sum(1)
    centre = (pts * valid[..., None]).sum(1) / np.maximum(n, 1)[:, None]
    ang = np.arctan2(pts[..., 1] - centre[:, None, 1], pts[..., 0] - centre[:, None, 0])
    ang = np.where(valid, ang, np.inf)
    order = np.argsort(ang, axis=1)
    pts = np.take_along_axis(pts, order[..., None], axis=1)
    valid = np.take_along_axis(valid, order, axis=1)
    pts = np.where(valid[..., None], pts, pts[:, :1, :])                                  # padding = first vertex: adds no area
    x, y = pts[..., 0] - centre[:, None, 0], pts[..., 1] - centre[:, None, 1]
    area = 0.5 * np.abs((x * np.roll(y, -1, axis=1) - np.roll(x, -1, axis=1) * y).sum(1))
    return np.where(n >= 3, area, 0.0)


def iou3d_matrix(a, b):
    """[na,7] x [nb,7] (x,y,z,dx,dy,dz,heading) -> [na,nb] 3-D IoU of the rotated boxes."""
    a, b = [np.asarray(x, np.float64).reshape(-1, np.shape(x)[-1] if np.ndim(x) == 2 else 7)[:, :7] for x in (a, b)]
    out = np.zeros((len(a), len(b)))
    if len(a) == 0 or len(b) == 0:
        return out
    ra, rb = 0.5 * np.hypot(a[:, 3], a[:, 4]), 0.5 * np.hypot(b[:, 3], b[:, 4])
    near = np.hypot(a[:, None, 0] - b[None, :, 0], a[:, None, 1] - b[None, :, 1]) <= ra[:, None] + rb[None, :]
    zov = np.minimum(a[:, None, 2] + a[:, None, 5] / 2, b[None, :, 2] + b[None, :, 5] / 2) - \
        np.maximum(a[:, None, 2] - a[:, None, 5] / 2, b[None, :, 2] - b[None, :, 5] / 2)
    i, j = np.nonzero(near & (zov > 0))
    if len(i):
        inter = bev_intersection(a[i], b[j]) * zov[i, j]
        union = a[i, 3] * a[i, 4] * a[i, 5] + b[j, 3] * b[j, 4] * b[j, 5] - inter
        out[i, j] = np.where(union > 0, inter / np.maximum(union, 1e-300), 0.0)
    return out


# ---- the adapters' `evaluation` up to the metric (pinned) -----------------------------------------------------------------------
def _in_range(boxes, eval_range):
    corners = boxes_to_corners_3d(np.asarray(boxes)[:, :7])
    return np.count_nonzero(((corners[..., :2] < eval_range[0:2]) | (corners[..., :2] > eval_range[2:4])).reshape(len(corners), -1),
                            axis=1) == 0


def filter_for_evaluation(dataset, det_annos, class_names, **kwargs):
    """-> (eval_det_annos, eval_gt_annos): waymo_dataset.py:229-320 / argo2_dataset.py:309-366 (`style` picks the variant:
    the Argoverse adapter drops 'unknown' ground truth here and has no IoU-based removal of detections)."""
    style = kwargs.get('style', 'argo2' if type(dataset).__name__.startswith('Argo2') else 'waymo')
    eval_range = np.asarray(kwargs.get('eval_range', dataset.point_cloud_range[[0, 1, 3, 4]]), dtype=np.float64)
    sampling_rate = kwargs.get('sampling_rate', 1)
    score_thresh = kwargs.get('score_thresh', 0.0)
    dets = deepcopy(det_annos)[::sampling_rate]
    for anno in dets:
        if len(anno['boxes_lidar']) > 0:
            if kwargs.get('bev', False):
                anno['boxes_lidar'][..., 2] = 0.0
                anno['boxes_lidar'][..., 5] = 1.0
            if kwargs.get('class_agnostic', False):
                anno['name'] = [class_names[0] for _ in range(len(anno['name']))]
            mask = _in_range(anno['boxes_lidar'], eval_range)
            mask[anno['score'] < score_thresh] = False
            anno['boxes_lidar'] = np.array(anno['boxes_lidar'])[mask]
            anno['name'] = np.array(anno['name'])[mask]
            anno['score'] = np.array(anno['score'])[mask]
            if style == 'waymo' and 'moving' in anno:
                anno['moving'] = np.array(anno['moving'])[mask]
    if kwargs.get('sequence', False):
        gts = [deepcopy(info['annos']) for info in dataset.sequence_infos]
    else:
        indices = kwargs.get('indices', dataset.index_mapping)
        indices = indices if len(indices) > 0 else dataset.index_mapping
        if style == 'argo2' and len(indices) == 0:
            indices = np.arange(len(dataset.infos))
        gts = [deepcopy(dataset.infos[idx]['annos']) for idx in indices]
    if kwargs.get('class_agnostic', False):
        for anno in gts:
            anno['name'] = np.array([class_names[0] if name in class_names else name for name in anno['name']])
    gts = gts[::sampling_rate]
    want_moving, want_static = kwargs.get('moving', False), kwargs.get('static', False)
    for a_idx, anno in enumerate(gts):
        if 'difficulty' not in anno or anno['difficulty'] is None:
            anno['difficulty'] = np.ones(len(anno['name']))
        if style == 'argo2':
            anno = drop_info_with_name(anno, name='unknown')           # a local copy: only the four keys below reach gts[a_idx]
        elif kwargs.get('bev', False) and len(anno['gt_boxes_lidar']) > 0:
            gts[a_idx]['gt_boxes_lidar'][..., 2] = 0.0
            gts[a_idx]['gt_boxes_lidar'][..., 5] = 1.0
        if len(anno['gt_boxes_lidar']) > 0:
            mask = _in_range(np.array(anno['gt_boxes_lidar']), eval_range)
            if style == 'waymo':
                check = mask.copy()            # ground truth of the OTHER kind: detections touching it are taken out of the count
                if want_moving:
                    check &= ~anno['moving']
                if want_static:
                    check &= anno['moving']
                if want_moving or want_static:
                    d = dets[a_idx]
                    other = np.array(anno['gt_boxes_lidar'])[check]
                    iou = iou3d_matrix(np.asarray(d['boxes_lidar'], np.float32), np.asarray(other, np.float32))
                    free = iou.sum(axis=1) == 0
                    d['boxes_lidar'], d['name'], d['score'] = (np.asarray(d[k])[free] for k in ('boxes_lidar', 'name', 'score'))
            if want_moving:
                mask &= anno['moving']
            if want_static:
                mask &= ~anno['moving']
            for k in ('difficulty', 'gt_boxes_lidar', 'name', 'num_points_in_gt'):
                gts[a_idx][k] = np.array(anno[k])[mask]
        if style == 'argo2' and kwargs.get('bev', False) and len(anno['gt_boxes_lidar']) > 0:
            gts[a_idx]['gt_boxes_lidar'][..., 2] = 0.0
            gts[a_idx]['gt_boxes_lidar'][..., 5] = 1.0
    return dets, gts


def limit_period(val, offset=0.5, period=np.pi):
    return val - np.floor(val / period + offset) * period


def waymo_type_results(infos, class_names, is_gt=False, fake_gt_infos=True):
    """The evaluator's flat per-box table over all frames: frame index, 7-value boxes (heading wrapped into [-pi, pi)), Waymo type
    index, score, no-label-zone overlap (always 0) and difficulty -- what `generate_waymo_type_results` hands the metric
    (waymo_eval.py:30-92), built as ONE pass over the concatenated frames instead of a per-frame append loop.
    Ground truth (is_gt): boxes of the evaluated classes with at least one LiDAR point; an unset difficulty (0) becomes level 1 with
    more than five points and level 2 otherwise -- written back into `info['difficulty']` like upstream does -- and with
    `fake_gt_infos` every frame's `gt_boxes_lidar` is replaced by its conversion from the old KITTI layout (also upstream's side effect).
    Detections: every box, difficulty 0, the frame's own scores."""
    infos = list(infos)
    key = 'gt_boxes_lidar' if is_gt else 'boxes_lidar'
    if is_gt:
        if any('num_points_in_gt' not in info for info in infos):
            raise NotImplementedError('num_points_in_gt is required for the Waymo evaluation')
        if fake_gt_infos:
            for info in infos:
                info['gt_boxes_lidar'] = boxes3d_kitti_fakelidar_to_lidar(info['gt_boxes_lidar'])
    per_frame = np.array([len(info[key]) for info in infos], dtype=np.int64)
    cuts = np.cumsum(per_frame)[:-1]
    frame_of = np.repeat(np.arange(len(infos), dtype=np.int64), per_frame)
    boxes = np.concatenate([np.asarray(info[key])[:, :7] for info in infos], axis=0)
    names = np.concatenate([np.asarray(info['name']) for info in infos])
    if is_gt:
        n_pts = np.concatenate([np.asarray(info['num_points_in_gt']) for info in infos])
        level = np.concatenate([np.asarray(info['difficulty']) for info in infos])
        unset = level == 0
        level[unset] = np.where(n_pts[unset] > 5, 1, 2)
        for info, part in zip(infos, np.split(level, cuts)):
            info['difficulty'][...] = part
        keep = np.isin(names, list(class_names)) & (n_pts > 0)
        frame_of, boxes, names, level = frame_of[keep], boxes[keep], names[keep], level[keep]
        score = np.ones(len(frame_of))
    else:
        level = np.zeros(len(frame_of), dtype=np.int64)
        score = np.concatenate([np.asarray(info['score']) for info in infos]).reshape(-1)
    type_of = {name: i for i, name in enumerate(WAYMO_CLASSES)}
    unknown = sorted(set(names.tolist()) - set(type_of))
    if unknown:
        raise ValueError(f'{unknown[0]!r} is not in list')           # what upstream's WAYMO_CLASSES.index raises
    obj_type = np.array([type_of[n] for n in names.tolist()], dtype=np.int64).reshape(-1)
    boxes = np.array(boxes)
    boxes[:, -1] = limit_period(boxes[:, -1], offset=0.5, period=np.pi * 2)
    return frame_of, boxes, obj_type, score, np.zeros(len(frame_of)), level.astype(np.int8)


def mask_by_distance(distance_thresh, boxes_3d, *args):
    mask = np.linalg.norm(boxes_3d[:, 0:2], axis=1) < distance_thresh + 0.5
    return (boxes_3d[mask],) + tuple(a[mask] for a in args)


def build_config(difficulties=(1, 2), breakdown_range=True, iou_thresholds=(0.7, 0.5, 0.5, 0.5), **kwargs):
    """waymo_eval.py:94-124 as a plain dict (the protobuf is not available)."""
    difficulties = list(difficulties)
    levels = ([1] if (1 in difficulties or len(difficulties) == 0) else []) + ([2] if 2 in difficulties else [])
    return {'breakdowns': ['OBJECT_TYPE'] + (['RANGE'] if breakdown_range else []), 'levels': levels,
            'iou_thresholds': [0.0] + [float(t) for t in iou_thresholds],
            'score_cutoffs': [x * 0.01 for x in range(0, 100)] + [1.0]}


# ---- the metric (restated, see the module header) ---------------------------------------------------------------------------------
def _heading_accuracy(pd_heading, gt_heading):
    d = np.abs(pd_heading - gt_heading) % (2 * np.pi)
    return 1.0 - np.minimum(d, 2 * np.pi - d) / np.pi


def _average_precision(precision, recall):
    """precision / recall per score cutoff (ascending cutoffs => recall falls): envelope, then the area over recall."""
    order = np.argsort(-np.asarray(recall), kind='stable')
    r, p = np.asarray(recall, np.float64)[order], np.asarray(precision, np.float64)[order]
    p = np.maximum.accumulate(p)                                        # best precision at any recall >= r (r falls with the index)
    r_next = np.r_[r[1:], 0.0]
    return float(np.sum((r - r_next) * p))


def detection_metrics(pd_frame, pd_box, pd_type, pd_score, gt_frame, gt_box, gt_type, gt_difficulty, config):
    """-> {'<BREAKDOWN>_<TYPE>[_<range>]_LEVEL_<L>/AP' | '/APH': [value]} (the key layout of the TF metric ops)."""
    cutoffs = np.asarray(config['score_cutoffs'], np.float64)
    nc = len(cutoffs)
    shards = []                                                     # (key prefix, type, r_lo, r_hi)
    for bd in config['breakdowns']:
        for t in (1, 2, 3, 4):
            if bd == 'OBJECT_TYPE':
                shards.append((f'OBJECT_TYPE_{_TYPE_KEY[t]}', t, 0.0, np.inf))
            else:
                shards += [(f'RANGE_{_TYPE_KEY[t]}_{name}', t, lo, hi) for name, lo, hi in _RANGES]
    levels = config['levels']
    tp = np.zeros((len(shards), len(levels), nc)); fp = np.zeros_like(tp); fn = np.zeros_like(tp); ha = np.zeros_like(tp)
    pd_rng = np.linalg.norm(np.asarray(pd_box)[:, :3], axis=1) if len(pd_box) else np.zeros(0)
    gt_rng = np.linalg.norm(np.asarray(gt_box)[:, :3], axis=1) if len(gt_box) else np.zeros(0)
    frames = np.union1d(np.unique(pd_frame), np.unique(gt_frame))
    for f in frames:
        pi_f, gi_f = np.flatnonzero(pd_frame == f), np.flatnonzero(gt_frame == f)
        for t in (1, 2, 3, 4):
            pi, gi = pi_f[pd_type[pi_f] == t], gi_f[gt_type[gi_f] == t]
            if len(pi) == 0 and len(gi) == 0:
                continue
            thr = config['iou_thresholds'][t]
            iou = iou3d_matrix(pd_box[pi], gt_box[gi])
            w = np.where((iou >= thr) & (iou > 0), iou, 0.0)
            sc = pd_score[pi]
            g_lvl = gt_difficulty[gi]
            prev_keep, match_p = None, None
            for ci, cut in enumerate(cutoffs):
                keep = np.flatnonzero(sc >= cut)
                if prev_keep is None or len(keep) != len(prev_keep):        # same prediction set => same assignment
                    match_p = np.full(len(pi), -1)
                    if len(keep) and len(gi):
                        rr, cc = linear_sum_assignment(-w[keep])
                        good = w[keep][rr, cc] > 0
                        match_p[keep[rr[good]]] = cc[good]
                    prev_keep = keep
                matched_g = np.zeros(len(gi), bool)
                matched_g[match_p[keep][match_p[keep] >= 0]] = True
                mk = match_p[keep]
                un = mk < 0
                gm = mk[~un]
                hacc = _heading_accuracy(pd_box[pi[keep[~un]], 6], gt_box[gi[gm], 6])
                for si, (_, st, lo, hi) in enumerate(shards):
                    if st != t:
                        continue
                    p_in = (pd_rng[pi] >= lo) & (pd_rng[pi] < hi)
                    g_in = (gt_rng[gi] >= lo) & (gt_rng[gi] < hi)
                    for li, lvl in enumerate(levels):
                        ok = g_in[gm] & (g_lvl[gm] <= lvl)
                        fp[si, li, ci] += np.count_nonzero(p_in[keep[un]])
                        tp[si, li, ci] += np.count_nonzero(ok)
                        ha[si, li, ci] += hacc[ok].sum()
                        fn[si, li, ci] += np.count_nonzero(~matched_g & g_in & (g_lvl <= lvl))
    out = {}
    for si, (prefix, _, _, _) in enumerate(shards):
        for li, lvl in enumerate(levels):
            den_p, den_r = tp[si, li] + fp[si, li], tp[si, li] + fn[si, li]
            prec = np.where(den_p > 0, tp[si, li] / np.maximum(den_p, 1), 0.0)
            rec = np.where(den_r > 0, tp[si, li] / np.maximum(den_r, 1), 0.0)
            prec_h = np.where(den_p > 0, ha[si, li] / np.maximum(den_p, 1), 0.0)
            rec_h = np.where(den_r > 0, ha[si, li] / np.maximum(den_r, 1), 0.0)
            out[f'{prefix}_LEVEL_{lvl}/AP'] = [_average_precision(prec, rec)]
            out[f'{prefix}_LEVEL_{lvl}/APH'] = [_average_precision(prec_h, rec_h)]
    return out


def waymo_evaluation(prediction_infos, gt_infos, class_name, distance_thresh=100, fake_gt_infos=True, cfg=None):
    """waymo_eval.py:197-231."""
    assert len(prediction_infos) == len(gt_infos), '%d vs %d' % (len(prediction_infos), len(gt_infos))
    pd_frame, pd_box, pd_type, pd_score, pd_nlz, _ = waymo_type_results(prediction_infos, class_name, is_gt=False)
    gt_frame, gt_box, gt_type, gt_score, _, gt_diff = waymo_type_results(gt_infos, class_name, is_gt=True, fake_gt_infos=fake_gt_infos)
    pd_box, pd_frame, pd_type, pd_score, pd_nlz = mask_by_distance(distance_thresh, pd_box, pd_frame, pd_type, pd_score, pd_nlz)
    gt_box, gt_frame, gt_type, gt_score, gt_diff = mask_by_distance(distance_thresh, gt_box, gt_frame, gt_type, gt_score, gt_diff)
    if len(pd_score) and pd_score.max() > 1:
        pd_score = 1 / (1 + np.exp(-pd_score))
    return detection_metrics(pd_frame, pd_box, pd_type, pd_score, gt_frame, gt_box, gt_type, gt_diff, build_config(**dict(cfg or {})))


def evaluate_detections(dataset, det_annos, class_names, **kwargs):
    dets, gts = filter_for_evaluation(dataset, det_annos, class_names, **kwargs)
    if kwargs.get('eval_metric', 'waymo') != 'waymo':
        raise NotImplementedError
    return waymo_evaluation(dets, gts, class_name=class_names, distance_thresh=1000,
                            fake_gt_infos=_get(dataset.dataset_cfg, 'INFO_WITH_FAKELIDAR', False), cfg=kwargs.get('eval_cfg', {}))


_NAMES = {'TYPE_VEHICLE': 'Vehicle', 'TYPE_PEDESTRIAN': 'Pedestrian', 'TYPE_CYCLIST': 'Cyclist'}


def eval_log_lines(ap_dict):
    """The rows eval_utils.print_eval_log prints (eval_utils.py:14-140), in its order: object-type AP / APH per level, then the
    range rows; values in percent with two decimals."""
    lines = []
    for t in ('TYPE_VEHICLE', 'TYPE_PEDESTRIAN', 'TYPE_CYCLIST'):
        for m, label in (('AP', 'AP '), ('APH', 'APH')):
            for lvl in (1, 2):
                k = f'OBJECT_TYPE_{t}_LEVEL_{lvl}/{m}'
                if k in ap_dict:
                    lines.append(f'{_NAMES[t]} {label} L{lvl}: {ap_dict[k][0] * 100:0.2f}')
    for t in ('TYPE_VEHICLE', 'TYPE_PEDESTRIAN', 'TYPE_CYCLIST'):
        for lvl in (1, 2):
            for m, label in (('AP', 'AP '), ('APH', 'APH')):
                for name, _, _ in _RANGES:
                    k = f'RANGE_{t}_{name}_LEVEL_{lvl}/{m}'
                    if k in ap_dict:
                        lines.append(f'{_NAMES[t]} {label} L{lvl} {name}: {ap_dict[k][0] * 100:0.2f}')
    return lines


def print_eval_log(ap_dict, logger):
    for line in eval_log_lines(ap_dict):
        logger.info(line)
