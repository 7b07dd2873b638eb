"""Per-frame state as structure-of-arrays, materialised into the reference's dict layout only when a
pickle is written (SURVEY §8a row T1, §8b "Output 1").

Mirrors what `LidarFrame` (src/vilgod/lidar_frame.py:15-59, 124-147) and `Detection`
(src/dataclass/objects.py:37-103, 136-142) hold and serialise:
  frame dict keys   `_detections`, `_ground_point_indices`, `_entropy_scores`, `_entropy_indices`, `_gt_cluster_mapping`
  detection keys    `cluster_id`, `_bounding_box`, `valid`, `static`, `gt_assigned`, `cluster_points_index`,
                    `object_class_predictions`, `tid`, `static_track`, `object_class_predictions_detailed`,
                    `object_class_predictions_score`, `object_class`, `object_class_score`
                    (a key is present only when its value is not None -- objects.py:95-100)
"""
import numpy as np

DETECTION_FIELDS = ['cluster_id', '_bounding_box', 'valid', 'static', 'gt_assigned', 'cluster_points_index',
                    'object_class_predictions', 'tid', 'static_track', 'object_class_predictions_detailed',
                    'object_class_predictions_score', 'object_class', 'object_class_score']       # objects.py:90-93


def pack_clusters(labels, probs, threshold):
    """lidar_frame.py:163-167, 230-237 as arrays: labels of low-probability points -> -1; clusters in ascending
    label order; each cluster's point indices ascending.  Returns (cluster_ids, packed index int32, seg_off int32).
    One counting sort on the host (csrc/hdbscan_tree.cpp vg_pack_clusters_host): the numpy form below -- a stable argsort of ~80k labels --
    was 2 ms of a frame's 12.6 ms front-stage latency (round 5)."""
    import ctypes
    from ._lib import lib, check
    labels = np.asarray(labels)
    if labels.size and (int(labels.max()) > np.iinfo(np.int32).max or int(labels.min()) < np.iinfo(np.int32).min):
        return pack_clusters_numpy(labels, probs, threshold)        # (labels beyond int32 would wrap in the cast below)
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    n = len(labels)
    pr = None if probs is None else np.ascontiguousarray(probs, dtype=np.float64)
    ids = np.empty(n, np.int64)
    index = np.empty(n, np.int32)
    seg = np.empty(n + 1, np.int32)
    nc = ctypes.c_int32(0)
    p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    check(lib.vg_pack_clusters_host(p(labels), p(pr), n, float(threshold), p(ids), p(index), p(seg), ctypes.byref(nc)),
          'vg_pack_clusters_host')
    c = nc.value
    return ids[:c].copy(), index[:int(seg[c])].copy(), seg[:c + 1].copy()


def pack_clusters_numpy(labels, probs, threshold):
    """The same grouping in numpy (what `pack_clusters` computed before round 5; tests compare the two)."""
    labels = np.asarray(labels).copy()
    if probs is not None:
        labels[np.asarray(probs) < threshold] = -1
    keep = np.flatnonzero(labels >= 0)
    if keep.size == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(1, np.int32)
    order = keep[np.argsort(labels[keep], kind='stable')]          # ascending label, then ascending index
    sl = labels[order]
    starts = np.flatnonzero(np.r_[True, sl[1:] != sl[:-1]])
    seg = np.r_[starts, len(order)].astype(np.int32)
    return sl[starts].astype(np.int64), order.astype(np.int32), seg


def vote(class_ids, scores, class_names_sorted):
    """LidarFrame.update_object_classes (lidar_frame.py:269-285) for all detections at once.
    class_ids: [C,V] indices into `class_names_sorted` (ALPHABETICAL order = np.unique order); scores [C,V] float32.
    Returns (winner index [C], score [C] float32).
      no tie for the highest count -> that name (first in alphabetical order), mean score of its views
      tie -> the name with the highest mean score among ALL names present, scanned alphabetically with `>`
             (max_score starts at 0)."""
    C, V = class_ids.shape
    K = len(class_names_sorted)
    counts = np.zeros((C, K), np.int64)
    sums = np.zeros((C, K), np.float32)
    for v in range(V):                                    # sequential float32 accumulation like np.mean on <8 items
        np.add.at(counts, (np.arange(C), class_ids[:, v]), 1)
        cur = sums[np.arange(C), class_ids[:, v]]
        sums[np.arange(C), class_ids[:, v]] = cur + scores[:, v].astype(np.float32)
    with np.errstate(invalid='ignore', divide='ignore'):
        means = (sums / counts.astype(np.float32)).astype(np.float32)
    mx = counts.max(axis=1)
    tie = (counts == mx[:, None]).sum(axis=1) > 1
    win = counts.argmax(axis=1)                           # first maximum = alphabetical first
    score = means[np.arange(C), win]
    for c in np.flatnonzero(tie):
        best, best_s = -1, np.float32(0)
        for k in range(K):
            if counts[c, k] > 0 and means[c, k] > best_s:
                best, best_s = k, means[c, k]
        win[c], score[c] = best, best_s
    return win, score


def static_from_entropy(ent_full, index, seg_off, percentile=30, min_percentile_pp_score=0.5):
    """`Detection.static = not filter_by_ephemeral_score(entropy[cluster_points_index], ...)` for every cluster
    (lidar_frame.py:238-243; cluster_utils.py:62-64): static <=> np.percentile(scores, percentile) > min score.
    np.percentile(method='linear') on a float32 array: virtual index t*(n-1), neighbours a <= b, float64
    a + (b-a)*g for g < 0.5 else b - (b-a)*(1-g).  ent_full float32 [M]; packed cluster index / offsets.  -> bool [C]."""
    C = len(seg_off) - 1
    if C == 0:
        return np.zeros(0, bool)
    vals = np.asarray(ent_full, np.float32)[index]
    seg_id = np.repeat(np.arange(C), np.diff(seg_off))
    order = np.lexsort((vals, seg_id))
    sv = vals[order].astype(np.float64)
    n = np.diff(seg_off).astype(np.int64)
    virt = (percentile / 100.0) * (n - 1)
    lo = np.floor(virt).astype(np.int64)
    g = virt - lo
    hi = np.minimum(lo + 1, n - 1)
    a, b = sv[seg_off[:-1] + lo], sv[seg_off[:-1] + hi]
    d = b - a
    q = np.where(g >= 0.5, b - d * (1 - g), a + d * g)
    return q > min_percentile_pp_score


class FrameState:
    def __init__(self, fnr, pose, ref_pose):
        self.fnr = fnr
        self.pose = np.asarray(pose, dtype=np.float64)
        self.ref_pose = np.asarray(ref_pose, dtype=np.float64)
        self.transform_to_ref = np.linalg.inv(self.ref_pose) @ self.pose          # lidar_frame.py:25
        self.transform_to_ego = np.linalg.inv(self.pose) @ self.ref_pose          # lidar_frame.py:26
        self.ground_point_indices = None
        self.n_points = 0
        self.ground_plane_model_ref = None
        self.entropy_scores = None                     # kept values (< 0.9), float64   (lidar_frame.py:256-258)
        self.entropy_indices = None                    # their indices into points_ref_wo_ground
        self.n_nonground = 0
        self.clear_detections()

    def clear_detections(self):
        self.cluster_ids = np.zeros(0, np.int64)
        self.index = np.zeros(0, np.int32)            # indices into points_ref_wo_ground
        self.seg_off = np.zeros(1, np.int32)
        self.valid = np.zeros(0, bool)
        self.static = np.zeros(0, bool)
        self.tid = np.zeros(0, np.int64)
        self.static_track = np.zeros(0, np.int8)       # -1 = None (never tracked), 0 = moving track, 1 = static track (objects.py:59)
        self.boxes = None                              # [C,7] float64 ref frame, NaN rows = no box
        self.cls = {}                                  # key -> dict(pred, detailed, score [C,V]; name, final [C]; has [C])
        self.filtered = False

    @property
    def n_detections(self):
        return len(self.cluster_ids)

    def set_clusters(self, cluster_ids, index, seg_off):
        self.clear_detections()
        C = len(cluster_ids)
        self.cluster_ids, self.index, self.seg_off = cluster_ids, index, seg_off
        self.valid = np.ones(C, bool)                  # Detection.valid default (objects.py:57)
        self.static = np.ones(C, bool)                 # :58
        self.tid = np.full(C, -1, np.int64)            # :64
        self.static_track = np.full(C, -1, np.int8)

    def cluster_index(self, c):
        return self.index[self.seg_off[c]:self.seg_off[c + 1]]

    def set_classes(self, key, which, pred, detailed, score, name, final_score):
        """which: bool [C] detections that were classified; the arrays have one row per True entry."""
        C = self.n_detections
        V = pred.shape[1] if len(pred) else 0
        e = dict(has=which.copy(), pred=np.empty((C, V), object), detailed=np.empty((C, V), object),
                 score=np.zeros((C, V), np.float32), name=np.empty(C, object), final=np.zeros(C, np.float64),
                 final_py=np.zeros(C, bool))
        # `final`: the vote's mean is a numpy float32 upstream (lidar_frame.py:276-283); propagate_labels may replace it by a
        # python float constant (0.5 / 0.7 / 1.0, zero_shot_detector.py:781-812), which decides the dtype of the result dict's
        # score array (:855) -- held here as float64 values (float32 means are exact in it) + a flag "python float"
        rows = np.flatnonzero(which)
        e['pred'][rows], e['detailed'][rows], e['score'][rows] = pred, detailed, score
        e['name'][rows], e['final'][rows] = name, final_score
        self.cls[key] = e

    def final_score(self, key, c):
        """Detection.object_class_score[key] with upstream's type: numpy float32 (vote) or python float (propagated constant)."""
        e = self.cls[key]
        return float(e['final'][c]) if e['final_py'][c] else np.float32(e['final'][c])

    def set_final_score(self, key, c, value):
        e = self.cls[key]
        e['final'][c] = value
        e['final_py'][c] = not isinstance(value, np.floating)

    # ---- reference-compatible (de)serialisation ---------------------------------------------------------
    def detection_dict(self, c):
        d = {'cluster_id': self.cluster_ids[c]}
        if self.boxes is not None and not np.isnan(self.boxes[c, 0]):
            d['_bounding_box'] = self.boxes[c].copy()
        # objects.py:158-181: the filters' numpy `&` leaves a numpy bool; the constructor default is the python True (:57)
        d['valid'] = np.bool_(self.valid[c]) if self.filtered else bool(self.valid[c])
        d['static'] = bool(self.static[c])
        d['gt_assigned'] = False
        d['cluster_points_index'] = self.cluster_index(c).astype(np.int64)
        has = [k for k, e in self.cls.items() if e['has'][c]]
        if has:
            d['object_class_predictions'] = {k: self.cls[k]['pred'][c].astype(str) for k in has}
        d['tid'] = int(self.tid[c])
        if self.static_track[c] >= 0:
            d['static_track'] = bool(self.static_track[c])
        if has:
            d['object_class_predictions_detailed'] = {k: self.cls[k]['detailed'][c].astype(str) for k in has}
            d['object_class_predictions_score'] = {k: self.cls[k]['score'][c].copy() for k in has}
            d['object_class'] = {k: np.str_(self.cls[k]['name'][c]) for k in has}      # an element of np.unique's result, lidar_frame.py:269-283
            d['object_class_score'] = {k: self.final_score(k, c) for k in has}
        return d

    @property
    def serialize(self):
        """lidar_frame.py:41-59."""
        frame = {'_detections': [self.detection_dict(c) for c in range(self.n_detections)]}
        if self.ground_point_indices is not None:
            frame['_ground_point_indices'] = self.ground_point_indices
        if self.entropy_scores is not None:
            frame['_entropy_scores'] = self.entropy_scores
            frame['_entropy_indices'] = self.entropy_indices
        frame['_gt_cluster_mapping'] = {}
        return frame

    # ---- compact form: what `serialize` reads, as a handful of arrays (the state-writer process builds the dicts from it) ----------
    _COMPACT_SHARED = ('cluster_ids', 'index', 'seg_off', 'ground_point_indices', 'entropy_scores', 'entropy_indices')
    _COMPACT_COPIED = ('valid', 'static', 'tid', 'static_track')

    def compact(self):
        """Everything `serialize` reads, detached from later changes: the arrays that later stages rewrite in place (flags, boxes,
        class results) are copied, the bulk that never changes once written (cluster membership, ground set, entropy scores) is
        shared.  ~15 arrays per frame instead of ~90 dicts of a dozen objects: cheap to make, cheap to send to another process."""
        d = {k: getattr(self, k) for k in self._COMPACT_SHARED}
        d.update({k: getattr(self, k).copy() for k in self._COMPACT_COPIED})
        d['boxes'] = None if self.boxes is None else self.boxes.copy()
        d['filtered'] = bool(self.filtered)
        d['cls'] = {k: {f: v.copy() for f, v in e.items()} for k, e in self.cls.items()}
        return d

    @classmethod
    def from_compact(cls, d):
        """A FrameState that can `serialize` (nothing else: no poses)."""
        fs = cls.__new__(cls)
        for k in cls._COMPACT_SHARED + cls._COMPACT_COPIED + ('boxes', 'filtered', 'cls'):
            setattr(fs, k, d[k])
        return fs

    def sync(self, data):
        """lidar_frame.py:124-147 + objects.py:136-142: restore from a serialised frame dict."""
        if '_ground_point_indices' in data:
            self.ground_point_indices = np.asarray(data['_ground_point_indices'])
        if data.get('_entropy_scores') is not None:
            self.entropy_scores = np.asarray(data['_entropy_scores'])
            self.entropy_indices = np.asarray(data['_entropy_indices'])
        dets = data.get('_detections')
        if not dets:
            return
        C = len(dets)
        idx = [np.asarray(d['cluster_points_index'], dtype=np.int32) for d in dets]
        self.set_clusters(np.array([d['cluster_id'] for d in dets], dtype=np.int64),
                          np.concatenate(idx) if idx else np.zeros(0, np.int32),
                          np.r_[0, np.cumsum([len(i) for i in idx])].astype(np.int32))
        self.valid = np.array([bool(d.get('valid', True)) for d in dets])
        self.static = np.array([bool(d.get('static', True)) for d in dets])
        self.tid = np.array([int(d.get('tid', -1)) for d in dets], dtype=np.int64)
        self.static_track = np.array([-1 if d.get('static_track') is None else int(bool(d['static_track'])) for d in dets], dtype=np.int8)
        self.filtered = bool((~self.valid).any())      # zero_shot_detector.py:265-274: any invalid detection = already filtered
        if any('_bounding_box' in d for d in dets):
            self.boxes = np.full((C, 7), np.nan)
            for c, d in enumerate(dets):
                if '_bounding_box' in d:
                    self.boxes[c] = d['_bounding_box']
        keys = sorted({k for d in dets for k in (d.get('object_class') or {})})
        for k in keys:
            has = np.array([k in (d.get('object_class') or {}) for d in dets])
            rows = [d for d in dets if k in (d.get('object_class') or {})]
            self.set_classes(k, has,
                             np.array([r['object_class_predictions'][k] for r in rows], dtype=object),
                             np.array([r['object_class_predictions_detailed'][k] for r in rows], dtype=object),
                             np.array([r['object_class_predictions_score'][k] for r in rows], dtype=np.float32),
                             np.array([r['object_class'][k] for r in rows], dtype=object),
                             np.array([r['object_class_score'][k] for r in rows], dtype=np.float64))
            for c, r in zip(np.flatnonzero(has), rows):
                self.cls[k]['final_py'][c] = not isinstance(r['object_class_score'][k], np.floating)
