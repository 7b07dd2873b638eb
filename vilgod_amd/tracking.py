"""Cluster tracking over a sequence -- SURVEY §8f row N2, first part: `track_clusters`.

Mirrors
  ZeroShotDetector.track_clusters      src/vilgod/zero_shot_detector.py:298-327
  Tracker.next / finish                src/vilgod/tracker.py:31-79
  Track (init/predict/update/finalize) src/dataclass/objects.py:200-333
  assign_detections_greedy             src/utils/tracking_utils.py:54-95
  filterpy.kalman.KalmanFilter, filterpy.common.Q_discrete_white_noise   (third-party, un-vendored and not installed here:
      the published predict / update equations are restated below -- parity unpinned for that dependency)

Host logic on small data (a few hundred cluster centres per frame), sequential over the frames like upstream.  Detections are
rows of the frames' `FrameState`; a track is a list of entries (frame, row of the detection it shows, is_prediction).

Upstream behaviours kept on purpose (they shape the output):
  * `Detection.cluster_mass_center` re-computes the median of the cluster's points on every access, so the Kalman-smoothed
    position written into it (objects.py:308,317) never survives: association always uses the raw medians.
  * the greedy assignment pairs EVERY detection with a track while both are free, whatever the distance; pairs farther than
    `max_distance` are dropped afterwards (tracker.py:45-48) -- unless the two clusters have similar sizes (ratio > 0.7) and
    their 5-column medians are < 5 apart (tracker.py:52-59): then the track is updated AND the detection also starts a new
    track (tracker.py:72-76 only looks at the near matches).
  * `finalize` only trims trailing predictions; it never invalidates a track (objects.py:321-333).
"""
import numpy as np
from scipy import spatial


class ConstantVelocityKalman:
    """filterpy.kalman.KalmanFilter(dim_x=4, dim_z=2) as configured by Track._init_kf (objects.py:264-277)."""

    def __init__(self, x, dt=0.1):
        self.x = np.array(x, dtype=np.float64)
        self.F = np.array([[1., 0., dt, 0.], [0., 1., 0., dt], [0., 0., 1., 0.], [0., 0., 0., 1.]])
        self.Q = q_discrete_white_noise4(dt, 0.15)
        self.H = np.array([[1., 0., 0., 0.], [0., 1., 0., 0.]])
        self.R = np.eye(2)                   # `R[2:, 2:] *= 10` touches nothing of a 2 x 2 matrix
        self.P = np.eye(4)
        self.P[2:, 2:] *= 50
        self.P *= 10.

    def predict(self):
        self.x = self.F @ self.x
        self.P = (self.F @ self.P) @ self.F.T + self.Q

    def update(self, z):
        z = np.asarray(z, dtype=np.float64)
        y = z - self.H @ self.x
        PHT = self.P @ self.H.T
        S = self.H @ PHT + self.R
        K = PHT @ np.linalg.inv(S)
        self.x = self.x + K @ y
        I_KH = np.eye(4) - K @ self.H
        self.P = (I_KH @ self.P) @ I_KH.T + (K @ self.R) @ K.T


def q_discrete_white_noise4(dt, var):
    """filterpy.common.Q_discrete_white_noise(dim=4, dt, var): the 4th-order single-variable block (used upstream as the
    process noise of the [x, y, vx, vy] state)."""
    return np.array([[dt ** 6 / 36, dt ** 5 / 12, dt ** 4 / 6, dt ** 3 / 6],
                     [dt ** 5 / 12, dt ** 4 / 4, dt ** 3 / 2, dt ** 2 / 2],
                     [dt ** 4 / 6, dt ** 3 / 2, dt ** 2, dt],
                     [dt ** 3 / 6, dt ** 2 / 2, dt, 1.]]) * var


def assign_greedy(det_xy, trk_xy, max_distance):
    """tracking_utils.py:54-95 -> (matches [k,2] (detection, track), near mask per detection)."""
    nd, nt = len(det_xy), len(trk_xy)
    if nd == 0 or nt == 0:
        return np.zeros((0, 2), np.int64), np.ones(nd, bool)
    cost = spatial.distance.cdist(det_xy, trk_xy, 'euclidean')
    order = np.argsort(cost.reshape(-1))
    det_free, trk_free = np.ones(nd, bool), np.ones(nt, bool)
    matches = []
    for flat in order:
        d, t = int(flat // nt), int(flat % nt)
        if det_free[d] and trk_free[t]:
            det_free[d] = trk_free[t] = False
            matches.append((d, t))
            if len(matches) == min(nd, nt):
                break
    matches = np.array(matches, np.int64).reshape(-1, 2)
    overlap = np.full(nd, max_distance + 1.0)
    overlap[matches[:, 0]] = cost[matches[:, 0], matches[:, 1]]
    return matches, overlap < max_distance


class Track:
    def __init__(self, tid):
        self.tid = tid
        self.active = True
        self.valid = True
        self.static = True
        self.frames = []             # frame index of every entry
        self.source = []             # (fnr, row) of the detection an entry shows (a prediction repeats the last real one)
        self.prediction = []         # True: cloned entry for a frame without a match
        self.n_missed = 0
        self.kf = None
        self.current = None          # predicted centre (5 columns, like the cluster median)
        self.class_label = 'Background'
        self.class_label_corrected = False
        self.class_label_corrected_by_size = False

    def __len__(self):
        return len(self.frames)

    def init(self, key, center, fnr):
        self.kf = ConstantVelocityKalman([center[0], center[1], 0., 0.])
        self._append(key, fnr, False)
        self.current = np.array(center, copy=True)               # keeps the median's float32: Kalman output is rounded into it

    def _append(self, key, fnr, pred):
        self.frames.append(fnr)
        self.source.append(key)
        self.prediction.append(pred)

    def predict(self, last_center):
        self.kf.predict()
        self.current[:2] = self.kf.x[:2]
        self.current[2] = last_center[2]

    def update(self, key, center, fnr):
        if key is not None:
            self.n_missed = 0
            self.kf.update(center[:2])
            self._append(key, fnr, False)
        else:
            self.n_missed += 1
            self._append(self.source[-1], fnr, True)

    def finalize(self):
        self.active = False
        k = 0
        for p in reversed(self.prediction):
            if not p:
                break
            k += 1
        if k:
            self.frames, self.source, self.prediction = self.frames[:-k], self.source[:-k], self.prediction[:-k]


class Tracker:
    def __init__(self, mode='cluster_center', max_distance=1.0, min_length=5, max_missed=3, **unused):
        if mode != 'cluster_center':
            raise NotImplementedError('tracking.cluster.mode: only cluster_center (the shipped configuration)')
        self.max_distance = float(max_distance)
        self.max_missed = int(max_missed)
        self.min_length = int(min_length)
        self.tracks = []

    @property
    def tracks_valid(self):
        return [t for t in self.tracks if t.valid]

    def next(self, fnr, keys, centers, n_points, lookup):
        """One frame.  keys: detection ids (fnr, row); centers [n,>=3] cluster medians; n_points [n];
        lookup(key) -> (median, n_points) of an earlier detection."""
        active = [t for t in self.tracks if t.active]
        for t in active:
            t.predict(lookup(t.source[-1])[0])
        trk = np.array([t.current for t in active]) if active else np.zeros((0, 5), np.float32)
        centers = np.asarray(centers)
        matches_all, near = assign_greedy(centers[:, :2], trk[:, :2], self.max_distance)
        matches = matches_all[near[matches_all[:, 0]]] if len(matches_all) else matches_all
        near_of = {int(t): int(d) for d, t in matches}
        any_of = {int(t): int(d) for d, t in matches_all}
        for ti, t in enumerate(active):
            if ti in near_of:
                d = near_of[ti]
                t.update(keys[d], centers[d], fnr)
            elif ti in any_of:
                d = any_of[ti]
                c2, n2 = lookup(t.source[-1])
                n1 = n_points[d]
                if min(n1, n2) / max(n1, n2) > 0.7 and np.linalg.norm(centers[d] - np.asarray(c2)) < 5:
                    t.update(keys[d], centers[d], fnr)
                else:
                    t.update(None, None, fnr)
            elif t.n_missed >= self.max_missed:
                t.finalize()
            else:
                t.update(None, None, fnr)
        taken = set(int(d) for d in matches[:, 0]) if len(matches) else set()
        for d in range(len(keys)):
            if d not in taken:
                t = Track(len(self.tracks))
                t.init(keys[d], centers[d], fnr)
                self.tracks.append(t)

    def finish(self):
        for t in self.tracks:
            if t.active:
                t.finalize()


# ================================================================================================================
# Boxes of tracked clusters -- the track branch of fit_bounding_boxes_simple (zero_shot_detector.py:463-684)
# ================================================================================================================
def static_box(points, rectangle):
    """zero_shot_detector.py:477-488 / :451-461: rectangle of the x,y footprint -> [cx, cy, cz, l, w, h + 0.3, rz] with l >= w.
    rectangle(xy) -> (corners [4,2], rz, area) is `minimum_bounding_rectangle` (pointcloud_utils.py:300-373)."""
    corners, rz, _ = rectangle(points[:, :2])
    l = np.linalg.norm(corners[0] - corners[1])
    w = np.linalg.norm(corners[0] - corners[-1])
    c = (corners[0] + corners[2]) / 2
    if w > l:
        l, w = w, l
        rz += np.pi / 2
    height = points[:, 2].max() - points[:, 2].min()
    return np.array([c[0], c[1], points[:, 2].min() + height / 2, l, w, height + 0.3, rz])


def _norm2(v):
    """np.linalg.norm of float32 2-vectors along the last axis (x.dot(x) then sqrt, in float32)."""
    return np.sqrt(v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1])


def motion_vectors(centers_xy, look_ahead=10, min_far=0.5, min_step=0.3, max_angle=60):
    """Direction of travel per track entry from the cluster medians (zero_shot_detector.py:491-566).
    For entry c: `far` = vector to the entry look_ahead-1 steps on (searched further while shorter than 0.5 m and none is known
    yet; otherwise the last good one is kept); the steps to the entries in between that point within 60 deg of `far` and are
    longer than 0.3 m are averaged with weights 0.95^(i+1) / sum 0.9^(i+1) (i = ABSOLUTE entry index) and blended 50:50 with
    the previous direction; no such step -> previous direction, else `far`.  Any entry without a `far` vector voids the whole
    track (-> []), which sends it down the static path.  float32 throughout, like the medians.

    Evaluated with array operations over the whole track (the line-by-line form, oracle/tracking_oracle.py, spent ~250 us of
    interpreter time per entry: 3.5 ms per frame of a 199-frame sequence); same float32 operations in the same order, so the
    directions are the same bits (tests/test_tracking.py)."""
    centers_xy = np.asarray(centers_xy)
    n = len(centers_xy)
    if n == 0:
        return []
    idx = np.arange(n)
    c_far = np.minimum(idx + look_ahead - 1, n - 1)
    cand = centers_xy[c_far] - centers_xy                                # [n,2] float32
    cand_norm = _norm2(cand)
    # ---- the far vector of every entry (sequential state: the last good one is kept) ----
    fars = np.empty_like(cand)
    far = None
    for c in range(n):
        if cand_norm[c] < min_far and far is None:
            k = c_far[c] + 1                                             # look further on while nothing longer than 0.5 m is known
            v, vn = cand[c], cand_norm[c]
            while vn < min_far and k < n:
                v = centers_xy[k] - centers_xy[c]
                vn = _norm2(v)
                k += 1
            if vn >= min_far:
                far = v
        elif cand_norm[c] >= min_far:
            far = cand[c]
        if far is None:
            return []
        fars[c] = far
    # ---- the steps c+1 .. c_far-1 of every entry, padded to look_ahead - 2 columns ----
    m = max(look_ahead - 2, 0)
    out = []
    if m:
        j = idx[:, None] + 1 + np.arange(m)[None, :]                     # absolute index i of the step's end
        ok = j < c_far[:, None]
        jj = np.minimum(j, n - 1)
        step = centers_xy[jj] - centers_xy[:, None, :]                   # [n,m,2]
        sn = _norm2(step)
        fn = _norm2(fars)
        with np.errstate(invalid='ignore', divide='ignore'):
            cos = (fars[:, None, 0] * step[..., 0] + fars[:, None, 1] * step[..., 1]) / (fn[:, None] * sn)
        ang = np.rad2deg(np.arccos(np.clip(cos, -0.9999, 0.9999)))
        sel = ok & (ang < max_angle) & (sn > min_step)
        w95 = np.array([0.95 ** (i + 1) for i in range(n + m + 1)], dtype=np.float32)[jj]
        w90 = np.array([0.9 ** (i + 1) for i in range(n + m + 1)])[jj]
        contrib = np.where(sel[..., None], step * w95[..., None], np.float32(0))
        cnt = sel.sum(axis=1)
        ssum = contrib.sum(axis=1, dtype=np.float32)                     # sequential over the steps: zeros of unselected ones change nothing
        wsum = np.zeros(n)
        for q in range(m):                                               # python floats added in step order, like upstream's `+=`
            wsum = np.where(sel[:, q], wsum + w90[:, q], wsum)
    else:
        cnt = np.zeros(n, int)
    for c in range(n):
        if cnt[c]:
            v = (ssum[c] / np.float32(cnt[c])) / float(wsum[c])       # (python float: the quotient stays float32, like upstream's)
            if out:
                v = v * 0.5 + out[-1] * 0.5
            out.append(v)
        elif out:
            out.append(out[-1])
        else:
            out.append(fars[c].copy())
    return out


def moving_boxes(points_list, directions, to_ego_list, top_k=3, centers3=None):
    """zero_shot_detector.py:572-659: a box per entry aligned with its direction of travel, all resized to the median size of the
    top_k entries with the most points and shifted so that the corner closest to the ego vehicle stays where it was.
    The rotations of a track come from one scipy call and the closest-corner shifts are array operations over the track; the
    products that go through BLAS upstream (np.dot) stay per-entry np.dot calls so that their rounding is the same."""
    from scipy.spatial.transform import Rotation as R
    n = len(points_list)
    d = np.asarray(directions)
    angle = np.arctan2(d[:, 1], d[:, 0])
    rots = R.from_euler('z', angle, degrees=False).as_matrix()
    boxes = np.empty((n, 7))
    corner_arr = np.empty((n, 4, 2))
    tops = np.empty(n)
    for j, pts_all in enumerate(points_list):
        rot = rots[j]
        center = np.median(pts_all[..., :3], axis=0) if centers3 is None else np.asarray(centers3[j], dtype=pts_all.dtype)
        proj = np.dot(pts_all[..., :3] - center, rot)
        mn, mx = proj.min(axis=0), proj.max(axis=0)
        rect = np.array([[mx[0], mn[1]], [mn[0], mn[1]], [mn[0], mx[1]], [mx[0], mx[1]]], dtype=np.float32)
        corners = np.dot(rect, rot[:2, :2].T)
        corners += center[:2]
        w = np.linalg.norm(corners[0] - corners[1])
        l = np.linalg.norm(corners[0] - corners[-1])
        c = (corners[0] + corners[2]) / 2
        corner_arr[j] = corners
        z = pts_all[:, 2]
        zmin, zmax = z.min(), z.max()
        height = zmax - zmin
        tops[j] = zmax
        boxes[j] = (c[0], c[1], zmin + height / 2, w, l, height, angle[j])
    top = np.argsort([len(p) for p in points_list])[-top_k:]
    ref = np.median(boxes[top], axis=0)
    cc = np.empty(n, dtype=np.int64)
    for i, T in enumerate(to_ego_list):
        h = np.hstack((np.concatenate([corner_arr[i], np.zeros((4, 1))], axis=1), np.ones((4, 1))))
        ego = np.einsum('ij,kj->ki', T, h)[:, :2]                      # apply_transform (pointcloud_utils.py:21-46)
        cc[i] = int(np.linalg.norm(ego, axis=1).argmin())
    dw, dl = ref[3] - boxes[:, 3], ref[4] - boxes[:, 4]
    ang = angle                                                         # (dtype of the directions, as upstream's np.arctan2 of them)
    sx = np.where((cc == 0) | (cc == 3), -1.0, 1.0)                     # corners 0,3 hold max x: grow towards -x
    sy = np.where((cc == 0) | (cc == 1), 1.0, -1.0)                     # corners 0,1 hold min y: grow towards +y
    boxes[:, 0] += sx * (dw / 2) * np.cos(ang)
    boxes[:, 1] += sx * (dw / 2) * np.sin(ang)
    boxes[:, 0] += sy * (dl / 2) * np.sin(-ang)
    boxes[:, 1] += sy * (dl / 2) * np.cos(-ang)
    boxes[..., 3:6] = ref[3:6]
    boxes[..., 2] = tops - (ref[5] / 2)
    return boxes


def moving_boxes_packed(xyz, seg, directions, to_ego, centers3):
    """moving_boxes on packed cluster points ([P,3] + offsets): the helper-process form (vilgod_amd/box_worker.py)."""
    pts = [xyz[seg[i]:seg[i + 1]] for i in range(len(seg) - 1)]
    return moving_boxes(pts, list(directions), list(to_ego), centers3=None if centers3 is None else list(centers3))


class DetectionTable:
    """Mutable per-detection state shared by all tracks that hold the detection (the far-match rule can put one detection into
    two tracks, and upstream mutates the one Detection object from both): box, static_track, valid, class name / score.
    Entries that are predictions are private clones of their track (objects.py:311-317) and live in the track instead."""

    def __init__(self):
        self.box, self.static_track, self.valid, self.name, self.score = {}, {}, {}, {}, {}


def _entry_get(tab, t, i, field):
    return getattr(t, 'clone_' + field)[i] if t.prediction[i] else getattr(tab, field).get(t.source[i])


def _entry_set(tab, t, i, field, value):
    if t.prediction[i]:
        getattr(t, 'clone_' + field)[i] = value
    else:
        getattr(tab, field)[t.source[i]] = value


def fit_track_boxes(tracker, tab, points_of, static_of, to_ego_of, rectangle=None, static_box_of=None, median_of=None,
                    moving_async=None, max_pending=8):
    """The track branch of fit_bounding_boxes_simple for every valid track, in track order: boxes and `static_track` flags of
    the entries (into `tab` for real detections, into the track for its clones) and `track.static`.
    points_of(key) -> cluster points [n,>=3]; static_of(key) -> Detection.static (the entropy flag); to_ego_of(fnr) -> 4x4;
    rectangle(xy) -> (corners, rz, area), or static_box_of(key) -> the finished static box (the GPU kernel's, vg_cluster_boxes).
    moving_async(points_list, directions, to_ego_list, centers3) -> object with .result(): the moving tracks' boxes computed
    elsewhere (helper processes, boxes.submit_moving_boxes) while this thread goes on with the next track.  What a track needs is
    decided first for all tracks (nothing of it reads `tab`); the entries are then written in track order, as before -- one detection
    can sit in two tracks (far-match rule), and the later track's values must win.  At most `max_pending` moving_async requests are
    outstanding (each holds its track's packed points until it is answered), and every request is answered BEFORE the first entry is
    written: a helper that fails leaves `tab` and the tracks untouched."""
    sbox = (lambda k, p: np.array(static_box_of(k), dtype=np.float64)) if static_box_of is not None else (lambda k, p: static_box(p, rectangle))
    need_pts = static_box_of is None                     # the finished static boxes make the points of static tracks unnecessary
    plan = []
    pending = []                                         # positions in `plan` whose boxes are still being computed elsewhere

    def settle(keep):
        while len(pending) > keep:
            j = pending.pop(0)
            t_, _, kind_, res_ = plan[j]
            plan[j] = (t_, None, kind_, res_.result())
    for t in tracker.tracks_valid:
        n = len(t)
        pts = [points_of(k) for k in t.source] if need_pts else [None] * n
        if all(static_of(k) for k in t.source):
            plan.append((t, pts, 'static', None))
            continue
        # median_of(key) -> np.median(cluster points, axis=0) (the medians track_clusters already has: vg_cluster_medians)
        if median_of is not None:
            med = [median_of(k) for k in t.source]
            centers = np.array([m[:2] for m in med])
        else:
            if not need_pts:
                pts = [points_of(k) for k in t.source]
            med = None
            centers = np.array([np.median(p[..., :2], axis=0) for p in pts])
        dirs = motion_vectors(centers)
        if dirs:
            if pts[0] is None:
                pts = [points_of(k) for k in t.source]
            c3 = [m[:3] for m in med] if med is not None else None
            egos = [to_ego_of(f) for f in t.frames]
            if moving_async is not None:
                settle(max(int(max_pending), 1) - 1)
                plan.append((t, None, 'moving', moving_async(pts, dirs, egos, c3)))
                pending.append(len(plan) - 1)
            else:
                plan.append((t, None, 'moving', moving_boxes(pts, dirs, egos, centers3=c3)))
        else:
            plan.append((t, pts, 'still', None))
    settle(0)
    for t, pts, kind, res in plan:
        n = len(t)
        t.clone_box = [None] * n
        t.clone_static_track = [None] * n
        t.clone_valid = [True] * n
        if kind == 'static':
            for i in range(n):
                _entry_set(tab, t, i, 'box', sbox(t.source[i], pts[i]))
        elif kind == 'moving':
            boxes = res
            for i in range(n):
                _entry_set(tab, t, i, 'box', boxes[i])
                _entry_set(tab, t, i, 'static_track', False)
            t.static = False
        else:
            for i in range(n):
                _entry_set(tab, t, i, 'static_track', True)
                _entry_set(tab, t, i, 'box', sbox(t.source[i], pts[i]))


# ================================================================================================================
# propagate_labels (zero_shot_detector.py:686-824)
# ================================================================================================================
def size_prior_class(box):
    """check_box (zero_shot_detector.py:691-701)."""
    l, w, h = box[3:6]
    if h > 0.8 and h <= 2.3 and w > 0.2 and w <= 1 and l > 0.2 and l <= 1:
        return 'Pedestrian'
    if h > 1.4 and h <= 2 and w > 0.5 and w <= 1 and l > 1 and l <= 2.5:
        return 'Cyclist'
    if w > 0.5 and w <= 3 and l > 0.5 and l <= 8.0 and h > 1 and h <= 3:
        return 'Vehicle'
    return 'Background'


def dominant_angles(angles, n_bins=45):
    """pointcloud_utils.bin_angles (:525-560): the angles (folded into [0, pi]) of the fullest of n_bins bins."""
    edges = np.linspace(0, np.pi, n_bins + 1)
    bins = [[] for _ in range(n_bins)]
    for a in angles:
        a = a % (2 * np.pi)
        if a > np.pi:
            a %= np.pi
        b = np.digitize(a, edges, right=False) - 1
        if 0 <= b < n_bins:
            bins[b].append(a)
    return bins[int(np.argmax([len(b) for b in bins]))]


def rectangles_overlap(a, b):
    """Do the rotated BEV rectangles of boxes a and b [cx,cy,cz,l,w,h,rz] share area, and their z ranges length?  The only
    thing upstream asks of `boxes_iou3d_gpu` (a pcdet CUDA op) is `iou > 0` (zero_shot_detector.py:737-739)."""
    if min(a[2] + a[5] / 2, b[2] + b[5] / 2) - max(a[2] - a[5] / 2, b[2] - b[5] / 2) <= 0:
        return False

    def corners(bx):
        c, s = np.cos(bx[6]), np.sin(bx[6])
        hx, hy = bx[3] / 2, bx[4] / 2
        loc = np.array([[hx, hy], [-hx, hy], [-hx, -hy], [hx, -hy]])
        return loc @ np.array([[c, s], [-s, c]]) + bx[:2]

    p, q = corners(np.asarray(a, np.float64)), corners(np.asarray(b, np.float64))
    for poly in (p, q):
        for i in range(4):
            e = poly[(i + 1) % 4] - poly[i]
            ax = np.array([-e[1], e[0]])
            pp, qq = p @ ax, q @ ax
            if pp.max() <= qq.min() or qq.max() <= pp.min():
                return False
    return True


def propagate_labels(tracker, tab, n_points_of, class_names, min_length=5, top_k=10):
    """Track-level label and box clean-up; mutates `tab` (real detections), the tracks' clone arrays and track flags."""
    for t in tracker.tracks_valid:
        n = len(t)
        real = [i for i in range(n) if not t.prediction[i]]
        if n < min_length:
            for i in range(n):
                _entry_set(tab, t, i, 'valid', False)
            continue
        max_score, class_name, count = 0, 'Background', {}
        for i in real:
            k = t.source[i]
            if tab.score[k] > max_score:
                max_score, class_name = tab.score[k], tab.name[k]
            count[tab.name[k]] = count.get(tab.name[k], 0) + 1
        if not t.static:                                    # a "moving" track whose boxes all overlap its largest one is static
            boxes = np.array([_entry_get(tab, t, i, 'box') for i in range(n)], dtype=np.float64).astype(np.float32).astype(np.float64)
            ref = boxes[int(np.argmax(np.prod(boxes[..., 3:5], axis=1)))].copy()
            ref[2], ref[5] = 0, 1
            boxes[..., 2], boxes[..., 5] = 0, 1
            if all(rectangles_overlap(ref, b) for b in boxes):
                t.static = True
                for i in range(n):
                    _entry_set(tab, t, i, 'static_track', True)
        if t.static:
            if real:
                boxes = np.array([tab.box[t.source[i]] for i in real])
                npts = [n_points_of(t.source[i]) for i in real]
                boxes = boxes[np.argsort(npts)[::-1][:top_k]]
                median_box = np.median(boxes, axis=0)
                median_box[6] = np.mean(dominant_angles(boxes[..., 6]))
                l, w, h = median_box[3:6]
                if l < 0.2 or l > 20 or w < 0.2 or w > 3.5 or h < 0.5 or h > 4:
                    t.valid = False
                    for i in range(n):
                        _entry_set(tab, t, i, 'valid', False)
                    continue
                for i in range(n):
                    _entry_set(tab, t, i, 'box', median_box.copy())
        frac = (lambda: count[class_name] / n)
        for i in real:
            k = t.source[i]
            in_names = class_name in class_names
            if in_names and (max_score >= 0.5 or frac() >= 0.6):
                tab.name[k], tab.score[k] = class_name, max_score
                t.class_label_corrected, t.class_label = True, class_name
            elif (not t.static) and in_names and class_name in ('Cyclist', 'Pedestrian') and (max_score >= 0.35 or frac() >= 0.6):
                tab.name[k], tab.score[k] = class_name, 0.7
                t.class_label_corrected, t.class_label = True, class_name
            elif class_name == 'Background' and max_score >= 0.3:
                tab.name[k], tab.score[k] = class_name, (max_score if not t.static else 1.0)
                t.class_label_corrected, t.class_label = True, class_name
            elif not t.static:
                new = size_prior_class(tab.box[k])
                t.class_label_corrected_by_size = new != tab.name[k]
                t.class_label = new
                tab.name[k], tab.score[k] = new, 0.5
            if not t.static:
                tab.static_track[k] = False
            box = np.array(tab.box[k], dtype=np.float64, copy=True)      # enlarge by a small margin
            box[3:5] += 0.3
            tab.box[k] = box
