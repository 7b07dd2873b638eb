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
