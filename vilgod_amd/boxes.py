"""Per-cluster oriented boxes: `fit_bounding_boxes_simple`, static branch
(src/vilgod/zero_shot_detector.py:444-462) over `minimum_bounding_rectangle` (src/utils/pointcloud_utils.py:309-372).

Two modes, selected by `PseudoLabelPipeline(box_mode=...)` / `device.box_mode`:

  'reference' (default)  the boxes the reference writes.  `minimum_bounding_rectangle` evaluates the rectangle only for the
              directions of qhull's vertex cycle WITHOUT its closing edge (:329-330 `hull_points[1:] - hull_points[:-1]`), so
              the result depends on which vertex qhull lists first -- an artefact of qhull's incremental construction (the
              oldest surviving facet) that no closed-form rule predicts.  The mode therefore does what the reference does where
              it matters: the hull comes from the same library call (`scipy.spatial.ConvexHull` on the cluster's float32 xy
              points), and the rectangle over its <= few dozen vertices is evaluated with the reference's own float32 numpy
              expression sequence (numpy's float32 `arctan2` / `cos` are SIMD routines whose last bit is host dependent, so only
              the same calls reproduce the reference's float32 boxes).  The per-point work stays on the GPU: cluster membership
              and packing, z extent (`vg_cluster_filter` statistics) and the validity filters; this module touches each
              cluster's xy points once, in a helper process (see `submit_reference_boxes`), while the GPU classifies the
              frame's crops (the boxes do not depend on the classes).
  'fast'      `vg_cluster_boxes` (csrc/segment.hip k_cluster_box): exact-predicate hull + rectangle over ALL hull edges in
              float64 on the GPU.  Identical to the reference whenever the best direction is not the dropped closing edge
              (~85 % of clusters), otherwise its rectangle is the smaller one.

Host code is numpy because the reference's is; there is no CPU fallback for the GPU parts.
"""
import numpy as np
from scipy import spatial

PI2 = np.pi / 2.


def minimum_bounding_rectangle(points):
    """pointcloud_utils.py:309-372, expression by expression (float32 in, float32 arithmetic like upstream).
    -> (corners (4,2) float64 array holding float32 values, rz, area)."""
    try:
        hull_points = points[spatial.ConvexHull(points).vertices]
    except Exception:                                     # qhull raises on < 3 points / flat input (:320-326)
        corners = np.ones((4, 2)) * np.mean(points[:, :2], axis=0)[:2]
        corners += np.array([[-0.05, -0.05], [0.05, -0.05], [0.05, 0.05], [-0.05, 0.05]])
        return corners, 0, 0
    edges = hull_points[1:] - hull_points[:-1]            # the closing edge of the vertex cycle is NOT there (:329-330)
    angles = np.arctan2(edges[:, 1], edges[:, 0])
    angles = np.abs(np.mod(angles, PI2))
    angles = np.unique(angles)
    rotations = np.vstack([np.cos(angles), np.cos(angles - PI2), np.cos(angles + PI2), np.cos(angles)]).T
    rotations = rotations.reshape((-1, 2, 2))
    rot_points = np.dot(rotations, hull_points.T)
    min_x = np.nanmin(rot_points[:, 0], axis=1)
    max_x = np.nanmax(rot_points[:, 0], axis=1)
    min_y = np.nanmin(rot_points[:, 1], axis=1)
    max_y = np.nanmax(rot_points[:, 1], axis=1)
    areas = (max_x - min_x) * (max_y - min_y)
    best_idx = np.argmin(areas)
    x1, x2, y1, y2 = max_x[best_idx], min_x[best_idx], max_y[best_idx], min_y[best_idx]
    r = rotations[best_idx]
    rval = np.zeros((4, 2))
    rval[0] = np.dot([x1, y2], r)
    rval[1] = np.dot([x2, y2], r)
    rval[2] = np.dot([x2, y1], r)
    rval[3] = np.dot([x1, y1], r)
    return rval, angles[best_idx], areas[best_idx]


def box_from_rectangle(corners, rz, zmin, zmax):
    """zero_shot_detector.py:452-461.  zmin / zmax: float32 scalars (the cluster's z extent)."""
    l = np.linalg.norm(corners[0] - corners[1])
    w = np.linalg.norm(corners[0] - corners[-1])
    c = (corners[0] + corners[2]) / 2
    if w > l:
        l, w = w, l
        rz += np.pi / 2
    height = zmax - zmin
    return np.array([c[0], c[1], zmin + height / 2, l, w, height + 0.3, rz])


def reference_boxes_packed(xy_packed, seg, zmin, zmax):
    """Boxes of clusters whose xy points are already packed cluster after cluster ([P,2] float32, seg offsets)."""
    C = len(seg) - 1
    out = np.empty((C, 7))
    for c in range(C):
        corners, rz, _ = minimum_bounding_rectangle(xy_packed[seg[c]:seg[c + 1]])
        out[c] = box_from_rectangle(corners, rz, zmin[c], zmax[c])
    return out


# ---- helper processes ------------------------------------------------------------------------------------------------
# The loop above is ~100 us of interpreter time per cluster (scipy's ConvexHull object + ~25 tiny numpy calls), i.e. ~9 ms per
# 150k-point frame -- in a process whose worker threads share one interpreter lock with the code that launches the GPU
# kernels of six frames in flight (measured: 53 -> 40 frames/s when it ran in the frames' threads).  It therefore runs in a
# small pool of helper PROCESSES (`python -m vilgod_amd.box_worker`: numpy / scipy only, never the GPU runtime); a frame's
# packed xy points (~0.3 MB) travel over a pipe, the frame's worker thread collects the boxes after it has queued the frame's
# classification.
class BoxWorkerPool:
    """n helper processes, each driven by one dispatcher thread that takes requests from a shared queue, so `submit` never blocks
    and any number of requests may be outstanding (a stage may queue every frame of a sequence before it reads the first answer)."""

    def __init__(self, n_procs):
        import queue
        self.requests = queue.Queue()
        self.procs, self.threads = [], []
        self.respawned = 0
        self.grow(n_procs)

    @staticmethod
    def _spawn():
        import os
        import subprocess
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
        env['PYTHONPATH'] = root + os.pathsep + env.get('PYTHONPATH', '')
        return subprocess.Popen([sys.executable, '-m', 'vilgod_amd.box_worker'], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env, cwd=root)

    def grow(self, n_procs):
        """At least n_procs helpers (a pipeline that asks for more than an earlier one built gets them)."""
        import threading
        while len(self.procs) < int(n_procs):
            slot = len(self.procs)
            self.procs.append(self._spawn())
            t = threading.Thread(target=self._serve, args=(slot,), daemon=True)
            t.start()
            self.threads.append(t)

    @staticmethod
    def _exchange(p, req):
        import pickle
        import struct
        blob = pickle.dumps(req, protocol=pickle.HIGHEST_PROTOCOL)
        p.stdin.write(struct.pack('<q', len(blob)))
        p.stdin.write(blob)
        p.stdin.flush()
        head = p.stdout.read(8)
        if len(head) < 8:
            raise BrokenPipeError('box helper process ended unexpectedly')
        (n,) = struct.unpack('<q', head)
        body = p.stdout.read(n)
        if len(body) < n:
            raise BrokenPipeError('box helper process ended in the middle of an answer')
        return pickle.loads(body)

    def _serve(self, slot):
        while True:
            item = self.requests.get()
            if item is None:
                return
            fut, req = item
            try:
                try:
                    status, val = self._exchange(self.procs[slot], req)
                except (BrokenPipeError, EOFError, OSError, ValueError) as e1:
                    # the helper died (or the pipe's framing is lost): a fresh child process, the request once more.  If that helper
                    # dies too the request itself is what kills them (qhull on a degenerate cluster, memory): it must NOT be computed in
                    # this -- the GPU pipeline's -- process, which the helpers exist to isolate; the caller gets a clear error.
                    import logging
                    log = logging.getLogger('vilgod_amd.boxes')
                    what = req[0] if isinstance(req[0], str) else 'static boxes'
                    seg_ = req[2] if isinstance(req[0], str) else req[1]
                    pts_ = req[1] if isinstance(req[0], str) else req[0]
                    log.warning('box helper %d ended (%s: %s); respawning and retrying a request (%s, %d clusters)',
                                slot, type(e1).__name__, e1, what, len(seg_) - 1)
                    try:
                        self.procs[slot].kill()
                    except Exception:       # noqa: BLE001
                        pass
                    self.procs[slot] = self._spawn()
                    self.respawned += 1
                    try:
                        status, val = self._exchange(self.procs[slot], req)
                    except (BrokenPipeError, EOFError, OSError, ValueError) as e2:
                        try:
                            self.procs[slot].kill()
                        except Exception:   # noqa: BLE001
                            pass
                        self.procs[slot] = self._spawn()
                        self.respawned += 1
                        log.error('box helper %d ended again on the same request (%s, %d clusters, %d points): giving up on it', slot,
                                  what, len(seg_) - 1, len(pts_))
                        raise RuntimeError(f'reference box fit: two helper processes ended on the same request ({what}, {len(seg_) - 1} clusters, '
                                           f'{len(pts_)} points; {type(e2).__name__}: {e2}); run with device.box_workers=0 to fit in-process '
                                           f"or device.box_mode='fast' for the GPU boxes") from e2
                if status != 'ok':
                    raise RuntimeError(f'box helper process: {val}')
                fut.set_result(val)
            except BaseException as e:      # noqa: BLE001  (delivered to the caller of .result())
                fut.set_exception(e)

    def submit(self, *req):
        from concurrent.futures import Future
        fut = Future()
        self.requests.put((fut, req))
        return fut

    def close(self):
        for _ in self.threads:
            self.requests.put(None)
        for p in self.procs:
            try:
                p.stdin.close()
            except Exception:           # noqa: BLE001
                pass
        for p in self.procs:
            try:
                p.wait(timeout=5)
            except Exception:           # noqa: BLE001
                p.kill()
        self.procs, self.threads = [], []


class _Done:
    def __init__(self, value):
        self._v = value

    def result(self):
        return self._v


_POOL = None
_POOL_LOCK = None


def _pool(n_procs):
    global _POOL, _POOL_LOCK
    import threading
    if _POOL_LOCK is None:
        _POOL_LOCK = threading.Lock()
    with _POOL_LOCK:
        if _POOL is None:
            import atexit
            _POOL = BoxWorkerPool(n_procs)
            atexit.register(shutdown_pool)
        else:
            _POOL.grow(n_procs)             # one pool per process; it only ever grows
    return _POOL


def shutdown_pool():
    global _POOL
    if _POOL is not None:
        _POOL.close()
        _POOL = None


def submit_reference_boxes(xy_host, index, seg, zmin, zmax, n_procs=4):
    """-> an object with .result() -> [C,7] boxes.  n_procs = 0 computes in the calling thread."""
    xy_packed = np.ascontiguousarray(xy_host[index, :2], dtype=np.float32)
    seg = np.asarray(seg, dtype=np.int64)
    zmin = np.asarray(zmin, dtype=np.float32).copy()
    zmax = np.asarray(zmax, dtype=np.float32).copy()
    if n_procs <= 0 or len(seg) <= 1:
        return _Done(reference_boxes_packed(xy_packed, seg, zmin, zmax))
    return _pool(n_procs).submit(xy_packed, seg, zmin, zmax)


def submit_moving_boxes(points_list, directions, to_ego_list, centers3, n_procs=4):
    """tracking.moving_boxes of ONE track in a helper process -> object with .result() -> [n,7].  The cluster points of the track's
    entries travel packed ([P,3] float32 + offsets); everything is evaluated by the same function on the same numpy, so the boxes are
    the ones the in-process call returns.  n_procs = 0 (or no pool): computed in the calling thread."""
    from .tracking import moving_boxes, moving_boxes_packed
    if n_procs <= 0:
        return _Done(moving_boxes(points_list, directions, to_ego_list, centers3=centers3))
    seg = np.r_[0, np.cumsum([len(p) for p in points_list])].astype(np.int64)
    xyz = np.ascontiguousarray(np.concatenate([p[:, :3] for p in points_list]))
    return _pool(n_procs).submit('moving_boxes', xyz, seg, np.asarray(directions), np.asarray(to_ego_list),
                                 None if centers3 is None else np.asarray(centers3))


def reference_boxes(xy_host, index, seg, zmin, zmax):
    """Boxes of the packed clusters (index / seg as in frame_state.pack_clusters), reference mode.
    xy_host: [M,>=2] float32 host array of points_ref_wo_ground; zmin / zmax: [C] float32 (vg_cluster_filter stats).
    -> [C,7] float64 [cx,cy,cz,l,w,h,rz] in the reference frame."""
    C = len(seg) - 1
    out = np.empty((C, 7))
    zmin = np.asarray(zmin, dtype=np.float32)
    zmax = np.asarray(zmax, dtype=np.float32)
    for c in range(C):
        pts = xy_host[index[seg[c]:seg[c + 1]], :2]
        corners, rz, _ = minimum_bounding_rectangle(pts)
        out[c] = box_from_rectangle(corners, rz, zmin[c], zmax[c])
    return out
