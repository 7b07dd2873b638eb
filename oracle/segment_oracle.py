"""CPU oracle for the per-cluster host logic.

TEST INFRASTRUCTURE ONLY.  Plain numpy/scipy restatements; each function cites the reference lines it
follows (paths relative to /root/reference).  tests/golden/make_golden.py pins B3/C1/E1/F1/D10 against the
reference's own code (stub-imported) on seeded inputs -> tests/golden/detect_golden.npz.

C2 is PARITY UNPINNED: the reference fits the ground plane with `pyransac3d` (un-vendored, unpinned,
README.md:79) driven by python's global `random` state.  The restatement keeps the algorithm and replaces the
RNG by a counter-based hash so that CPU and GPU draw the same samples.
"""
import numpy as np
from scipy import spatial
from scipy.spatial.transform import Rotation as R

MASK64 = (1 << 64) - 1


# ---- B1 ---------------------------------------------------------------------------------------------
def apply_transform(pts, T, box=False):
    """pointcloud_utils.py:21-46 (numpy branch, mode='left')."""
    if len(pts) == 0:
        return pts
    out = pts.copy()
    h = np.hstack((out[:, :3], np.ones((len(out), 1))))
    out[..., :3] = np.einsum('ij,kj->ki', T, h)[..., :3]
    if box:
        out[..., 6] += R.from_matrix(T[:3, :3]).as_euler('xyz')[-1]
    return out


# ---- C2 ---------------------------------------------------------------------------------------------
def mix64(z):
    z = (z + 0x9E3779B97F4A7C15) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def sample3(seed, it, n):
    s, j = [], 0
    while len(s) < 3:
        v = mix64((seed * 0x100000001B3 + (it << 20) + j) & MASK64) % n
        j += 1
        if v not in s:
            s.append(int(v))
    return s


def plane_ransac(pts, thresh=0.1, iters=100, seed=0):
    """pyransac3d.Plane.fit(pts, thresh, maxIteration=iters) with hashed sample indices.  float64.
    Returns (plane [a,b,c,d], inlier index array)."""
    P = np.asarray(pts, dtype=np.float64)[:, :3]
    n = len(P)
    best_eq, best_cnt, best_in = np.zeros(4), 0, np.zeros(0, int)
    for it in range(iters):
        s = sample3(seed, it, n)
        A, B = P[s[1]] - P[s[0]], P[s[2]] - P[s[0]]
        C = np.array([A[1] * B[2] - A[2] * B[1], A[2] * B[0] - A[0] * B[2], A[0] * B[1] - A[1] * B[0]])
        with np.errstate(all='ignore'):
            C = C / np.sqrt((C[0] * C[0] + C[1] * C[1]) + C[2] * C[2])
            d = -((C[0] * P[s[1], 0] + C[1] * P[s[1], 1]) + C[2] * P[s[1], 2])
            dist = (((C[0] * P[:, 0] + C[1] * P[:, 1]) + C[2] * P[:, 2]) + d) / np.sqrt((C[0] * C[0] + C[1] * C[1]) + C[2] * C[2])
            inl = np.where(np.abs(dist) <= thresh)[0]
        if len(inl) > best_cnt:
            best_eq, best_cnt, best_in = np.array([C[0], C[1], C[2], d]), len(inl), inl
    return best_eq, best_in


def fit_plane(points, seed=0, threshold=0.1, max_iteration=100):
    """pointcloud_utils.py:375-387: two RANSAC stages (all points, then the inliers), normal flipped to +z."""
    eq1, in1 = plane_ransac(points, 0.1, max_iteration, seed)
    eq2, _ = plane_ransac(np.asarray(points)[in1], threshold, max_iteration, seed + 1)
    if eq2[2] < 0:
        eq2 = eq2 * -1
    return eq2


# ---- B3 ---------------------------------------------------------------------------------------------
def generate_detections(labels, probs, threshold=0.3):
    """lidar_frame.py:163-167, 230-237: list of (cluster_id, point index array) in ascending id order."""
    labels = np.array(labels, copy=True)
    labels[np.asarray(probs) < threshold] = -1
    ids = np.sort(np.unique(labels[labels != -1]))
    return [(int(c), np.where(labels == c)[0]) for c in ids]


# ---- B4 + C1 ------------------------------------------------------------------------------------------
def filter_cluster(points, plane, min_points=10, max_points=999999, max_min_height=1.0, min_max_height=0.5,
                   min_height=0.3, max_height=6):
    """objects.py:158-181 with the three active filters (waymo.yaml:16-49), all `and` + required:
    cluster_utils.py:14-15, :48-49 (height = objects.py:112-114), :51-60."""
    p3 = points[..., :3]
    ok_n = (p3.shape[0] >= min_points) & (p3.shape[0] <= max_points)
    height = np.max(points[..., 2]) - np.min(points[..., 2])
    ok_h = (height >= min_height) & (height <= max_height)
    d = p3 @ plane[:3] + plane[3]
    d = d / np.sqrt((plane[:3] ** 2).sum())
    ok_p = (d.min() <= max_min_height) & (d.max() >= min_max_height)
    return bool(ok_n and ok_p and ok_h), (p3.shape[0], float(np.min(points[..., 2])), float(np.max(points[..., 2])),
                                          float(d.min()), float(d.max()), float(height))


# ---- E1 ---------------------------------------------------------------------------------------------
def minimum_bounding_rectangle(points, all_edges=False):
    """pointcloud_utils.py:309-372.  all_edges=False is the reference (closing hull edge dropped, :329-330);
    all_edges=True also tries the closing edge (what the GPU kernel does)."""
    pi2 = np.pi / 2.
    try:
        hull_points = points[spatial.ConvexHull(points).vertices]
    except Exception:
        corners = np.ones((4, 2)) * np.mean(points[:, :2], axis=0)[:2]
        corners += np.array([[-0.05, -0.05], [0.05, -0.05], [0.05, 0.05], [-0.05, 0.05]])
        return corners, 0, 0
    hp = np.concatenate([hull_points, hull_points[:1]]) if all_edges else hull_points
    edges = hp[1:] - hp[:-1]
    angles = np.arctan2(edges[:, 1], edges[:, 0])
    angles = np.abs(np.mod(angles, pi2))
    angles = np.unique(angles)
    rotations = np.vstack([np.cos(angles), np.cos(angles - pi2), np.cos(angles + pi2), np.cos(angles)]).T
    rotations = rotations.reshape((-1, 2, 2))
    rot_points = np.dot(rotations, hull_points.T)
    min_x, max_x = np.nanmin(rot_points[:, 0], axis=1), np.nanmax(rot_points[:, 0], axis=1)
    min_y, max_y = np.nanmin(rot_points[:, 1], axis=1), np.nanmax(rot_points[:, 1], axis=1)
    areas = (max_x - min_x) * (max_y - min_y)
    b = np.argmin(areas)
    x1, x2, y1, y2, r = max_x[b], min_x[b], max_y[b], min_y[b], rotations[b]
    rval = np.zeros((4, 2))
    rval[0] = np.dot([x1, y2], r)
    rval[1] = np.dot([x2, y2], r)
    rval[2] = np.dot([x2, y1], r)
    rval[3] = np.dot([x1, y1], r)
    return rval, angles[b], areas[b]


def fit_box(cluster_points, all_edges=False):
    """zero_shot_detector.py:450-461 (static branch): [cx,cy,cz,l,w,h+0.3,rz]."""
    corners, rz, area = minimum_bounding_rectangle(cluster_points[:, :2], all_edges)
    l = np.linalg.norm(corners[0] - corners[1])
    w = np.linalg.norm(corners[0] - corners[-1])
    c = (corners[0] + corners[2]) / 2
    if w > l:
        l, w = w, l
        rz += np.pi / 2
    height = cluster_points[:, 2].max() - cluster_points[:, 2].min()
    return np.array([c[0], c[1], cluster_points[:, 2].min() + height / 2, l, w, height + 0.3, rz])


def box_corners_bev(box):
    """4 BEV corners of [cx,cy,cz,l,w,h,rz] (for orientation-agnostic comparisons: rz is only defined mod pi/2
    up to an l/w swap)."""
    cx, cy, _, l, w, _, rz = box
    c, s = np.cos(rz), np.sin(rz)
    d = np.array([[l / 2, w / 2], [-l / 2, w / 2], [-l / 2, -w / 2], [l / 2, -w / 2]])
    return d @ np.array([[c, s], [-s, c]]) + [cx, cy]
