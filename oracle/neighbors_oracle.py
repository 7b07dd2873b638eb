"""CPU oracle for SURVEY §8f row N1: entropy (PP) scores and the two-frame clustering input.

TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py cpu_baseline).

Restates
  count_neighbors / count_neighbors_inter_frame   src/utils/pointcloud_utils.py:74-107
  compute_ephe_score / calculate_entropy_scores   src/utils/pointcloud_utils.py:110-123
  ZeroShotDetector.calculate_entropy_scores       src/vilgod/zero_shot_detector.py:153-195 (sliding window, < 0.9 cut)
  the n_frames > 1 branch of spatial_clustering   src/vilgod/zero_shot_detector.py:208-242
  knn / knn_labels                                src/utils/pointcloud_utils.py:496-513
  filter_by_ephemeral_score                       src/utils/cluster_utils.py:62-64

PARITY: `compute_ephe_score` and `filter_by_ephemeral_score` are pinned against the reference's own functions
(tests/golden/make_golden.py imports them).  The neighbour searches are **parity unpinned**: the reference calls
CUDA-only third-party ops that cannot run here -- `pcdet.ops.pointnet2.pointnet2_stack.pointnet2_utils.ball_query`
(OpenPCDet, un-vendored, README.md:60-66) and `pytorch3d.ops.knn_points` (README.md:70).  Their published kernels are
restated: ball_query_kernel_stack marks target k a neighbour when
    d2 = (nx-x)*(nx-x) + (ny-y)*(ny-y) + (nz-z)*(nz-z) < radius*radius          (float32)
and stops after nsample hits, so the count the reference derives is min(nsample, #{d2 < r2}); KNearestNeighbor
accumulates `dist += diff * diff` over the three coordinates (float32) and returns squared distances ascending.
Under nvcc's default -fmad=true both compile to fma(dz,dz,fma(dy,dy,dx*dx)); that is the arithmetic used here and in
the HIP kernels.  `np.random.choice` (global Mersenne state) is replaced by a counter-based hash, like the RANSAC.
"""
import numpy as np
from scipy.spatial import cKDTree

from .segment_oracle import mix64, MASK64

ENTROPY_KEEP = 0.9          # zero_shot_detector.py:186
MOVING = 0.6                # zero_shot_detector.py:224


def d2_f32(q, t):
    """float32 fma(dz,dz,fma(dy,dy,dx*dx)) for broadcastable (…,3) arrays (products are exact in float64)."""
    q = np.asarray(q, np.float32); t = np.asarray(t, np.float32)
    d = (q - t).astype(np.float64)          # float32 subtraction, widened
    p0 = (d[..., 0] * d[..., 0]).astype(np.float32).astype(np.float64)
    p1 = (d[..., 1] * d[..., 1] + p0).astype(np.float32).astype(np.float64)
    return (d[..., 2] * d[..., 2] + p1).astype(np.float32)


def _candidates(query, target, r):
    """CSR lists of target candidates within a slightly enlarged radius (exact test follows)."""
    tree = cKDTree(np.asarray(target, np.float64)[:, :3])
    lists = tree.query_ball_point(np.asarray(query, np.float64)[:, :3], r * 1.001 + 1e-6)
    lens = np.fromiter((len(l) for l in lists), np.int64, len(lists))
    flat = np.fromiter((j for l in lists for j in l), np.int64, int(lens.sum()))
    return lens, flat


def ball_count(query, target, r2, cap):
    """min(cap, #{t : d2_f32(q,t) < r2}) per query point."""
    query = np.asarray(query, np.float32)[:, :3]; target = np.asarray(target, np.float32)[:, :3]
    if len(target) == 0 or len(query) == 0:
        return np.zeros(len(query), np.int32)
    r2 = np.float32(r2)
    lens, flat = _candidates(query, target, float(np.sqrt(np.float64(r2))))
    qi = np.repeat(np.arange(len(query)), lens)
    hit = d2_f32(query[qi], target[flat]) < r2
    cnt = np.bincount(qi[hit], minlength=len(query))
    return np.minimum(cnt, cap).astype(np.int32)


def count_neighbors(pts_buffer, seek=0, skip_frames=1, max_neighbor_point_dist=0.3, max_neighbor_points=1000):
    """pointcloud_utils.py:74-94 -> [n_query, n_used_frames] int."""
    skip = skip_frames + 1
    r2 = np.float32(max_neighbor_point_dist) * np.float32(max_neighbor_point_dist)
    cols = []
    for i in list(range(len(pts_buffer)))[::skip]:
        c = ball_count(pts_buffer[seek], pts_buffer[i], r2, max_neighbor_points).astype(np.int64)
        if i == seek:
            c -= 1
        cols.append(c)
    return np.stack(cols).T


def compute_ephe_score(count):
    """pointcloud_utils.py:110-117."""
    N = count.shape[1]
    P = count / (np.expand_dims(count.sum(axis=1), -1) + 1e-8)
    return (-P * np.log(P + 1e-8)).sum(axis=1) / np.log(N)


def window(fnr, length, n_neighbouring_frames):
    """Frames in the sliding buffer and the query's position in it when frame `fnr` is scored
    (zero_shot_detector.py:165-181): the buffer starts at the query frame until the sequence end is reached."""
    n = min(n_neighbouring_frames, length)
    start = min(fnr, max(length - n_neighbouring_frames, 0))
    return list(range(start, start + n)), fnr - start


def entropy_scores_sequence(X_list, n_neighbouring_frames=15, skip_frames=1, max_neighbor_point_dist=0.3,
                            max_neighbor_points=1000):
    """-> per frame (kept scores float64, kept indices) with score < 0.9 (zero_shot_detector.py:183-187)."""
    out = []
    L = len(X_list)
    for fnr in range(L):
        frames, seek = window(fnr, L, n_neighbouring_frames)
        buf = [np.asarray(X_list[f], np.float32)[:, :3] for f in frames]
        H = compute_ephe_score(count_neighbors(buf, seek, skip_frames, max_neighbor_point_dist, max_neighbor_points))
        keep = H < ENTROPY_KEEP
        out.append((H[keep], np.where(keep)[0]))
    return out


def full_scores(n, kept_scores, kept_idx):
    """LidarFrame.entropy_scores (lidar_frame.py:111-118): float32, 1.0 where nothing was stored."""
    e = np.ones(n, np.float32)
    e[kept_idx] = kept_scores
    return e


def subsample_keys(seed, tag, n):
    i = np.arange(n, dtype=np.uint64)
    base = (int(seed) * 0x100000001B3 + (int(tag) << 32)) & MASK64
    return np.array([mix64((base + int(j)) & MASK64) >> 1 for j in i], dtype=np.int64)


def subsample_indices(seed, tag, n, k):
    """the k points with the smallest keys (stable)."""
    return np.sort(np.argsort(subsample_keys(seed, tag, n), kind='stable')[:k])


def two_frame_input(X_list, ent_list, fnr, n_frames=2, seed=0):
    """zero_shot_detector.py:213-239 -> points_seq [m,5] float32 and, per used frame, the point mask."""
    L = len(X_list)
    rng = list(range(min(fnr, L - n_frames), min(fnr + n_frames, L)))
    parts, masks = [], []
    for rel, f in enumerate(rng):
        X = np.asarray(X_list[f], np.float32)
        n = len(X)
        ent = ent_list[f]
        counts = ball_count(X, X, np.float32(0.2) * np.float32(0.2), 100)
        moving = ent < MOVING
        mp = X[moving][:, :3]
        others = ball_count(mp, mp, np.float32(0.1), 4) - 1 if len(mp) else np.zeros(0, np.int32)   # self is always a hit
        dists_mask = np.minimum(others, 3) > 1
        mask = np.zeros(n, bool)
        mask[subsample_indices(seed, f, n, int(n / len(rng)))] = True
        mask[counts < 2] = False
        mask[moving] = False
        mask[moving] |= dists_mask
        parts.append(np.concatenate([X[mask][:, :3], ent[mask, None], np.ones((int(mask.sum()), 1)) * (rel * 0.1)], axis=1))
        masks.append(mask)
    return np.concatenate(parts, dtype=np.float32), masks, rng


def nearest(query, target, max_d2):
    """index of the nearest target (lowest index on ties) with d2_f32 <= max_d2, else -1; and d2 (inf if none)."""
    query = np.asarray(query, np.float32)[:, :3]; target = np.asarray(target, np.float32)[:, :3]
    idx = np.full(len(query), -1, np.int64); best = np.full(len(query), np.inf, np.float32)
    if len(target) == 0 or len(query) == 0:
        return idx, best
    lens, flat = _candidates(query, target, float(np.sqrt(np.float64(max_d2))))
    qi = np.repeat(np.arange(len(query)), lens)
    d2 = d2_f32(query[qi], target[flat])
    ok = d2 <= np.float32(max_d2)
    qi, flat, d2 = qi[ok], flat[ok], d2[ok]
    order = np.lexsort((flat, d2, qi))
    qi, flat, d2 = qi[order], flat[order], d2[order]
    first = np.flatnonzero(np.r_[True, qi[1:] != qi[:-1]])
    idx[qi[first]] = flat[first]; best[qi[first]] = d2[first]
    return idx, best


def knn_labels(points, label_points, labels, probabilities, dist_threshold=0.2):
    """pointcloud_utils.py:505-513 with K=1: label of the nearest label point, -1 beyond the (squared) gate.
    Beyond the gate the probability is irrelevant downstream (label -1); it is reported as 0."""
    gate = np.nextafter(np.float32(dist_threshold), np.float32(0)) if np.float64(np.float32(dist_threshold)) > dist_threshold \
        else np.float32(dist_threshold)
    idx, _ = nearest(points, label_points, gate)
    lab = np.where(idx >= 0, np.asarray(labels)[np.maximum(idx, 0)], -1)
    prob = np.where(idx >= 0, np.asarray(probabilities)[np.maximum(idx, 0)], 0.0)
    return lab, prob


def filter_by_ephemeral_score(scores, percentile=30, min_percentile_pp_score=0.5):
    """cluster_utils.py:62-64: True = moving."""
    return not (np.percentile(scores, percentile) > min_percentile_pp_score)
