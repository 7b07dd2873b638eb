"""Entropy (PP) scores over a sliding window of frames and the two-frame clustering input -- SURVEY §8f row N1, the
reference's DEFAULT configuration (tools/configs/preprocessing.yaml:57-68).

Mirrors
  ZeroShotDetector.calculate_entropy_scores   src/vilgod/zero_shot_detector.py:153-195
  pointcloud_utils.count_neighbors / compute_ephe_score / calculate_entropy_scores   src/utils/pointcloud_utils.py:74-123
  the n_frames > 1 branch of ZeroShotDetector.spatial_clustering   src/vilgod/zero_shot_detector.py:208-242
  pointcloud_utils.knn_labels   src/utils/pointcloud_utils.py:505-513

All neighbour searches run on the GPU through the cell grid of a `vilgod_amd.hdbscan.HDBSCAN` handle
(csrc/cluster.hip: vg_cluster_grid / vg_cluster_ball_count / vg_cluster_nearest); the scores come from
vg_entropy_scores, the random half-sample from vg_subsample_keys (counter-based; the reference draws from numpy's global
Mersenne state, which no implementation can reproduce).
"""
import ctypes

import numpy as np
import torch

from ._lib import lib, ptr, stream_ptr, check

ENTROPY_KEEP = 0.9        # zero_shot_detector.py:186: only scores below 0.9 are stored
MOVING = 0.6              # zero_shot_detector.py:224: entropy < 0.6 = moving point


def window(fnr, length, n_neighbouring_frames):
    """Frames in the reference's buffer when frame `fnr` is scored, and the query's position (`seek`) in it
    (zero_shot_detector.py:165-181): the window starts at the query frame until it reaches the end of the sequence."""
    n = min(n_neighbouring_frames, length)
    start = min(fnr, max(length - n_neighbouring_frames, 0))
    return list(range(start, start + n)), fnr - start


def spatial_order(X, cell=0.4):
    """Permutation that sorts the points of X (CUDA [n,>=3]) by 4 x 4-cell column, height, cell within the column."""
    c = torch.floor(X[:, :3] * (1.0 / cell)).to(torch.int64) + (1 << 18)
    cx, cy, cz = c[:, 0], c[:, 1], c[:, 2] & 0xFFFF
    key = ((cy >> 2) << 40) | ((cx >> 2) << 20) | (cz << 4) | ((cy & 3) << 2) | (cx & 3)
    return torch.sort(key).indices


class EntropyScorer:
    def __init__(self, grid_model, n_neighbouring_frames=15, skip_frames=1, max_neighbor_point_dist=0.3,
                 max_neighbor_points=1000, **unused):
        self.grid_model = grid_model
        self.n_neighbouring_frames = int(n_neighbouring_frames)
        self.skip = int(skip_frames) + 1
        r = np.float32(max_neighbor_point_dist)
        self.r2 = np.float32(r * r)                     # ball_query: radius2 = radius * radius in float32
        self.cap = int(max_neighbor_points)

    def frames_needed(self, queries, length):
        """Frames whose points must be resident to score the `queries`."""
        need = set()
        for f in queries:
            need.update(window(f, length, self.n_neighbouring_frames)[0][::self.skip])
            need.add(f)
        return sorted(need)

    def plan(self, X_list, queries=None):
        """The work of `score_sequence` laid out so that it can also run piecewise (the stage dispatcher streams it through the frame
        pass): per-query count buffers, the spatially coherent query order, and for every TARGET frame j the (query, row) pairs that
        count their neighbours in it.  -> dict."""
        L = len(X_list)
        as_list = queries is None
        queries = list(range(L)) if queries is None else sorted(queries)
        wins = {f: window(f, L, self.n_neighbouring_frames) for f in queries}
        used = {f: wins[f][0][::self.skip] for f in queries}         # pointcloud_utils.py:81 idx_list[::skip]
        if any(len(u) < 2 for u in used.values()):
            raise NotImplementedError('entropy scores need at least two neighbouring frames')
        # queries run in a spatially coherent order (4 x 4-cell columns of the 0.4 m grid): a wave's 64 query points then read the
        # same few cells of the target grid, whatever order the data set stores its points in; the scores are put back at the end
        order = {f: spatial_order(X_list[f]) for f in queries}
        Xq = {f: X_list[f].index_select(0, order[f])[:, :3].contiguous() for f in queries}
        counts = {f: torch.zeros((len(used[f]), X_list[f].shape[0]), dtype=torch.int32, device=X_list[f].device)
                  for f in queries}
        users = {}
        for f in queries:
            for col, j in enumerate(used[f]):
                users.setdefault(j, []).append((f, col))
        return dict(X_list=X_list, L=L, as_list=as_list, queries=queries, wins=wins, used=used, order=order, Xq=Xq, counts=counts, users=users)

    def run_target(self, model, plan, j):
        """Grid of target frame j (on `model`'s handle, current stream) and the neighbour counts of every query frame whose window holds it."""
        model.grid(plan['X_list'][j])
        for f, col in plan['users'][j]:
            model.ball_count(plan['Xq'][f], self.r2, self.cap, out=plan['counts'][f][col])

    def finish_query(self, plan, f):
        """Entropy scores of query frame f once all targets of its window have run: float64 CUDA [n] in the frame's point order."""
        frames, seek = plan['wins'][f]
        used = plan['used'][f]
        seek_row = used.index(frames[seek]) if frames[seek] in used else -1
        X = plan['X_list'][f]
        n = X.shape[0]
        H = torch.empty(n, dtype=torch.float64, device=X.device)
        check(lib.vg_entropy_scores(ptr(plan['counts'][f]), len(used), n, seek_row, ptr(H), stream_ptr()), 'vg_entropy_scores')
        return torch.empty_like(H).index_copy_(0, plan['order'][f], H)          # back to the frame's point order

    def score_sequence(self, X_list, queries=None, mapper=None):
        """X_list: per frame CUDA float32 [n,>=3] (`points_ref_wo_ground`; entries of frames that are not needed may be
        None).  -> {fnr: float64 CUDA tensor [n] of entropy scores} for the `queries` (default: every frame; then a
        list).  Every frame's grid is built once and queried by all the frames whose window contains it.
        mapper(items, fn): runs fn(grid_model, item) for the items on several handles / streams at once
        (PseudoLabelPipeline.map_workers); target frames are independent, every (query, column) row has one writer."""
        plan = self.plan(X_list, queries)
        if mapper is None:
            for j in sorted(plan['users']):
                self.run_target(self.grid_model, plan, j)
        else:
            mapper(sorted(plan['users']), lambda model, j: self.run_target(model, plan, j))
        out = {f: self.finish_query(plan, f) for f in plan['queries']}
        return [out[f] for f in range(plan['L'])] if plan['as_list'] else out

    @staticmethod
    def reduce(H):
        """zero_shot_detector.py:186-187: keep scores < 0.9 -> (values float64, indices int64) on the host."""
        keep = torch.nonzero(H < ENTROPY_KEEP).squeeze(1)
        return H.index_select(0, keep).cpu().numpy(), keep.cpu().numpy()


def full_scores(n, kept_scores, kept_idx, device=None):
    """LidarFrame.entropy_scores (lidar_frame.py:111-118): float32 [n], 1.0 where nothing was stored."""
    e = np.ones(n, np.float32)
    e[kept_idx] = kept_scores
    return e if device is None else torch.from_numpy(e).to(device)


class TwoFrameClusterer:
    """spatial_clustering with n_frames > 1 (zero_shot_detector.py:208-242)."""

    def __init__(self, cluster_model, n_frames=2, seed=0, dist_threshold=0.2, parts=None):
        self.model = cluster_model
        self._parts = parts                                  # precomputed frame parts (read-only), see precompute_parts
        self.n_frames = int(n_frames)
        self.seed = int(seed)
        g = np.float32(dist_threshold)                      # `dists > 0.2` on float32 squared distances (float64 compare)
        self.gate = float(np.nextafter(g, np.float32(0)) if np.float64(g) > dist_threshold else g)
        self._cache = {}

    def reset(self):
        self._cache = {}

    def precompute_parts(self, X_list, ent_list, mapper=None):
        """Every frame's clustering rows once, up front (a frame enters the input of two consecutive query frames): the
        returned dict can be handed to other TwoFrameClusterer instances (worker threads) through `parts=`.
        mapper: as in EntropyScorer.score_sequence (frames are independent)."""
        L = len(X_list)
        n_used = min(self.n_frames, L)
        if mapper is None:
            return {(f, n_used): self.frame_part(f, X_list[f], ent_list[f], n_used) for f in range(L)}
        rows = mapper(range(L), lambda model, f: TwoFrameClusterer(model, n_frames=self.n_frames, seed=self.seed)
                      .frame_part(f, X_list[f], ent_list[f], n_used))
        return {(f, n_used): r for f, r in enumerate(rows)}

    def frame_part(self, f, X, ent, n_used):
        """Rows of frame f that enter the clustering input: [x, y, z, entropy] (CUDA float32 [m,4]).
        X: CUDA float32 [n,>=3]; ent: CUDA float32 [n] (full entropy array)."""
        key = (f, n_used)
        if self._parts is not None and key in self._parts:
            return self._parts[key]
        if key in self._cache:
            return self._cache[key]
        n = X.shape[0]
        dev = X.device
        m = self.model
        m.grid(X)
        o = spatial_order(X)                                                          # coherent query order, see score_sequence
        counts = torch.empty(n, dtype=torch.int32, device=dev).index_copy_(
            0, o, m.ball_count(X.index_select(0, o)[:, :3].contiguous(), np.float32(0.2) * np.float32(0.2), 100))   # count_neighbors_inter_frame(points, 0.2)
        moving = ent < MOVING
        mi = torch.nonzero(moving).squeeze(1)
        mask = torch.zeros(n, dtype=torch.bool, device=dev)
        k = int(n / n_used)
        if k > 0:
            keys = torch.empty(n, dtype=torch.int64, device=dev)
            check(lib.vg_subsample_keys(self.seed, int(f), n, ptr(keys), stream_ptr()), 'vg_subsample_keys')
            mask[torch.sort(keys, stable=True).indices[:k]] = True
        mask &= counts >= 2
        mask[mi] = False
        if mi.numel():
            mp = X.index_select(0, mi).contiguous()
            m.grid(mp)
            others = m.ball_count(mp, np.float32(0.1), 4) - 1       # 3 nearest OTHER moving points with d2 < 0.1 (self always hits)
            mask[mi] = others >= 2                                   # np.sum(dists < 0.1, axis=1) > 1
        rows = torch.nonzero(mask).squeeze(1)
        part = torch.cat([X.index_select(0, rows)[:, :3], ent.index_select(0, rows)[:, None]], dim=1)
        if self._parts is None:
            self._cache = {kk: v for kk, v in self._cache.items() if kk[0] >= f - 1}
        self._cache[key] = part
        return part

    def used_frames(self, fnr, length):
        return list(range(min(fnr, length - self.n_frames), min(fnr + self.n_frames, length)))

    def cluster_input(self, fnr, X_list, ent_list):
        rng = self.used_frames(fnr, len(X_list))
        parts = []
        for rel, f in enumerate(rng):
            p = self.frame_part(f, X_list[f], ent_list[f], len(rng))
            t = torch.full((p.shape[0], 1), rel * 0.1, dtype=torch.float32, device=p.device)
            parts.append(torch.cat([p, t], dim=1))
        return torch.cat(parts, dim=0).contiguous()

    def labels(self, fnr, X_list, ent_list):
        """-> (labels int64 [n], probabilities float64 [n]) for the points of frame fnr (host)."""
        if len(X_list) < self.n_frames:
            raise ValueError('sequence shorter than n_frames')
        seq = self.cluster_input(fnr, X_list, ent_list)
        n = X_list[fnr].shape[0]
        m = seq.shape[0]
        if m < 2:
            return np.full(n, -1, np.int64), np.zeros(n)
        lo, hi, w2 = self.model.mst(seq, dim=5)
        on_device = getattr(self.model, 'hierarchy', 'host') == 'device'
        if on_device:        # the hierarchy stage as kernels: the tree stays on the GPU, the label transfer below is a device gather
            d_lab, d_prob, _ = self.model.tree_device(lo, hi, w2, m)
        else:
            lab_seq, prob_seq, _ = self.model.tree(lo.cpu().numpy(), hi.cpu().numpy(), w2.cpu().numpy(), m)
        self.model.grid(seq)
        Xf = X_list[fnr]
        o = spatial_order(Xf)
        idx_s, _ = self.model.nearest(Xf.index_select(0, o)[:, :3].contiguous(), self.gate)
        if on_device:
            d_idx = torch.empty_like(idx_s).index_copy_(0, o, idx_s).long()
            ok = d_idx >= 0
            g = d_idx.clamp_min(0)
            labels = torch.where(ok, d_lab.index_select(0, g).long(), torch.full_like(g, -1))
            probs = torch.where(ok, d_prob.index_select(0, g), torch.zeros_like(d_prob[:1]).expand_as(g))
            return labels.cpu().numpy(), probs.cpu().numpy()
        idx = torch.empty_like(idx_s).index_copy_(0, o, idx_s).cpu().numpy().astype(np.int64)
        ok = idx >= 0
        labels = np.where(ok, lab_seq[np.maximum(idx, 0)], -1).astype(np.int64)
        probs = np.where(ok, prob_seq[np.maximum(idx, 0)], 0.0)
        return labels, probs
