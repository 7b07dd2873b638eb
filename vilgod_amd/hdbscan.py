"""`HDBSCAN` with the constructor and result attributes the reference uses
(tools/configs/preprocessor/waymo.yaml:10-15 -> `hdbscan.HDBSCAN(cluster_selection_epsilon=0.15,
min_cluster_size=15, metric='euclidean', core_dist_n_jobs=-1)`; `.fit(X)` then `.labels_`,
`.probabilities_`, src/vilgod/zero_shot_detector.py:248-250), running on the GPU:

    core distances + exact mutual-reachability MST + edge sort   csrc/cluster.hip   (GPU)
    single linkage / condense / EOM / epsilon / labels           csrc/hdbscan_tree.cpp (host, C++), or -- `hierarchy='device'` --
                                                                 csrc/hdbscan_device.hip (GPU, the same results bit for bit)

`fit` accepts a numpy array (reference call) or a CUDA float32 tensor (fused pipeline).
"""
import ctypes

import numpy as np
import torch

from ._lib import lib, ptr, stream_ptr, check


class HDBSCAN:
    def __init__(self, min_cluster_size=5, min_samples=None, cluster_selection_epsilon=0.0, metric='euclidean',
                 core_dist_n_jobs=None, max_points=400_000, device='cuda', hierarchy='host', **unused):
        if metric != 'euclidean':
            raise NotImplementedError('only the euclidean metric of the reference configuration is implemented')
        for k in unused:
            if k not in ('alpha', 'algorithm', 'leaf_size', 'approx_min_span_tree', 'gen_min_span_tree',
                         'cluster_selection_method', 'allow_single_cluster', 'prediction_data', 'memory'):
                raise TypeError(f'unexpected keyword {k}')
        if unused.get('cluster_selection_method', 'eom') != 'eom' or unused.get('allow_single_cluster', False):
            raise NotImplementedError('cluster_selection_method=eom, allow_single_cluster=False only')
        self.min_cluster_size = int(min_cluster_size)
        self.min_samples = int(min_samples) if min_samples is not None else self.min_cluster_size
        if not 1 <= self.min_samples <= 15:
            raise NotImplementedError('min_samples must be in [1, 15] (register-resident neighbour list)')
        self.cluster_selection_epsilon = float(cluster_selection_epsilon)
        self.device = torch.device(device)
        self.max_points = int(max_points)
        # `fit`'s hierarchy stage: 'host' (csrc/hdbscan_tree.cpp) or 'device' (csrc/hdbscan_device.hip: the tree stays on the GPU);
        # identical results.  (The fused pipeline chooses for itself: PseudoLabelPipeline(hierarchy=...).)
        if hierarchy not in ('host', 'device'):
            raise ValueError("hierarchy: 'host' or 'device'")
        if self.min_cluster_size > 32 or self.max_points > (1 << 20):       # what the device stage holds (include/vilgod_hip.h)
            hierarchy = 'host'
        self.hierarchy = hierarchy
        self._hier = None
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.vg_cluster_create(ctypes.byref(h), self.max_points), 'vg_cluster_create')
        self._h = h
        self.labels_ = None
        self.probabilities_ = None
        self.n_rounds_ = 0

    def __del__(self):
        h = getattr(self, '_h', None)
        if h is not None and lib is not None:
            lib.vg_cluster_destroy(h)
            self._h = None

    # ---- fixed-radius queries on the same cell grid -----------------------------------
    def grid(self, T, stream=None):
        """Build the handle's cell grid over target points T (CUDA float32 [n,>=3]); valid until the next grid()/mst()."""
        assert T.is_cuda and T.dtype == torch.float32 and (T.shape[0] == 0 or T.stride(1) == 1)
        check(lib.vg_cluster_grid(self._h, ptr(T), T.shape[0], T.stride(0) if T.shape[0] else 3, stream_ptr(stream)),
              'vg_cluster_grid')
        self._grid_ref = T                      # the kernels read the handle's own sorted copy; kept for clarity only

    def ball_count(self, Q, r2, cap, out=None, stream=None):
        """-> int32 [nq]: min(cap, #{grid points with float32 d2 < r2}) per query row of Q (CUDA float32 [nq,>=3])."""
        assert Q.is_cuda and Q.dtype == torch.float32 and (Q.shape[0] == 0 or Q.stride(1) == 1)
        nq = Q.shape[0]
        if out is None:
            out = torch.empty(nq, dtype=torch.int32, device=Q.device)
        check(lib.vg_cluster_ball_count(self._h, ptr(Q), nq, Q.stride(0) if nq else 3, float(np.float32(r2)), int(cap), ptr(out),
                                        stream_ptr(stream)), 'vg_cluster_ball_count')
        return out

    def nearest(self, Q, max_d2, stream=None):
        """-> (idx int32 [nq] row of the grid's point array or -1, d2 float32 [nq]) of the nearest grid point with
        float32 d2 <= max_d2."""
        assert Q.is_cuda and Q.dtype == torch.float32 and (Q.shape[0] == 0 or Q.stride(1) == 1)
        nq = Q.shape[0]
        idx = torch.empty(nq, dtype=torch.int32, device=Q.device)
        d2 = torch.empty(nq, dtype=torch.float32, device=Q.device)
        check(lib.vg_cluster_nearest(self._h, ptr(Q), nq, Q.stride(0) if nq else 3, float(np.float32(max_d2)), ptr(idx), ptr(d2),
                                     stream_ptr(stream)), 'vg_cluster_nearest')
        return idx, d2

    # ---- GPU stage ---------------------------------------------------------------------------------
    def mst(self, X, want_core=False, stream=None, dim=3):
        """X: CUDA float32 [n,>=dim]; the first `dim` (3..5) columns are the clustering space.
        -> (lo int32[n-1], hi int32[n-1], w2 float64[n-1]) CUDA, sorted by w2 (+ squared core distances in input order)."""
        assert X.is_cuda and X.dtype == torch.float32 and X.stride(1) == 1 and 3 <= dim <= 5 and X.shape[1] >= dim
        n = X.shape[0]
        m = max(n - 1, 0)
        dev = X.device
        lo = torch.empty(m, dtype=torch.int32, device=dev)
        hi = torch.empty(m, dtype=torch.int32, device=dev)
        w2 = torch.empty(m, dtype=torch.float64, device=dev)
        core2 = torch.empty(n, dtype=torch.float64, device=dev) if want_core else None
        rounds = ctypes.c_int32(0)
        check(lib.vg_cluster_mst_nd(self._h, ptr(X), n, X.stride(0), int(dim), self.min_samples, ptr(core2), ptr(lo), ptr(hi),
                                    ptr(w2), ctypes.byref(rounds), stream_ptr(stream)), 'vg_cluster_mst_nd')
        self.n_rounds_ = rounds.value
        return (lo, hi, w2, core2) if want_core else (lo, hi, w2)

    # ---- host stage ----------------------------------------------------------------------------------
    def tree(self, lo, hi, w2, n):
        """numpy (host) sorted MST -> labels int32 [n], probabilities float64 [n]."""
        labels = np.empty(n, np.int32)
        probs = np.empty(n, np.float64)
        nc = ctypes.c_int32(0)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        check(lib.vg_hdbscan_tree_host(p(lo), p(hi), p(w2), n, self.min_cluster_size, self.cluster_selection_epsilon,
                                       p(labels), p(probs), ctypes.byref(nc)), 'vg_hdbscan_tree_host')
        return labels, probs, nc.value

    def tree_device(self, lo, hi, w2, n, stream=None):
        """CUDA sorted MST (as `mst` returns it) -> (labels int32 [n], probabilities float64 [n], n_clusters int32 [1]) CUDA tensors, queued on
        the stream, nothing waited for: the hierarchy stage as kernels (csrc/hdbscan_device.hip), the same results as `tree`."""
        if self._hier is None:
            self._hier = DeviceHierarchy(max_points=self.max_points, device=self.device)
        return self._hier.tree_async(lo, hi, w2, n, self.min_cluster_size, self.cluster_selection_epsilon, stream)

    def fit(self, X, dim=None):
        """numpy input: every column is a clustering coordinate, like the library (3 = `points_ref_wo_ground[..., :3]`,
        zero_shot_detector.py:246-248; 5 = the two-frame `points_seq`, :239-241).  CUDA tensor input (fused pipeline):
        point rows, the first `dim` (default 3) columns are used."""
        if isinstance(X, np.ndarray):
            if dim is None:
                dim = X.shape[1]
            if not 3 <= dim <= 5:
                raise NotImplementedError('3 to 5 clustering coordinates (x, y, z first)')
            Xd = torch.from_numpy(np.ascontiguousarray(X[:, :dim], dtype=np.float32)).to(self.device)
        else:
            Xd = X if (X.dtype == torch.float32 and X.stride(1) == 1) else X.float().contiguous()
            dim = 3 if dim is None else dim
        n = Xd.shape[0]
        if n < 2:
            self.labels_ = np.full(n, -1, np.int64)
            self.probabilities_ = np.zeros(n)
            return self
        lo, hi, w2 = self.mst(Xd, dim=dim)
        if self.hierarchy == 'device':
            d_labels, d_probs, _ = self.tree_device(lo, hi, w2, n)
            labels, probs = d_labels.cpu().numpy(), d_probs.cpu().numpy()
        else:
            labels, probs, _ = self.tree(lo.cpu().numpy(), hi.cpu().numpy(), w2.cpu().numpy(), n)
        self.labels_ = labels.astype(np.int64)
        self.probabilities_ = probs
        return self


class DeviceHierarchy:
    """The hierarchy stage on the device (csrc/hdbscan_device.hip): the tree `HDBSCAN.mst` returns -> labels int32 [n], probabilities
    float64 [n] (CUDA tensors) and the cluster count, equal bit for bit to `HDBSCAN.tree` on the same tree.  Holds the stage's buffers
    for trees of up to `max_points` points; one call at a time per object."""

    def __init__(self, max_points=400_000, device='cuda'):
        self.device = torch.device(device)
        self.max_points = int(max_points)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.vg_hier_create(ctypes.byref(h), self.max_points), 'vg_hier_create')
        self._h = h

    def __del__(self):
        h = getattr(self, '_h', None)
        if h is not None and lib is not None:
            lib.vg_hier_destroy(h)
            self._h = None

    def tree_async(self, lo, hi, w2, n, min_cluster_size, eps, stream=None):
        """-> (labels, probs, n_clusters) CUDA tensors, queued on `stream` (nothing waited for)."""
        dev = self.device
        labels = torch.empty(n, dtype=torch.int32, device=dev)
        probs = torch.empty(n, dtype=torch.float64, device=dev)
        nc = torch.zeros(1, dtype=torch.int32, device=dev)
        check(lib.vg_hdbscan_tree_device(self._h, ptr(lo), ptr(hi), ptr(w2), int(n), int(min_cluster_size), float(eps), ptr(labels), ptr(probs),
                                         ptr(nc), stream_ptr(stream)), 'vg_hdbscan_tree_device')
        return labels, probs, nc

    def tree(self, lo, hi, w2, n, min_cluster_size, eps, stream=None):
        labels, probs, nc = self.tree_async(lo, hi, w2, n, min_cluster_size, eps, stream)
        return labels, probs, int(nc.item())
