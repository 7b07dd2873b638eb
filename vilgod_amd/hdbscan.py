"""`HDBSCAN` with the constructor and result attributes the reference uses
(tools/configs/preprocessor/waymo.yaml:10-15 -> `hdbscan.HDBSCAN(cluster_selection_epsilon=0.15,
min_cluster_size=15, metric='euclidean', core_dist_n_jobs=-1)`; `.fit(X)` then `.labels_`,
`.probabilities_`, src/vilgod/zero_shot_detector.py:248-250), running on the GPU:

    core distances + exact mutual-reachability MST + edge sort   csrc/cluster.hip   (GPU)
    single linkage / condense / EOM / epsilon / labels           csrc/hdbscan_tree.cpp (host, C++)

`fit` accepts a numpy array (reference call) or a CUDA float32 tensor (fused pipeline).
"""
import ctypes

import numpy as np
import torch

from ._lib import lib, ptr, stream_ptr, check


class HDBSCAN:
    def __init__(self, min_cluster_size=5, min_samples=None, cluster_selection_epsilon=0.0, metric='euclidean',
                 core_dist_n_jobs=None, max_points=400_000, device='cuda', **unused):
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

    # ---- GPU stage ---------------------------------------------------------------------------------
    def mst(self, X, want_core=False, stream=None):
        """X: CUDA float32 [n,>=3].  -> (lo int32[n-1], hi int32[n-1], w2 float64[n-1]) CUDA, sorted by w2
        (+ squared core distances in input order)."""
        assert X.is_cuda and X.dtype == torch.float32 and X.stride(1) == 1
        n = X.shape[0]
        m = max(n - 1, 0)
        dev = X.device
        lo = torch.empty(m, dtype=torch.int32, device=dev)
        hi = torch.empty(m, dtype=torch.int32, device=dev)
        w2 = torch.empty(m, dtype=torch.float64, device=dev)
        core2 = torch.empty(n, dtype=torch.float64, device=dev) if want_core else None
        rounds = ctypes.c_int32(0)
        check(lib.vg_cluster_mst(self._h, ptr(X), n, X.stride(0), self.min_samples, ptr(core2), ptr(lo), ptr(hi), ptr(w2),
                                 ctypes.byref(rounds), stream_ptr(stream)), 'vg_cluster_mst')
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

    def fit(self, X):
        if isinstance(X, np.ndarray):
            Xd = torch.from_numpy(np.ascontiguousarray(X[:, :3], dtype=np.float32)).to(self.device)
        else:
            Xd = X if (X.dtype == torch.float32 and X.stride(1) == 1) else X.float().contiguous()
        n = Xd.shape[0]
        if n < 2:
            self.labels_ = np.full(n, -1, np.int64)
            self.probabilities_ = np.zeros(n)
            return self
        lo, hi, w2 = self.mst(Xd)
        labels, probs, _ = self.tree(lo.cpu().numpy(), hi.cpu().numpy(), w2.cpu().numpy(), n)
        self.labels_ = labels.astype(np.int64)
        self.probabilities_ = probs
        return self
