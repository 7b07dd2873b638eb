"""Spatial clustering parity.

The arithmetic of the reference lives in the un-vendored, unpinned `hdbscan` package, which is absent here
-> the oracle (oracle/hdbscan_oracle.py) restates the published algorithm and is pinned against the
scikit-learn implementation of the same algorithm:
  * mutual-reachability MST weights: bit-identical multiset to scikit-learn's (`_single_linkage_tree_`)
  * tree stages: identical labels/probabilities to scikit-learn's own `tree_to_labels` on the same linkage
  * end to end: adjusted Rand index >= 0.99 and >= 99 % agreement on the noise set (equal-weight MST edges are
    ordered arbitrarily by both libraries; the oracle and the GPU use the strict order (w2, lo, hi))
Product side on CPU: the host hierarchy stage (csrc/hdbscan_tree.cpp) == the oracle's, exactly.
GPU: see the gpu-marked tests below (exact equality with the oracle: labels, probabilities, MST).
"""
import ctypes

import numpy as np
import pytest

from oracle import hdbscan_oracle as ho
from oracle import patchworkpp as opw
from vilgod_amd import synthetic


def lidar_scene(seed, n):
    pts = synthetic.make_frame(seed, n, n_objects=max(4, n // 2500))
    params = opw.Parameters()
    params.min_range = 1.5
    idx = opw.mask_ground_points(pts, opw.patchworkpp(params), 1.723)
    m = np.ones(len(pts), bool)
    m[idx] = False
    return pts[m, :3]


def blob_scene(seed):
    rng = np.random.default_rng(seed)
    k = rng.integers(5, 12)
    cents = rng.uniform(-40, 40, size=(k, 3)) * [1, 1, 0.05]
    pts = [c + rng.normal(size=(rng.integers(40, 400), 3)) * rng.uniform(0.1, 0.6) for c in cents]
    pts.append(np.stack([rng.uniform(-50, 50, 300), rng.uniform(-50, 50, 300), rng.uniform(0, 3, 300)], 1))
    X = np.concatenate(pts).astype(np.float32)
    return X[rng.permutation(len(X))]


def host_tree(lo, hi, w2, n, mcs=15, eps=0.15):
    from vilgod_amd._lib import lib
    lo = np.ascontiguousarray(lo, np.int32)
    hi = np.ascontiguousarray(hi, np.int32)
    w2 = np.ascontiguousarray(w2, np.float64)
    labels = np.zeros(n, np.int32)
    probs = np.zeros(n, np.float64)
    nc = ctypes.c_int32(0)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.vg_hdbscan_tree_host(p(lo), p(hi), p(w2), n, mcs, eps, p(labels), p(probs), ctypes.byref(nc))
    assert rc == 0
    return labels, probs, nc.value


# ------------------------------------------------------------------------------------------- CPU
@pytest.mark.parametrize('scene', ['lidar', 'blob'])
def test_oracle_pinned_to_sklearn(scene):
    from sklearn.cluster import HDBSCAN
    from sklearn.cluster._hdbscan import _tree
    from sklearn.cluster._hdbscan._linkage import HIERARCHY_dtype
    from sklearn.metrics import adjusted_rand_score
    X = lidar_scene(1, 4000) if scene == 'lidar' else blob_scene(0)
    n = len(X)
    m = HDBSCAN(min_cluster_size=15, min_samples=16, cluster_selection_epsilon=0.15, algorithm='kd_tree').fit(X.astype(np.float64))
    core2 = ho.core_distances_sq(X)
    edges, w2 = ho.mst_prim(X, core2)
    e, w2s = ho.sort_edges(edges, w2)
    # (1) MST weights: bit identical
    assert np.array_equal(np.sort(m._single_linkage_tree_['value']), np.sqrt(w2s))
    # (2) tree stages: identical to scikit-learn's own code on the same linkage
    l, r, v, s = ho.single_linkage(e, np.sqrt(w2s), n)
    mine = np.zeros(n - 1, dtype=HIERARCHY_dtype)
    mine['left_node'], mine['right_node'], mine['value'], mine['cluster_size'] = l, r, v, s
    lab_sk, prob_sk = _tree.tree_to_labels(mine, 15, 'eom', False, 0.15, None)
    lab, prob = ho.tree_from_mst(e, w2s, n)
    assert np.array_equal(ho.canonical(lab), ho.canonical(lab_sk))
    assert np.abs(prob - prob_sk).max() < 1e-12
    # (3) end to end
    assert adjusted_rand_score(lab, m.labels_) >= 0.99
    assert ((lab < 0) == (m.labels_ < 0)).mean() >= 0.99


def test_oracle_mst_is_a_spanning_tree_and_minimal_under_swaps():
    X = blob_scene(3)[:600]
    n = len(X)
    core2 = ho.core_distances_sq(X)
    edges, w2 = ho.mst_prim(X, core2)
    assert len(edges) == n - 1
    parent = list(range(n))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for a, b in edges:
        ra, rb = find(a), find(b)
        assert ra != rb
        parent[ra] = rb
    # brute-force MST weight (dense Prim without tie-breaking) agrees
    D = ((X[:, None, :].astype(np.float64) - X[None]) ** 2).sum(-1)
    W = np.maximum(np.maximum(D, core2[:, None]), core2[None, :])
    np.fill_diagonal(W, np.inf)
    from scipy.sparse.csgraph import minimum_spanning_tree
    ref = minimum_spanning_tree(np.where(np.isinf(W), 0, W)).data
    assert np.allclose(np.sort(ref), np.sort(w2), rtol=1e-12)


@pytest.mark.parametrize('case', ['lidar', 'blob', 'tiny', 'two_blobs'])
def test_host_tree_equals_oracle(case):
    if case == 'lidar':
        X = lidar_scene(2, 5000)
    elif case == 'blob':
        X = blob_scene(1)
    elif case == 'tiny':
        X = blob_scene(2)[:16]
    else:
        rng = np.random.default_rng(0)
        X = np.concatenate([rng.normal(size=(40, 3)) * 0.2, rng.normal(size=(40, 3)) * 0.2 + [5, 0, 0]]).astype(np.float32)
    n = len(X)
    core2 = ho.core_distances_sq(X)
    edges, w2 = ho.mst_prim(X, core2)
    e, w2s = ho.sort_edges(edges, w2)
    want_l, want_p = ho.tree_from_mst(e, w2s, n)
    got_l, got_p, nc = host_tree(e[:, 0], e[:, 1], w2s, n)
    assert np.array_equal(got_l, want_l)
    assert np.array_equal(got_p, want_p)
    assert nc == want_l.max() + 1


def test_host_tree_degenerate_inputs():
    l, p, nc = host_tree(np.zeros(0), np.zeros(0), np.zeros(0), 0)
    assert len(l) == 0 and nc == 0
    # n <= min_cluster_size: everything is noise (lidar_frame.py then creates no detection)
    X = np.random.default_rng(0).normal(size=(15, 3)).astype(np.float32)
    l, p, nc = host_tree(np.arange(14), np.arange(1, 15), np.ones(14), 15)
    assert (l == -1).all() and (p == 0).all() and nc == 0
    # duplicate points (zero distances -> lambda = inf -> probability 1)
    X = np.concatenate([np.zeros((20, 3)), np.ones((20, 3)) * 3]).astype(np.float32)
    core2 = ho.core_distances_sq(X)
    edges, w2 = ho.mst_prim(X, core2)
    e, w2s = ho.sort_edges(edges, w2)
    want_l, want_p = ho.tree_from_mst(e, w2s, len(X))
    got_l, got_p, _ = host_tree(e[:, 0], e[:, 1], w2s, len(X))
    assert np.array_equal(got_l, want_l) and np.array_equal(got_p, want_p)


def test_detections_from_labels_matches_reference_rule():
    """lidar_frame.py:163-167, 232-237 (run against the reference in make_golden.py `detect`)."""
    labels = np.array([0, 0, 1, -1, 1, 2, 2, 2])
    probs = np.array([1, .2, .9, 0, .31, .29, .1, .2])
    det = ho.detections_from_labels(labels, probs)
    assert [d.tolist() for d in det] == [[0], [2, 4]]
