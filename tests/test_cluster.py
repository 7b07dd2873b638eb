"""Spatial clustering parity.

The arithmetic of the reference lives in the un-vendored, unpinned `hdbscan` package, which is absent here
-> the oracle (oracle/hdbscan_oracle.py) restates the published algorithm and is pinned against the
scikit-learn implementation of the same algorithm:
  * mutual-reachability MST weights: bit-identical multiset to scikit-learn's (`_single_linkage_tree_`)
  * tree stages: identical labels/probabilities to scikit-learn's own `tree_to_labels` on the same linkage
  * end to end: adjusted Rand index >= 0.99 and >= 99 % agreement on the noise set (equal-weight MST edges are
    ordered arbitrarily by both libraries; the oracle and the GPU use the strict order (w2, pair d2, lo, hi) for the MST;
    the linkage stage orders equal-weight MST edges by (w2, lo, hi))
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


# ------------------------------------------------------------------------------------------- GPU
def _gpu_model(cuda, n, hierarchy='host'):
    from vilgod_amd.hdbscan import HDBSCAN
    return HDBSCAN(cluster_selection_epsilon=0.15, min_cluster_size=15, metric='euclidean', core_dist_n_jobs=-1,
                   max_points=n + 16, device=cuda, hierarchy=hierarchy)


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['lidar4k', 'lidar10k', 'blob', 'dupes', 'line', 'shifted'])
def test_hip_core_mst_labels_equal_oracle(cuda, case):
    """Bit-exact: squared core distances, the MST edge set + weights, labels, probabilities."""
    import torch
    rng = np.random.default_rng(1)
    if case == 'lidar4k':
        X = lidar_scene(1, 4000)
    elif case == 'lidar10k':
        X = lidar_scene(5, 10000)
    elif case == 'blob':
        X = blob_scene(0)
    elif case == 'dupes':        # exact duplicate points and many equal distances (integer lattice)
        X = np.concatenate([rng.integers(0, 6, size=(400, 3)).astype(np.float32) * 0.25,
                            rng.integers(0, 6, size=(300, 3)).astype(np.float32) * 0.25 + [20, 0, 0]]).astype(np.float32)
    elif case == 'line':         # extremely sparse: forces the coarse levels / brute-force fallback
        X = np.stack([np.arange(200) * 7.0, np.zeros(200), np.zeros(200)], 1).astype(np.float32)
    else:                        # far from the origin (ref frame after 100 m of driving) + outliers beyond the grid
        X = lidar_scene(2, 4000) + np.array([140.0, -90.0, 3.0], dtype=np.float32)
        X = np.concatenate([X, np.array([[500, 500, 40], [-400, 0, -30]], dtype=np.float32)]).astype(np.float32)
    n = len(X)
    model = _gpu_model(cuda, n)
    lo, hi, w2, core2 = model.mst(torch.from_numpy(X).to(cuda), want_core=True)
    lo, hi, w2, core2 = lo.cpu().numpy(), hi.cpu().numpy(), w2.cpu().numpy(), core2.cpu().numpy()
    want_core2 = ho.core_distances_sq(X)
    assert np.array_equal(core2, want_core2)
    edges, ew2 = ho.mst_prim(X, want_core2)
    e, w2s = ho.sort_edges(edges, ew2)
    assert np.array_equal(w2, w2s)                                   # weights, sorted
    got = np.stack([lo, hi], 1)[np.lexsort((hi, lo, w2))]
    assert np.array_equal(got, e)                                    # the same (unique) tree
    want_l, want_p = ho.tree_from_mst(e, w2s, n)
    got_model = model.fit(X)
    assert np.array_equal(got_model.labels_, want_l)
    assert np.array_equal(got_model.probabilities_, want_p)
    on_device = _gpu_model(cuda, n, hierarchy='device').fit(X)        # the hierarchy stage as kernels (csrc/hdbscan_device.hip) against the oracle
    assert np.array_equal(on_device.labels_, want_l)
    assert np.array_equal(on_device.probabilities_, want_p)
    print(f'{case}: n={n} clusters={want_l.max() + 1} rounds={model.n_rounds_}')


@pytest.mark.gpu
def test_hip_cluster_small_and_degenerate(cuda):
    import torch
    for hierarchy in ('host', 'device'):
        model = _gpu_model(cuda, 1000, hierarchy)
        for n in [0, 1, 2, 15, 16, 17, 40]:
            X = np.random.default_rng(n).normal(size=(n, 3)).astype(np.float32)
            m = model.fit(X)
            want_l, want_p = ho.fit(X)
            assert np.array_equal(m.labels_, want_l), (hierarchy, n)
            assert np.array_equal(m.probabilities_, want_p), (hierarchy, n)


@pytest.mark.gpu
def test_hip_cluster_full_size_properties(cuda):
    """BASELINE size (150k-point frame -> ~80k non-ground points): properties that need no CPU run --
    spanning tree, weights sorted, invariance of weights/labels under a permutation of the input points
    up to tie-breaking (we compare the weight multiset exactly and the partition by ARI), cluster sizes."""
    import torch
    from sklearn.metrics import adjusted_rand_score
    X = lidar_scene(0, 150_000)
    n = len(X)
    model = _gpu_model(cuda, n)
    lo, hi, w2 = [t.cpu().numpy() for t in model.mst(torch.from_numpy(X).to(cuda))]
    assert len(lo) == n - 1 and (lo < hi).all() and (np.diff(w2) >= 0).all()
    parent = np.arange(n)

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for a, b in zip(lo, hi):
        ra, rb = find(a), find(b)
        assert ra != rb
        parent[ra] = rb
    labels = model.fit(X).labels_
    sizes = np.bincount(labels[labels >= 0])
    assert sizes.min() >= 15
    perm = np.random.default_rng(0).permutation(n)
    lo2, hi2, w22 = [t.cpu().numpy() for t in model.mst(torch.from_numpy(X[perm]).to(cuda))]
    assert np.array_equal(w2, w22)
    labels2 = model.fit(X[perm]).labels_
    ari = adjusted_rand_score(labels[perm], labels2)
    print(f'full size: n={n} clusters={labels.max() + 1} rounds={model.n_rounds_} ARI under permutation={ari:.5f}')
    assert ari > 0.99


def test_prim_in_c_equals_python_prim():
    """oracle/hdbscan_oracle.cpp (the form that finishes at full size) == the readable Python Prim, ties and 5-D included."""
    rng = np.random.default_rng(3)
    for n, dim in [(300, 3), (1500, 3), (900, 5)]:
        X = rng.normal(size=(n, dim)).astype(np.float32)
        X[:40] = X[40:80]                                   # exact duplicates -> equal weights and equal pair distances
        X[100:160, :] = np.round(X[100:160, :] * 4) / 4       # lattice points -> many equal distances
        core2 = ho.core_distances_sq(X)
        e1, w1 = ho.mst_prim(X, core2)
        e2, w2 = ho.mst_prim_c(X, core2)
        assert np.array_equal(e1, e2) and np.array_equal(w1, w2)


def test_full_size_fixture_is_committed(golden_dir):
    import json
    g = json.load(open(f'{golden_dir}/cluster_full_golden.json'))
    assert g['n'] > 70_000 and g['n_clusters'] > 50 and len(g['mst_w2_sha256']) == 64


@pytest.mark.gpu
def test_hip_cluster_full_size_equals_oracle_fixture(cuda, golden_dir):
    """BASELINE size: the HIP clustering of the ~79k non-ground points of one synthetic 150k-point frame against the CPU
    oracle's result frozen by tests/golden/make_cluster_fixture.py (core distances, MST edges + weights, labels,
    probabilities, detection lists: sha256, i.e. bit for bit)."""
    import json
    import sys
    import torch
    sys.path.insert(0, golden_dir)
    import make_cluster_fixture as mk
    g = json.load(open(f'{golden_dir}/cluster_full_golden.json'))
    X = mk.build_input()
    if mk.sha(X) != g['x_sha256']:
        # the input itself came out differently on this host (libm / compiler of the ground oracle): re-run the oracle here
        core2 = ho.core_distances_sq(X)
        edges, ew2 = ho.mst_prim_c(X, core2)
        e, w2s = ho.sort_edges(edges, ew2)
        wl, wp = ho.tree_from_mst(e, w2s, len(X))
        g = mk.digests(X, core2, e[:, 0], e[:, 1], w2s, wl, wp)
        print('fixture input differs on this host: oracle re-run live')
    n = len(X)
    model = _gpu_model(cuda, n)
    lo, hi, w2, core2 = model.mst(torch.from_numpy(X).to(cuda), want_core=True)
    lo, hi, w2, core2 = lo.cpu().numpy(), hi.cpu().numpy(), w2.cpu().numpy(), core2.cpu().numpy()
    order = np.lexsort((hi, lo, w2))
    for hierarchy in ('host', 'device'):                              # both forms of the hierarchy stage against the frozen oracle result
        m = _gpu_model(cuda, n, hierarchy).fit(X)
        got = mk.digests(X, core2, lo[order], hi[order], w2[order], m.labels_, m.probabilities_)
        for k in ('n', 'core2_sha256', 'mst_w2_sha256', 'mst_edges_sha256', 'labels_sha256', 'canonical_labels_sha256', 'probs_sha256',
                  'n_clusters', 'n_noise', 'n_detections', 'detection_sizes_sha256'):
            assert got[k] == g[k], (hierarchy, k, got[k], g[k])
    print(f"full-size fixture: n={n} clusters={got['n_clusters']} noise={got['n_noise']} rounds={model.n_rounds_}: identical")
