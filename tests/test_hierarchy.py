"""HDBSCAN hierarchy stage on the device (csrc/hdbscan_device.hip, round 6) against the host stage (csrc/hdbscan_tree.cpp), bit for bit.

CPU: the kernels' per-element bodies (csrc/hdbscan_device.inc) compiled by g++ into an emulation that runs them phase by phase
(tests/emul/hdbscan_device_emul.cpp) -- checks the RULES of the data-parallel formulation against the sequential union-find on random
trees, ties, paths, stars, zero weights, every min_cluster_size the walk stacks allow.
GPU: the kernels themselves against the host stage on the same inputs and on real minimum spanning trees.
Replaces the tail of `cluster_model.fit` (src/vilgod/zero_shot_detector.py:248)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL_SRC = os.path.join(ROOT, 'tests', 'emul', 'hdbscan_device_emul.cpp')
EMUL_DIR = os.path.join(ROOT, 'tests', 'emul', '_build')


@pytest.fixture(scope='module')
def emul():
    os.makedirs(EMUL_DIR, exist_ok=True)
    so = os.path.join(EMUL_DIR, 'libhd_emul.so')
    inc = os.path.join(ROOT, 'vilgod_amd', 'csrc', 'hdbscan_device.inc')
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(EMUL_SRC), os.path.getmtime(inc)):
        subprocess.run(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', '-I', os.path.dirname(inc), EMUL_SRC, '-o', so], check=True)
    lib = ctypes.CDLL(so)
    lib.hd_emul_tree.restype = ctypes.c_int
    return lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def host_tree(lo, hi, w2, n, mcs, eps):
    from vilgod_amd._lib import lib, check
    labels = np.empty(n, np.int32)
    probs = np.empty(n, np.float64)
    nc = ctypes.c_int32(0)
    check(lib.vg_hdbscan_tree_host(_p(lo), _p(hi), _p(w2), n, mcs, ctypes.c_double(eps), _p(labels), _p(probs), ctypes.byref(nc)), 'vg_hdbscan_tree_host')
    return labels, probs, nc.value


def emul_tree(lib, lo, hi, w2, n, mcs, eps):
    labels = np.empty(n, np.int32)
    probs = np.empty(n, np.float64)
    nc, ns, sw = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
    rc = lib.hd_emul_tree(_p(lo), _p(hi), _p(w2), n, mcs, ctypes.c_double(eps), _p(labels), _p(probs), ctypes.byref(nc), ctypes.byref(ns), ctypes.byref(sw))
    assert rc == 0, rc
    return labels, probs, nc.value, ns.value, sw.value


def random_tree(rng, n, kind):
    """-> (lo, hi, w2) sorted by w2 ONLY (ties in arbitrary order, as the GPU's edge sort leaves them)."""
    if kind == 0:                                # random attachment
        lo = np.array([rng.integers(0, i) for i in range(1, n)], np.int64)
        hi = np.arange(1, n)
        w = rng.random(n - 1)
    elif kind == 1:                              # blobs: the mutual-reachability tree of a 2-D point set
        from scipy.sparse.csgraph import minimum_spanning_tree
        from scipy.spatial.distance import cdist
        c = rng.normal(size=(max(2, n // 60), 2)) * 8
        P = c[rng.integers(0, len(c), n)] + rng.normal(size=(n, 2))
        D = cdist(P, P)
        core = np.sort(D, axis=1)[:, min(5, n - 1)]
        R = np.maximum(D, np.maximum(core[:, None], core[None, :]))
        np.fill_diagonal(R, 0)
        T = minimum_spanning_tree(R).tocoo()
        lo, hi, w = T.row.astype(np.int64), T.col.astype(np.int64), T.data
    elif kind == 2:                              # a path
        lo, hi = np.arange(n - 1), np.arange(1, n)
        w = rng.random(n - 1)
    elif kind == 3:                              # a few hubs (stars joined in a chain): long adjacency lists
        hubs = max(1, n // 200)
        lo = np.array([i - 1 if i <= hubs else rng.integers(0, hubs) for i in range(1, n)], np.int64)
        lo[0] = 0
        hi = np.arange(1, n)
        w = rng.random(n - 1)
    else:                                        # a caterpillar of blobs with growing gaps: a deep split tree
        lo = np.array([rng.integers(max(0, i - 3), i) for i in range(1, n)], np.int64)
        hi = np.arange(1, n)
        w = rng.random(n - 1) * 0.1
        step = max(8, n // 40)
        w[step::step] = 1.0 + np.arange(len(w[step::step])) * 0.01
    perm = rng.permutation(n)
    lo, hi = perm[lo], perm[hi]
    l2, h2 = np.minimum(lo, hi).astype(np.int32), np.maximum(lo, hi).astype(np.int32)
    if kind != 1 and rng.random() < 0.5:
        w = np.round(w * 20) / 20               # many ties, zeros included
    w2 = (w * w).astype(np.float64)
    # shuffle inside runs of equal weight: the device entry must not depend on the order of ties
    sh = rng.permutation(len(w2))
    order = sh[np.argsort(w2[sh], kind='stable')]
    return np.ascontiguousarray(l2[order]), np.ascontiguousarray(h2[order]), np.ascontiguousarray(w2[order])


CASES = [(kind, n, mcs, eps, seed) for seed, (kind, n) in enumerate([(0, 300), (1, 400), (2, 500), (3, 700), (4, 900), (0, 2500), (1, 1200),
                                                                       (2, 64), (3, 3000), (4, 4000), (0, 17), (2, 33)])
         for mcs, eps in ((2, 0.0), (5, 0.15), (15, 0.15), (15, 0.0), (32, 0.5))]


@pytest.mark.parametrize('kind,n,mcs,eps,seed', CASES)
def test_rules_of_the_device_formulation_equal_the_host_stage(emul, kind, n, mcs, eps, seed):
    rng = np.random.default_rng(1000 + seed)
    lo, hi, w2 = random_tree(rng, n, kind)
    L0, P0, c0 = host_tree(lo, hi, w2, n, mcs, eps)
    L1, P1, c1, ns, sweeps = emul_tree(emul, lo, hi, w2, n, mcs, eps)
    assert c0 == c1
    assert np.array_equal(L0, L1)
    assert np.array_equal(P0.view(np.uint64), P1.view(np.uint64))           # bit for bit
    assert ns <= n // mcs                                                    # the bound the per-split arrays are sized by


def test_rules_on_small_and_degenerate_inputs(emul):
    for n, mcs in [(2, 2), (3, 2), (3, 5), (16, 15), (15, 15), (40, 15)]:
        rng = np.random.default_rng(n * 100 + mcs)
        lo, hi, w2 = random_tree(rng, n, 0)
        for zero in (False, True):
            w = np.zeros_like(w2) if zero else w2              # all points coincide: every weight 0, lambda = inf
            L0, P0, c0 = host_tree(lo, hi, w, n, mcs, 0.15)
            L1, P1, c1, _, _ = emul_tree(emul, lo, hi, w, n, mcs, 0.15)
            assert c0 == c1 and np.array_equal(L0, L1) and np.array_equal(P0.view(np.uint64), P1.view(np.uint64)), (n, mcs, zero)


# ---- GPU ---------------------------------------------------------------------------------------------------------------------------------
def device_tree(lo, hi, w2, n, mcs, eps, max_points=None):
    import torch
    from vilgod_amd.hdbscan import DeviceHierarchy
    dev = torch.device('cuda:0')
    h = DeviceHierarchy(max_points=max_points or max(n, 64), device=dev)
    labels, probs, nc = h.tree(torch.from_numpy(lo).to(dev), torch.from_numpy(hi).to(dev), torch.from_numpy(w2).to(dev), n, mcs, eps)
    return labels.cpu().numpy(), probs.cpu().numpy(), nc


@pytest.mark.gpu
@pytest.mark.parametrize('kind,n,mcs,eps,seed', CASES + [(1, 2000, 15, 0.15, 99), (4, 60000, 15, 0.15, 98), (3, 120000, 15, 0.15, 97), (0, 150000, 5, 0.15, 96)])
def test_device_hierarchy_equals_host_stage(cuda, kind, n, mcs, eps, seed):
    rng = np.random.default_rng(1000 + seed)
    lo, hi, w2 = random_tree(rng, n, kind)
    L0, P0, c0 = host_tree(lo, hi, w2, n, mcs, eps)
    L1, P1, c1 = device_tree(lo, hi, w2, n, mcs, eps)
    assert c0 == c1
    assert np.array_equal(L0, L1)
    assert np.array_equal(P0.view(np.uint64), P1.view(np.uint64))


@pytest.mark.gpu
def test_device_hierarchy_on_real_trees_and_handle_reuse(cuda):
    """The mutual-reachability trees of synthetic LiDAR frames (the pipeline's own input), one handle for frames of different sizes."""
    import torch
    from vilgod_amd import synthetic
    from vilgod_amd.hdbscan import HDBSCAN, DeviceHierarchy
    model = HDBSCAN(min_cluster_size=15, cluster_selection_epsilon=0.15, max_points=200_000, device=cuda)
    h = DeviceHierarchy(max_points=200_000, device=cuda)
    for seed, npts in [(1, 150_000), (2, 40_000), (3, 150_000)]:
        pts = synthetic.make_frame(seed, npts)
        X = torch.from_numpy(np.ascontiguousarray(pts[pts[:, 2] > 0.3][:, :3])).to(cuda)
        n = X.shape[0]
        lo, hi, w2 = model.mst(X)
        L0, P0, c0 = model.tree(lo.cpu().numpy(), hi.cpu().numpy(), w2.cpu().numpy(), n)
        L1, P1, c1 = h.tree(lo, hi, w2, n, 15, 0.15)
        assert c0 == c1 and c0 > 5
        assert np.array_equal(L0, L1.cpu().numpy())
        assert np.array_equal(P0.view(np.uint64), P1.cpu().numpy().view(np.uint64))


@pytest.mark.gpu
def test_device_hierarchy_refuses_what_it_cannot_hold(cuda):
    from vilgod_amd.hdbscan import DeviceHierarchy
    import torch
    h = DeviceHierarchy(max_points=1000, device=cuda)
    lo = torch.zeros(1999, dtype=torch.int32, device=cuda)
    with pytest.raises(RuntimeError):
        h.tree(lo, lo, torch.zeros(1999, dtype=torch.float64, device=cuda), 2000, 15, 0.15)      # more points than the handle holds
    with pytest.raises(RuntimeError):
        h.tree(lo[:99], lo[:99], torch.zeros(99, dtype=torch.float64, device=cuda), 100, 33, 0.15)   # min_cluster_size beyond the walk stacks
