"""CPU oracle for spatial clustering: HDBSCAN as the reference configures it
(tools/configs/preprocessor/waymo.yaml:10-15: min_cluster_size=15, cluster_selection_epsilon=0.15,
euclidean, EOM, allow_single_cluster=False; call sites src/utils/cluster_utils.py:11-12,
src/vilgod/zero_shot_detector.py:248) followed by the probability cut of lidar_frame.py:163-167.

TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py cpu_baseline).

The arithmetic lives in a THIRD-PARTY dependency that is not vendored under /root/reference and not
pinned: `hdbscan` (scikit-learn-contrib; README.md:74-75 `pip install hdbscan`).  It is not installed
here and cannot be fetched -> this file restates the PUBLISHED algorithm (Campello et al. 2013; McInnes &
Healy 2017) with the tree semantics of the scikit-learn port whose sources are readable in this image
(sklearn/cluster/_hdbscan/{_linkage,_tree,_reachability}.pyx; line numbers below refer to those files):

  core distance   distance to the 15th nearest OTHER point (hdbscan counts without self; sklearn's
                  min_samples counts self, so the sklearn stand-in uses min_samples=16)
  d_mr(a,b)       max(core[a], core[b], |a-b|)                      _reachability.pyx / _linkage.pyx:171-181
  MST             exact minimum spanning tree of the complete d_mr graph (the reference library's
                  default is an approximate Boruvka; any exact MST has the same weight multiset)
  single linkage  sort edges, union-find with sizes                 _linkage.pyx:226-274
  condense        _tree.pyx:122-238;  stability :240-278;  EOM + epsilon :644-761 (epsilon_search :606-641,
                  traverse_upwards :578-604);  labels :433-512;  probabilities :515-554

PARITY PIN: `sklearn_fit` (the scikit-learn implementation itself) is the stand-in for the absent library;
tests/test_cluster.py checks `fit` against it (same partition after canonical relabel, same noise set,
probabilities within 1e-9) on seeded scenes.  Equal-weight MST edges (ties) are resolved here by
(weight, pair distance, min index, max index); the libraries leave tie order unspecified.
"""
import numpy as np
from scipy.spatial import cKDTree

MIN_CLUSTER_SIZE = 15
EPSILON = 0.15
PROB_THRESHOLD = 0.3      # waymo.yaml:63 `propability_threshold`
PRIM_C_ABOVE = 6000       # larger inputs take the C form of the same Prim (mst_prim_c)


# ------------------------------------------------------------------------------------------------
def _d2(d):
    """float64 squared norm over the last axis, summed left to right: (((dx2 + dy2) + dz2) + de2) + dt2 --
    3-D for single-frame clustering, 5-D for the two-frame input (zero_shot_detector.py:232-239)."""
    acc = d[..., 0] * d[..., 0]
    for c in range(1, d.shape[-1]):
        acc = acc + d[..., c] * d[..., c]
    return acc


def core_distances_sq(X, k=MIN_CLUSTER_SIZE):
    """Squared distance to the k-th nearest other point, float64, d2 = (dx*dx + dy*dy) + dz*dz (+ ...)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    n = len(X)
    kk = min(n, k + 9)
    _, idx = cKDTree(X).query(X, k=kk)
    idx = idx.reshape(n, kk)
    d = X[:, None, :] - X[idx]
    d2 = _d2(d)
    d2.sort(axis=1)
    if n <= k:
        return np.full(n, np.inf)
    return d2[:, k]        # column 0 is the point itself (0.0)


def mst_prim(X, core2):
    """THE minimum spanning tree of the mutual-reachability graph under the strict total edge order
    (w2, d2, min(a,b), max(a,b)) -- mutual-reachability weight, then the plain squared distance of the pair, then the
    ids -- unique, so any exact algorithm using the same order (the GPU's Boruvka) returns the same edge set.
    (Equal weights are the rule, not the exception: w = core[a] for every neighbour inside a's core ball.  The
    libraries leave their order unspecified; preferring the SHORTER pair lets a nearest-neighbour search stop at the
    nearest qualifying point instead of scanning the whole ball for the smallest id.)  O(n^2) Prim; squared weights."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    n = len(X)
    ids = np.arange(n)
    in_tree = np.zeros(n, bool)
    best = np.full(n, np.inf)
    bd2 = np.full(n, np.inf)
    blo = np.full(n, n, np.int64)
    bhi = np.full(n, n, np.int64)
    src = np.zeros(n, np.int64)
    edges = np.zeros((n - 1, 2), np.int64)
    w2 = np.zeros(n - 1)
    cur = 0
    in_tree[0] = True
    for e in range(n - 1):
        d = X - X[cur]
        d2 = _d2(d)
        w = np.maximum(np.maximum(d2, core2), core2[cur])
        lo, hi = np.minimum(ids, cur), np.maximum(ids, cur)
        tie_w = w == best
        tie_d = tie_w & (d2 == bd2)
        upd = ~in_tree & ((w < best) | (tie_w & (d2 < bd2)) | (tie_d & ((lo < blo) | ((lo == blo) & (hi < bhi)))))
        best[upd], bd2[upd], blo[upd], bhi[upd], src[upd] = w[upd], d2[upd], lo[upd], hi[upd], cur
        cand = np.where(in_tree, np.inf, best)
        tie = np.flatnonzero(cand == cand.min())
        nxt = int(tie[np.lexsort((bhi[tie], blo[tie], bd2[tie]))[0]]) if len(tie) > 1 else int(tie[0])
        edges[e] = (src[nxt], nxt)
        w2[e] = cand[nxt]
        in_tree[nxt] = True
        cur = nxt
    return edges, w2


def mst_prim_c(X, core2):
    """`mst_prim` in C (oracle/hdbscan_oracle.cpp: same numeric model, same total order; checked against `mst_prim` in
    tests/test_cluster.py) -- the form that finishes at the benchmark's full size (80k points: seconds instead of minutes)."""
    import ctypes
    from . import patchworkpp as _pw
    lib = _pw.load()
    X = np.ascontiguousarray(X, dtype=np.float64)
    core2 = np.ascontiguousarray(core2, dtype=np.float64)
    n, dim = X.shape
    edges = np.zeros((n - 1, 2), np.int64)
    w2 = np.zeros(n - 1)
    lib.vgo_mst_prim.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    rc = lib.vgo_mst_prim(X.ctypes.data, n, dim, core2.ctypes.data, edges.ctypes.data, w2.ctypes.data)
    if rc != 0:
        raise RuntimeError(f'vgo_mst_prim failed ({rc})')
    return edges, w2


def sort_edges(edges, w2):
    lo = np.minimum(edges[:, 0], edges[:, 1])
    hi = np.maximum(edges[:, 0], edges[:, 1])
    order = np.lexsort((hi, lo, w2))
    return np.stack([lo, hi], 1)[order], w2[order]


# ------------------------------------------------------------------------------------------------
def single_linkage(edges_sorted, dist_sorted, n):
    """_linkage.pyx:226-274: returns (left, right, value, size) arrays of length n-1; node ids >= n."""
    parent = np.arange(2 * n - 1)
    size = np.ones(2 * n - 1, np.int64)

    def find(x):
        r = x
        while parent[r] != r:
            r = parent[r]
        while parent[x] != r:
            parent[x], x = r, parent[x]
        return r

    left = np.zeros(n - 1, np.int64)
    right = np.zeros(n - 1, np.int64)
    nxt = n
    for i in range(n - 1):
        a, b = find(edges_sorted[i, 0]), find(edges_sorted[i, 1])
        left[i], right[i] = a, b
        parent[a] = parent[b] = nxt
        size[nxt] = size[a] + size[b]
        nxt += 1
    return left, right, np.asarray(dist_sorted, dtype=np.float64), size[n:]


def condense(left, right, value, sizes, n, min_cluster_size=MIN_CLUSTER_SIZE):
    """_tree.pyx:122-238 -> rows (parent, child, lambda, child_size) in the same BFS order."""
    root = 2 * (n - 1)
    relabel = {root: n}
    next_label = n + 1
    rows = []
    ignore = set()

    def leaves(node):
        out, stack = [], [node]
        while stack:
            x = stack.pop()
            if x < n:
                out.append(x)
            else:
                ignore.add(x)
                stack.extend((right[x - n], left[x - n]))
        return out

    queue = [root]
    while queue:
        nxt_q = []
        for node in queue:
            if node < n or node in ignore:
                continue
            l, r, d = left[node - n], right[node - n], value[node - n]
            lam = 1.0 / d if d > 0.0 else np.inf
            lc = sizes[l - n] if l >= n else 1
            rc = sizes[r - n] if r >= n else 1
            p = relabel[node]
            if lc >= min_cluster_size and rc >= min_cluster_size:
                relabel[l] = next_label
                rows.append((p, next_label, lam, lc))
                next_label += 1
                relabel[r] = next_label
                rows.append((p, next_label, lam, rc))
                next_label += 1
            elif lc < min_cluster_size and rc < min_cluster_size:
                for s in leaves(l) + leaves(r):
                    rows.append((p, s, lam, 1))
            elif lc < min_cluster_size:
                relabel[r] = p
                for s in leaves(l):
                    rows.append((p, s, lam, 1))
            else:
                relabel[l] = p
                for s in leaves(r):
                    rows.append((p, s, lam, 1))
            nxt_q.extend((l, r))
        queue = nxt_q
    return rows


def select_and_label(rows, n, eps=EPSILON):
    """stability (_tree.pyx:240-278), EOM (:729-741), epsilon (:743-761), labels (:433-512),
    probabilities (:515-554); allow_single_cluster=False."""
    parents = np.array([r[0] for r in rows], np.int64)
    children = np.array([r[1] for r in rows], np.int64)
    lambdas = np.array([r[2] for r in rows], np.float64)
    csizes = np.array([r[3] for r in rows], np.int64)
    root = n
    n_clusters = int(parents.max()) - root + 1
    birth = np.zeros(n_clusters)
    cpar = np.full(n_clusters, -1, np.int64)
    is_c = csizes > 1
    for p, c, lam in zip(parents[is_c], children[is_c], lambdas[is_c]):
        birth[c - root] = lam
        cpar[c - root] = p - root
    stab = np.zeros(n_clusters)
    np.add.at(stab, parents - root, (lambdas - birth[parents - root]) * csizes)
    kids = [[] for _ in range(n_clusters)]
    for c in range(1, n_clusters):
        kids[cpar[c]].append(c)
    selected = np.ones(n_clusters, bool)
    selected[0] = False
    for c in range(n_clusters - 1, 0, -1):            # descending id = leaves first, root excluded
        sub = sum(stab[k] for k in kids[c])
        if sub > stab[c]:
            selected[c] = False
            stab[c] = sub
        else:
            stack = list(kids[c])
            while stack:
                k = stack.pop()
                selected[k] = False
                stack.extend(kids[k])
    if eps != 0.0 and n_clusters > 1:
        sel = [c for c in range(1, n_clusters) if selected[c]]
        new_sel = set()
        for c in sel:
            if 1.0 / birth[c] < eps:
                node = c                                 # traverse_upwards, :578-604
                while True:
                    p = cpar[node]
                    if p == 0:
                        break                            # parent is the root: keep the child of the root
                    if 1.0 / birth[p] > eps:
                        node = p
                        break
                    node = p
                new_sel.add(node)
            else:
                new_sel.add(c)
        # epsilon_search marks every descendant of a chosen ancestor as processed (:634-636) so that no
        # selected cluster sits below another one; the library reaches that by iterating a Python set
        # (order unspecified).  Deterministic form of the same intent: the top-most candidate wins.
        selected[:] = False
        for c in new_sel:
            a, nested = cpar[c], False
            while a > 0:
                if a in new_sel:
                    nested = True
                    break
                a = cpar[a]
            if not nested:
                selected[c] = True
    # labelling: a point belongs to the nearest selected ancestor-or-self of the cluster it fell out of
    sel_ids = np.nonzero(selected)[0]
    label_of = {c: i for i, c in enumerate(sorted(sel_ids))}
    owner = np.full(n_clusters, -1, np.int64)
    for c in range(1, n_clusters):                       # parents have smaller ids than children
        owner[c] = c if selected[c] else owner[cpar[c]]
    labels = np.full(n, -1, np.int64)
    probs = np.zeros(n)
    death = np.zeros(n_clusters)
    np.maximum.at(death, parents - root, lambdas)        # max lambda over ALL rows of a parent (hdbscan lib)
    pt = ~is_c
    pp, pc, pl = parents[pt] - root, children[pt], lambdas[pt]
    own = owner[pp]
    ok = own >= 0
    labels[pc[ok]] = [label_of[o] for o in own[ok]]
    for point, o, lam in zip(pc[ok], own[ok], pl[ok]):
        mx = death[o]
        if mx == 0.0 or np.isinf(lam):
            probs[point] = 1.0
        else:
            probs[point] = min(lam, mx) / mx
    return labels, probs


def fit(X, min_cluster_size=MIN_CLUSTER_SIZE, eps=EPSILON):
    """-> (labels_ int64 [n], probabilities_ float64 [n]) like HDBSCAN(...).fit(X)."""
    X = np.asarray(X)
    n = len(X)
    if n <= min_cluster_size:
        return np.full(n, -1, np.int64), np.zeros(n)
    core2 = core_distances_sq(X, min_cluster_size)
    edges, w2 = (mst_prim_c if n > PRIM_C_ABOVE else mst_prim)(X, core2)
    e, w2s = sort_edges(edges, w2)
    l, r, v, s = single_linkage(e, np.sqrt(w2s), n)
    rows = condense(l, r, v, s, n, min_cluster_size)
    return select_and_label(rows, n, eps)


def tree_from_mst(edges_sorted, w2_sorted, n, min_cluster_size=MIN_CLUSTER_SIZE, eps=EPSILON):
    """Tree stages only, from an already sorted exact MST (used to check the product's tree code)."""
    l, r, v, s = single_linkage(edges_sorted, np.sqrt(w2_sorted), n)
    return select_and_label(condense(l, r, v, s, n, min_cluster_size), n, eps)


def sklearn_fit(X, min_cluster_size=MIN_CLUSTER_SIZE, eps=EPSILON):
    """The scikit-learn implementation with parameters mapped as SURVEY §8c prescribes."""
    from sklearn.cluster import HDBSCAN
    m = HDBSCAN(min_cluster_size=min_cluster_size, min_samples=min_cluster_size + 1, cluster_selection_epsilon=eps,
                algorithm='kd_tree', allow_single_cluster=False, copy=True).fit(np.asarray(X, dtype=np.float64))
    return m.labels_.astype(np.int64), m.probabilities_.astype(np.float64)


# ------------------------------------------------------------------------------------------------
def canonical(labels):
    """Relabel clusters by their smallest member index (partition comparison)."""
    labels = np.asarray(labels)
    out = np.full(len(labels), -1, np.int64)
    first = {}
    for i, l in enumerate(labels):
        if l >= 0 and l not in first:
            first[l] = len(first)
    for l, k in first.items():
        out[labels == l] = k
    return out


def detections_from_labels(labels, probs, threshold=PROB_THRESHOLD):
    """lidar_frame.py:163-167, 232-237: low-probability points become noise; clusters are enumerated by
    ascending label; each is the ascending list of its point indices."""
    labels = np.array(labels, copy=True)
    labels[np.asarray(probs) < threshold] = -1
    ids = np.sort(np.unique(labels[labels != -1]))
    return [np.where(labels == c)[0] for c in ids]
