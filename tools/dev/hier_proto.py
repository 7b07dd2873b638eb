"""Prototype (CPU, plain loops) of the data-parallel formulation of the HDBSCAN hierarchy stage that csrc/hdbscan_device.hip runs:
every step is a per-edge / per-point / per-cluster rule with bounded local work instead of the host's sequential union-find over all
n - 1 edges.  Checked here against vg_hdbscan_tree_host on random trees (ties included) before any kernel was written.

    python tools/dev/hier_proto.py [n] [trials]

Rules (rank = position of an edge in the (w2, lo, hi) order; "side of u at e" = the vertices u reaches over edges of lower rank):
  1. a dendrogram node (edge e) is a TRUE SPLIT iff both sides hold >= mcs vertices: two bounded walks (<= mcs vertices each).
  2. a point leaves its cluster at a(p) = the largest-rank edge among the first mcs - 1 edges Prim's algorithm takes from p on the tree
     (lambda_p = 1 / dist(a(p))): a(p) is never a split, its node always holds >= mcs points.
  3. the tree minus its split edges falls into SEGMENTS; Kruskal over the split edges alone (a few hundred) on the segments gives the
     cluster tree: a non-split edge has a small side, which holds no split, so contracting all of them first changes no ancestry.
  4. the cluster of a non-split node j: from the segment's lowest-rank split upward, the first split of larger rank than j -- its child
     on this side (none: the root cluster).
  5. the library numbers clusters in BFS order of the DENDROGRAM: depth of a split = depth of its cluster's top + the cluster's chain
     length (its non-split nodes that hold >= mcs points), order inside a depth = preorder of the split tree.
  6. stability sums in the library's row order: per cluster its chain nodes by descending rank, k_j equal terms each, then the two
     child rows.
"""
import ctypes
import heapq
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def host_tree(lo, hi, w2, n, mcs, eps):
    from vilgod_amd._lib import lib, check
    labels = np.empty(n, np.int32)
    probs = np.empty(n, np.float64)
    nc = ctypes.c_int32(0)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    check(lib.vg_hdbscan_tree_host(p(lo), p(hi), p(w2), n, mcs, eps, p(labels), p(probs), ctypes.byref(nc)), 'tree')
    return labels, probs, nc.value


def proto_tree(lo, hi, w2, n, mcs, eps):
    m = n - 1
    labels = np.full(n, -1, np.int32)
    probs = np.zeros(n)
    if n <= mcs:
        return labels, probs, 0
    order = np.lexsort((hi, lo, w2))
    lo, hi, w2 = lo[order], hi[order], w2[order]
    adj = [[] for _ in range(n)]                 # (rank, other) ascending by rank
    for r in range(m):
        adj[lo[r]].append((r, hi[r]))
        adj[hi[r]].append((r, lo[r]))

    def side(u, frm, r):                         # vertices u reaches over edges of rank < r without going back to frm, capped at mcs
        cnt = 0
        st = [(u, frm)]
        while st:
            x, f = st.pop()
            cnt += 1
            if cnt >= mcs:
                return mcs
            for (rr, y) in adj[x]:
                if rr >= r:
                    break
                if y != f:
                    st.append((y, x))
        return cnt
    su = np.array([side(lo[r], hi[r], r) for r in range(m)])
    sv = np.array([side(hi[r], lo[r], r) for r in range(m)])
    split = (su >= mcs) & (sv >= mcs)
    # ---- rule 2 ----
    a = np.empty(n, np.int64)
    for p in range(n):
        heap = [(rr, y, p) for (rr, y) in adj[p][:mcs - 1]]
        heapq.heapify(heap)
        best = -1
        for _ in range(mcs - 1):
            rr, y, f = heapq.heappop(heap)
            best = max(best, rr)
            k = 0
            for (r2, z) in adj[y]:
                if z == f:
                    continue
                heapq.heappush(heap, (r2, z, y))
                k += 1
                if k >= mcs - 1:
                    break
        a[p] = best
    assert not split[a].any()
    # ---- rule 3: segments, then Kruskal over the splits ----
    par = np.arange(n)

    def find(x):
        while par[x] != x:
            par[x] = par[par[x]]
            x = par[x]
        return x
    for r in range(m):
        if not split[r]:
            par[find(lo[r])] = find(hi[r])
    seg = np.array([find(x) for x in range(n)])
    S = np.nonzero(split)[0]                     # split edges, ascending rank
    ns = len(S)
    sp_parent = np.full(ns, -1)
    sp_side = np.zeros(ns, np.int64)
    kid = np.full((ns, 2), -1)                   # split below each side (or -1)
    first_split = {}                             # segment -> (split index, side)
    top = {}
    par2 = {}

    def find2(x):
        while par2.setdefault(x, x) != x:
            par2[x] = par2[par2[x]]
            x = par2[x]
        return x
    for k, r in enumerate(S):
        for sd, v in ((0, lo[r]), (1, hi[r])):
            g = seg[v]
            if g not in first_split:
                first_split[g] = (k, sd)
            root = find2(g)
            t = top.get(root, -1)
            kid[k, sd] = t
            if t >= 0:
                sp_parent[t] = k
                sp_side[t] = sd
        ra, rb = find2(seg[lo[r]]), find2(seg[hi[r]])
        par2[ra] = rb
        top[rb] = k
    # provisional cluster ids: 0 = root, 1 + 2 k + side
    # ---- rule 4: cluster of every non-split node that holds >= mcs points ----
    bignode = (~split) & (su + sv >= mcs)
    cl_of_edge = np.full(m, -1)
    for r in np.nonzero(bignode)[0]:
        g = seg[lo[r]]
        c = 0
        if g in first_split:
            k, sd = first_split[g]
            while k >= 0 and S[k] < r:
                sd = sp_side[k]
                k = sp_parent[k]
            c = 0 if k < 0 else 1 + 2 * k + sd
        cl_of_edge[r] = c
    ncp = 1 + 2 * ns
    cnt_edge = np.bincount(a, minlength=m)       # k_j
    assert (cnt_edge[bignode] > 0).all() and (cnt_edge[~bignode] == 0).all()
    chainlen = np.bincount(cl_of_edge[bignode], minlength=ncp)
    npts = np.bincount(cl_of_edge[a], minlength=ncp)
    # ---- rule 5: BFS numbering ----
    term = np.full(ncp, -1)                      # terminating split of each provisional cluster
    if ns:
        term[0] = ns - 1
        for k in range(ns):
            for sd in (0, 1):
                term[1 + 2 * k + sd] = kid[k, sd]
    depth = np.zeros(ns, np.int64)
    pre = np.zeros(ns, np.int64)
    if ns:
        st = [(ns - 1, 0)]                       # (split, depth of its cluster's top)
        counter = 0
        while st:
            k, dtop = st.pop()
            c = 0 if sp_parent[k] < 0 else 1 + 2 * sp_parent[k] + sp_side[k]
            depth[k] = dtop + chainlen[c]
            pre[k] = counter
            counter += 1
            for sd in (1, 0):
                if kid[k, sd] >= 0:
                    st.append((kid[k, sd], depth[k] + 1 + 0))
        # NOTE the child cluster's top sits one level below the split
    bfs = sorted(range(ns), key=lambda k: (depth[k], pre[k]))
    final = np.zeros(ncp, np.int64)              # provisional -> library cluster id
    for q, k in enumerate(bfs):
        final[1 + 2 * k] = 2 * q + 1
        final[1 + 2 * k + 1] = 2 * q + 2
    nc = ncp
    inv = np.zeros(nc, np.int64)
    inv[final] = np.arange(ncp)
    lam_edge = np.where(w2 > 0, 1.0 / np.sqrt(w2), np.inf)
    # sizes bottom-up
    size = npts.copy()
    for k in range(ns):                          # ascending rank = children before parents
        c = 0 if sp_parent[k] < 0 else 1 + 2 * sp_parent[k] + sp_side[k]
        size[c] += size[1 + 2 * k] + size[1 + 2 * k + 1]
    assert size[0] == n
    birth = np.zeros(nc)
    cpar = np.full(nc, -1)
    for k in range(ns):
        c = 0 if sp_parent[k] < 0 else 1 + 2 * sp_parent[k] + sp_side[k]
        for sd in (0, 1):
            birth[final[1 + 2 * k + sd]] = lam_edge[S[k]]
            cpar[final[1 + 2 * k + sd]] = final[c]
    # ---- rule 6: stability ----
    stab = np.zeros(nc)
    death = np.zeros(nc)
    chains = [[] for _ in range(ncp)]
    for r in np.nonzero(bignode)[0][::-1]:
        chains[cl_of_edge[r]].append(r)
    for c in range(ncp):
        f = final[c]
        s = 0.0
        for r in chains[c]:
            t = lam_edge[r] - birth[f]
            for _ in range(cnt_edge[r]):
                s += t
            death[f] = max(death[f], lam_edge[r])
        k = term[c]
        if k >= 0:
            t = lam_edge[S[k]] - birth[f]
            s += t * float(size[1 + 2 * k])
            s += t * float(size[1 + 2 * k + 1])
            death[f] = max(death[f], lam_edge[S[k]])
        stab[f] = s
    kids = [[] for _ in range(nc)]
    for c in range(1, nc):
        kids[cpar[c]].append(c)
    selected = np.ones(nc, bool)
    selected[0] = False
    for c in range(nc - 1, 0, -1):
        sub = 0.0
        for k in kids[c]:
            sub += stab[k]
        if sub > stab[c]:
            selected[c] = False
            stab[c] = sub
        else:
            st = list(kids[c])
            while st:
                k = st.pop()
                selected[k] = False
                st.extend(kids[k])
    if eps != 0.0 and nc > 1:
        cand = np.zeros(nc, bool)
        for c in range(1, nc):
            if not selected[c]:
                continue
            if 1.0 / birth[c] < eps:
                node = c
                while True:
                    p = cpar[node]
                    if p == 0:
                        break
                    if 1.0 / birth[p] > eps:
                        node = p
                        break
                    node = p
                cand[node] = True
            else:
                cand[c] = True
        for c in range(1, nc):
            selected[c] = False
            if not cand[c]:
                continue
            nested = False
            x = cpar[c]
            while x > 0:
                if cand[x]:
                    nested = True
                    break
                x = cpar[x]
            selected[c] = not nested
    label_of = np.full(nc, -1)
    nl = 0
    for c in range(1, nc):
        if selected[c]:
            label_of[c] = nl
            nl += 1
    owner = np.full(nc, -1)
    for c in range(1, nc):
        owner[c] = c if selected[c] else owner[cpar[c]]
    for p in range(n):
        f = final[cl_of_edge[a[p]]]
        o = owner[f]
        if o < 0:
            continue
        labels[p] = label_of[o]
        mx, lam = death[o], lam_edge[a[p]]
        probs[p] = 1.0 if (mx == 0.0 or np.isinf(lam)) else min(lam, mx) / mx
    return labels, probs, nl


def random_tree(rng, n, kind):
    if kind == 0:                                # random attachment
        lo = np.array([rng.integers(0, i) for i in range(1, n)], np.int32)
        hi = np.arange(1, n, dtype=np.int32)
    elif kind == 1:                              # blobs: MST of mutual reachability of a 2-D point set
        from scipy.sparse.csgraph import minimum_spanning_tree
        from scipy.spatial.distance import cdist
        c = rng.normal(size=(max(2, n // 60), 2)) * 8
        P = c[rng.integers(0, len(c), n)] + rng.normal(size=(n, 2))
        D = cdist(P, P)
        core = np.sort(D, axis=1)[:, min(5, n - 1)]
        R = np.maximum(D, np.maximum(core[:, None], core[None, :]))
        np.fill_diagonal(R, 0)
        T = minimum_spanning_tree(R).tocoo()
        lo, hi = T.row.astype(np.int32), T.col.astype(np.int32)
        w = T.data
        perm = rng.permutation(n).astype(np.int32)
        lo, hi = perm[lo], perm[hi]
        l2, h2 = np.minimum(lo, hi), np.maximum(lo, hi)
        return l2, h2, (w * w)
    else:                                        # a path
        perm = rng.permutation(n).astype(np.int32)
        lo, hi = perm[:-1], perm[1:]
    perm = rng.permutation(n).astype(np.int32)
    lo, hi = perm[lo], perm[hi]
    l2, h2 = np.minimum(lo, hi), np.maximum(lo, hi)
    w = rng.random(n - 1)
    if rng.random() < 0.5:
        w = np.round(w * 20) / 20               # many ties, zeros included
    return l2.astype(np.int32), h2.astype(np.int32), w * w


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    rng = np.random.default_rng(0)
    bad = 0
    for t in range(trials):
        kind = t % 3
        nn = int(rng.integers(max(8, n // 4), n + 1))
        mcs = int(rng.choice([2, 3, 5, 15]))
        eps = float(rng.choice([0.0, 0.15, 0.5]))
        lo, hi, w2 = random_tree(rng, nn, kind)
        order = np.lexsort((hi, lo, w2))
        lo, hi, w2 = np.ascontiguousarray(lo[order]), np.ascontiguousarray(hi[order]), np.ascontiguousarray(w2[order])
        L0, P0, c0 = host_tree(lo, hi, w2, nn, mcs, eps)
        L1, P1, c1 = proto_tree(lo, hi, w2, nn, mcs, eps)
        ok = c0 == c1 and np.array_equal(L0, L1) and np.array_equal(P0, P1)
        print(f'trial {t}: kind {kind} n {nn} mcs {mcs} eps {eps}: clusters {c0} / {c1}  {"ok" if ok else "DIFFERENT"}', flush=True)
        bad += not ok
    print('all equal' if bad == 0 else f'{bad} trials differ')
    return bad


if __name__ == '__main__':
    sys.exit(1 if main() else 0)
