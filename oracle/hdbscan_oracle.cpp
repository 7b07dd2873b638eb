// CPU oracle, clustering row B2: the O(n^2) Prim of oracle/hdbscan_oracle.py::mst_prim in C, so that the oracle can be
// run at the benchmark's full size (~80k non-ground points of a 150k-point frame) in seconds instead of minutes.
//
// TEST INFRASTRUCTURE ONLY (tests/, tests/golden/make_*.py, smoke(), bench.py cpu_baseline); never linked into the product.
//
// Same numeric model and the same strict total edge order as the Python restatement (which stays the readable statement of
// the algorithm and is checked against this file on small inputs, tests/test_cluster.py):
//   d2(a,b)  = ((dx*dx + dy*dy) + dz*dz) (+ ...)   float64, left to right, no FMA contraction (-ffp-contract=off)
//   w2(a,b)  = max(max(d2, core2[b]), core2[a])    squared mutual-reachability weight
//   order    = (w2, d2, min(a,b), max(a,b))        unique MST (hdbscan_oracle.py header)
// The arithmetic it stands in for lives in the un-vendored `hdbscan` package (scikit-learn-contrib; call sites
// /root/reference/src/utils/cluster_utils.py:11-12, src/vilgod/zero_shot_detector.py:236,248).
#include <cstdint>
#include <cmath>
#include <limits>
#include <vector>

extern "C" int vgo_mst_prim(const double* X, int64_t n, int dim, const double* core2, int64_t* edges /*[n-1][2]*/,
                            double* w2 /*[n-1]*/) {
    if (n < 2 || dim < 1 || dim > 8) return 1;
    const double INF = std::numeric_limits<double>::infinity();
    std::vector<double> best((size_t)n, INF), bd2((size_t)n, INF);
    std::vector<int64_t> blo((size_t)n, n), bhi((size_t)n, n), src((size_t)n, 0);
    std::vector<char> in_tree((size_t)n, 0);
    int64_t cur = 0;
    in_tree[0] = 1;
    for (int64_t e = 0; e < n - 1; ++e) {
        const double* xc = X + (size_t)cur * dim;
        const double cc = core2[cur];
        int64_t nxt = -1;
        double nb = INF, nd = INF;
        int64_t nlo = n, nhi = n;
        for (int64_t i = 0; i < n; ++i) {
            if (in_tree[i]) continue;
            const double* xi = X + (size_t)i * dim;
            double d0 = xi[0] - xc[0];
            double d2 = d0 * d0;
            for (int c = 1; c < dim; ++c) {
                const double d = xi[c] - xc[c];
                d2 = d2 + d * d;
            }
            double w = d2 > core2[i] ? d2 : core2[i];
            w = w > cc ? w : cc;
            const int64_t lo = i < cur ? i : cur, hi = i < cur ? cur : i;
            if (w < best[i] || (w == best[i] && (d2 < bd2[i] || (d2 == bd2[i] && (lo < blo[i] || (lo == blo[i] && hi < bhi[i])))))) {
                best[i] = w; bd2[i] = d2; blo[i] = lo; bhi[i] = hi; src[i] = cur;
            }
            if (best[i] < nb || (best[i] == nb && (bd2[i] < nd || (bd2[i] == nd && (blo[i] < nlo || (blo[i] == nlo && bhi[i] < nhi)))))) {
                nb = best[i]; nd = bd2[i]; nlo = blo[i]; nhi = bhi[i]; nxt = i;
            }
        }
        if (nxt < 0) return 2;
        edges[2 * e] = src[nxt];
        edges[2 * e + 1] = nxt;
        w2[e] = nb;
        in_tree[nxt] = 1;
        cur = nxt;
    }
    return 0;
}
