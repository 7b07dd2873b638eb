// HDBSCAN hierarchy stage (kernel list K2d): from the sorted minimum spanning tree of
// the mutual-reachability graph to flat labels and membership probabilities.
//
// This is the sequential O(n alpha(n)) tail of the clustering path; it runs on the host (one thread per
// rank, overlapped with GPU work of the neighbouring frames) exactly as SURVEY §7 step 4 plans
// ("union-find dendrogram/condense/EOM+eps, C++ host").  The GPU produces and sorts the MST (cluster.hip).
//
// Semantics = the library the reference calls (hdbscan.HDBSCAN(min_cluster_size=15,
// cluster_selection_epsilon=0.15), tools/configs/preprocessor/waymo.yaml:10-15; src/vilgod/
// zero_shot_detector.py:248) as documented by its scikit-learn port:
//   single linkage  sklearn/cluster/_hdbscan/_linkage.pyx:226-274
//   condense        _tree.pyx:122-238      stability :240-278      EOM :729-741
//   epsilon         :578-641, :743-761     labels    :433-512      probabilities :515-554
// allow_single_cluster = False, cluster_selection_method = 'eom' (library defaults).
#include <math.h>
#include <stdint.h>
#include <algorithm>
#include <limits>
#include <vector>

#include "vilgod_hip.h"

namespace {

struct Tree {
    int n;
    std::vector<int> left, right, size;     // internal node i (id n+i)
    std::vector<double> dist;
};

// The stage's arrays, kept per calling thread between frames (a worker thread of the pipeline runs one frame after the other): a
// 79k-point frame needs ~5 MB of them, and fresh std::vectors of that size are mmap'ed and page-faulted anew on every call.
struct Work {
    std::vector<int32_t> lo, hi;
    std::vector<std::pair<int32_t, int32_t>> run;
    Tree t;
    std::vector<int> parent, top, sz, r_parent, r_child, r_size, relabel, queue, next_q, dfs;
    std::vector<double> r_lambda;
    std::vector<char> ignore;
};

}  // namespace

extern "C" int vg_hdbscan_tree_host(const int32_t* h_lo, const int32_t* h_hi, const double* h_w2, int n,
                                    int min_cluster_size, double eps, int32_t* h_labels, double* h_probs,
                                    int32_t* h_n_clusters) {
    if (n < 0 || (n > 1 && (!h_lo || !h_hi || !h_w2)) || !h_labels || !h_probs || min_cluster_size < 2) return 1;
    if (h_n_clusters) *h_n_clusters = 0;
    for (int i = 0; i < n; ++i) {
        h_labels[i] = -1;
        h_probs[i] = 0.0;
    }
    if (n <= min_cluster_size) return 0;
    const int m = n - 1;
    // ---- strict total order (w2, lo, hi): the GPU sorts by weight only, fix up runs of equal weight ----
    static thread_local Work ws;
    std::vector<int32_t>&lo_v = ws.lo, &hi_v = ws.hi;
    lo_v.assign(h_lo, h_lo + m);
    hi_v.assign(h_hi, h_hi + m);
    {
        std::vector<std::pair<int32_t, int32_t>>& run = ws.run;
        for (int i = 0; i < m;) {
            int j = i + 1;
            while (j < m && h_w2[j] == h_w2[i]) ++j;
            if (j - i > 1) {
                run.clear();
                for (int k = i; k < j; ++k) run.emplace_back(lo_v[k], hi_v[k]);
                std::sort(run.begin(), run.end());
                for (int k = i; k < j; ++k) { lo_v[k] = run[k - i].first; hi_v[k] = run[k - i].second; }
            }
            i = j;
        }
    }
    h_lo = lo_v.data();
    h_hi = hi_v.data();
    // ---- single linkage ------------------------------------------------------------------------
    // Union-find over the n POINTS (union by size, path halving) with the dendrogram node that currently stands for each set kept
    // beside the root: the library's form -- every merge makes the new node n + i the parent of both sets, sets found by walking the
    // 2n - 1 node forest -- builds chains that path compression has to flatten again and again (5.4 of the stage's 8.7 ms on 79k points).
    Tree& t = ws.t;
    t.n = n;
    t.left.resize(m); t.right.resize(m); t.size.resize(m); t.dist.resize(m);
    {
        std::vector<int>&parent = ws.parent, &top = ws.top, &sz = ws.sz;
        parent.resize(n); top.resize(n); sz.assign(n, 1);
        for (int i = 0; i < n; ++i) parent[i] = top[i] = i;
        auto find = [&](int x) {
            while (parent[x] != x) {
                parent[x] = parent[parent[x]];
                x = parent[x];
            }
            return x;
        };
        for (int i = 0; i < m; ++i) {
            const int ra = find(h_lo[i]), rb = find(h_hi[i]);
            if (ra == rb) return 1;   // not a tree
            t.left[i] = top[ra];
            t.right[i] = top[rb];
            t.dist[i] = sqrt(h_w2[i]);
            const int big = sz[ra] >= sz[rb] ? ra : rb, small = big == ra ? rb : ra;
            parent[small] = big;
            sz[big] += sz[small];
            top[big] = n + i;
            t.size[i] = sz[big];
        }
    }
    // ---- condense (BFS, ids in visiting order like the library) ----------------------------------
    // rows: parent cluster (0-based, 0 = root), child (point id, or cluster index if csize > 1), lambda, csize
    // (every point is one row; a cluster row pair needs two sides of >= min_cluster_size points: the row count is bounded, the arrays are
    // written by index -- four push_backs per row and a std::vector stack were 3.6 ms of this stage on 79k points)
    const size_t row_cap = (size_t)n + 2 * ((size_t)n / (size_t)min_cluster_size + 2);
    std::vector<int>&r_parent = ws.r_parent, &r_child = ws.r_child, &r_size = ws.r_size, &relabel = ws.relabel;
    std::vector<double>& r_lambda = ws.r_lambda;
    std::vector<char>& ignore = ws.ignore;
    r_parent.resize(row_cap); r_child.resize(row_cap); r_size.resize(row_cap); r_lambda.resize(row_cap);
    size_t nr = 0;
    relabel.assign(2 * n - 1, -1);
    ignore.assign(2 * n - 1, 0);
    int next_label = 1;
    const int root = 2 * n - 2;
    relabel[root] = 0;
    std::vector<int>&queue = ws.queue, &next_q = ws.next_q, &dfs = ws.dfs;
    std::vector<int> stack;
    queue.clear(); next_q.clear();
    dfs.resize(n + 1);                                   // a subtree's pending nodes: never more than its leaves
    queue.push_back(root);
    const int* tl = t.left.data();
    const int* tr = t.right.data();
    auto emit_leaves = [&](int node, int p, double lam) {
        int sp = 0;
        dfs[sp++] = node;
        while (sp > 0) {
            const int x = dfs[--sp];
            if (x < n) {
                r_parent[nr] = p; r_child[nr] = x; r_lambda[nr] = lam; r_size[nr] = 1;
                ++nr;
            } else {
                ignore[x] = 1;
                dfs[sp++] = tr[x - n];
                dfs[sp++] = tl[x - n];
            }
        }
    };
    const double INF = std::numeric_limits<double>::infinity();
    while (!queue.empty()) {
        next_q.clear();
        for (int node : queue) {
            if (node < n || ignore[node]) continue;
            const int l = tl[node - n], r = tr[node - n];
            const double d = t.dist[node - n];
            const double lam = d > 0.0 ? 1.0 / d : INF;
            const int lc = l >= n ? t.size[l - n] : 1, rc = r >= n ? t.size[r - n] : 1;
            const int p = relabel[node];
            if (lc >= min_cluster_size && rc >= min_cluster_size) {
                relabel[l] = next_label;
                r_parent[nr] = p; r_child[nr] = next_label++; r_lambda[nr] = lam; r_size[nr] = lc; ++nr;
                relabel[r] = next_label;
                r_parent[nr] = p; r_child[nr] = next_label++; r_lambda[nr] = lam; r_size[nr] = rc; ++nr;
            } else if (lc < min_cluster_size && rc < min_cluster_size) {
                emit_leaves(l, p, lam);
                emit_leaves(r, p, lam);
            } else if (lc < min_cluster_size) {
                relabel[r] = p;
                emit_leaves(l, p, lam);
            } else {
                relabel[l] = p;
                emit_leaves(r, p, lam);
            }
            if (l >= n && !ignore[l]) next_q.push_back(l);
            if (r >= n && !ignore[r]) next_q.push_back(r);
        }
        queue.swap(next_q);
    }
    const int nc = next_label;          // clusters 0..nc-1, 0 = root
    const size_t nrows = nr;
    // ---- stability, cluster tree ------------------------------------------------------------------
    std::vector<double> birth(nc, 0.0), stab(nc, 0.0), death(nc, 0.0);
    std::vector<int> cpar(nc, -1);
    for (size_t i = 0; i < nrows; ++i)
        if (r_size[i] > 1) {
            birth[r_child[i]] = r_lambda[i];
            cpar[r_child[i]] = r_parent[i];
        }
    for (size_t i = 0; i < nrows; ++i) {
        stab[r_parent[i]] += (r_lambda[i] - birth[r_parent[i]]) * (double)r_size[i];
        if (r_lambda[i] > death[r_parent[i]]) death[r_parent[i]] = r_lambda[i];
    }
    // children lists (CSR)
    std::vector<int> kid_off(nc + 1, 0), kids(nc > 1 ? nc - 1 : 0);
    for (int c = 1; c < nc; ++c) kid_off[cpar[c] + 1]++;
    for (int c = 0; c < nc; ++c) kid_off[c + 1] += kid_off[c];
    {
        std::vector<int> cur(kid_off.begin(), kid_off.end() - 1);
        for (int c = 1; c < nc; ++c) kids[cur[cpar[c]]++] = c;
    }
    // ---- excess of mass ----------------------------------------------------------------------------
    std::vector<char> selected(nc, 1);
    selected[0] = 0;
    for (int c = nc - 1; c >= 1; --c) {
        double sub = 0.0;
        for (int k = kid_off[c]; k < kid_off[c + 1]; ++k) sub += stab[kids[k]];
        if (sub > stab[c]) {
            selected[c] = 0;
            stab[c] = sub;
        } else {
            stack.clear();
            for (int k = kid_off[c]; k < kid_off[c + 1]; ++k) stack.push_back(kids[k]);
            while (!stack.empty()) {
                int k = stack.back();
                stack.pop_back();
                selected[k] = 0;
                for (int j = kid_off[k]; j < kid_off[k + 1]; ++j) stack.push_back(kids[j]);
            }
        }
    }
    // ---- cluster_selection_epsilon -------------------------------------------------------------------
    if (eps != 0.0 && nc > 1) {
        std::vector<char> cand(nc, 0);
        for (int c = 1; c < nc; ++c) {
            if (!selected[c]) continue;
            if (1.0 / birth[c] < eps) {
                int node = c;
                while (true) {
                    int p = cpar[node];
                    if (p == 0) break;
                    if (1.0 / birth[p] > eps) { node = p; break; }
                    node = p;
                }
                cand[node] = 1;
            } else
                cand[c] = 1;
        }
        // top-most candidate wins (descendants of a chosen ancestor are dropped, _tree.pyx:634-636)
        for (int c = 1; c < nc; ++c) {
            selected[c] = 0;
            if (!cand[c]) continue;
            bool nested = false;
            for (int a = cpar[c]; a > 0; a = cpar[a])
                if (cand[a]) { nested = true; break; }
            selected[c] = !nested;
        }
    }
    // ---- labels + probabilities -------------------------------------------------------------------------
    std::vector<int> label_of(nc, -1), owner(nc, -1);
    int nl = 0;
    for (int c = 1; c < nc; ++c)
        if (selected[c]) label_of[c] = nl++;
    for (int c = 1; c < nc; ++c) owner[c] = selected[c] ? c : owner[cpar[c]];
    for (size_t i = 0; i < nrows; ++i) {
        if (r_size[i] != 1) continue;
        const int o = owner[r_parent[i]];
        if (o < 0) continue;
        const int pt = r_child[i];
        h_labels[pt] = label_of[o];
        const double mx = death[o], lam = r_lambda[i];
        h_probs[pt] = (mx == 0.0 || std::isinf(lam)) ? 1.0 : std::min(lam, mx) / mx;
    }
    if (h_n_clusters) *h_n_clusters = nl;
    return 0;
}

// LidarFrame.generate_detections' grouping (src/vilgod/lidar_frame.py:163-167, 230-237) as one counting sort: labels of points whose
// membership probability is below the threshold become noise, the clusters come in ascending label order, each cluster's point indices
// ascending.  h_ids: [<= n] labels that own at least one point; h_index: [<= n] packed point indices; h_seg: [n_clusters + 1] offsets.
extern "C" int vg_pack_clusters_host(const int32_t* h_labels, const double* h_probs, int n, double threshold, int64_t* h_ids,
                                     int32_t* h_index, int32_t* h_seg, int32_t* h_n_clusters) {
    if (n < 0 || (n > 0 && !h_labels) || !h_ids || !h_index || !h_seg || !h_n_clusters) return 1;
    int max_label = -1;
    for (int i = 0; i < n; ++i) max_label = std::max(max_label, (int)h_labels[i]);
    int nc = 0;
    if ((long long)max_label >= 2ll * n + 1024) {
        // label VALUES far beyond the point count (not what the hierarchy stage produces: its labels are 0 .. clusters - 1): no table of that
        // size -- sort the kept points by (label, index) instead
        static thread_local std::vector<std::pair<int32_t, int32_t>> kept;
        kept.clear();
        for (int i = 0; i < n; ++i)
            if (h_labels[i] >= 0 && !(h_probs && h_probs[i] < threshold)) kept.emplace_back(h_labels[i], i);
        std::sort(kept.begin(), kept.end());
        for (size_t k = 0; k < kept.size(); ++k) {
            if (k == 0 || kept[k].first != kept[k - 1].first) {
                h_ids[nc] = kept[k].first;
                h_seg[nc++] = (int32_t)k;
            }
            h_index[k] = kept[k].second;
        }
        h_seg[nc] = (int32_t)kept.size();
        *h_n_clusters = nc;
        return 0;
    }
    static thread_local std::vector<int> count;
    count.assign((size_t)max_label + 2, 0);
    for (int i = 0; i < n; ++i) {
        const int l = h_labels[i];
        if (l >= 0 && !(h_probs && h_probs[i] < threshold)) count[l + 1]++;
    }
    int total = 0;
    for (int l = 0; l <= max_label; ++l) {
        const int c = count[l + 1];
        count[l + 1] = total;                              // start of label l's segment (if it has one)
        if (c > 0) {
            h_ids[nc] = l;
            h_seg[nc++] = total;
            total += c;
        }
    }
    h_seg[nc] = total;
    for (int i = 0; i < n; ++i) {
        const int l = h_labels[i];
        if (l >= 0 && !(h_probs && h_probs[i] < threshold)) h_index[count[l + 1]++] = i;
    }
    *h_n_clusters = nc;
    return 0;
}
