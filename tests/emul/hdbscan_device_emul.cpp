// CPU emulation of csrc/hdbscan_device.hip for the test-suite: the SAME per-element bodies (csrc/hdbscan_device.inc, compiled here by
// g++ without HD_DEVICE) run phase by phase, every "thread" of a phase in a loop, the library calls (radix sorts, the scan) replaced by
// std::stable_sort / a loop.  tests/test_hierarchy.py compares the result with vg_hdbscan_tree_host bit for bit on random trees: the rules
// of the data-parallel formulation are checked without a GPU; the kernels themselves are checked on the GPU against both.
// Test infrastructure only -- nothing in vilgod_amd/ loads this.
#include <algorithm>
#include <vector>

#include "hdbscan_device.inc"

extern "C" int hd_emul_tree(const int32_t* lo_in, const int32_t* hi_in, const double* w2_in, int n, int mcs, double eps,
                            int32_t* labels, double* probs, int32_t* n_clusters, int32_t* n_splits, int32_t* sweeps) {
    if (n_clusters) *n_clusters = 0;
    if (n_splits) *n_splits = 0;
    if (sweeps) *sweeps = 0;
    for (int i = 0; i < n; ++i) { labels[i] = -1; probs[i] = 0.0; }
    if (mcs < 2 || mcs > HD_MAX_MCS) return 1;
    if (n <= mcs) return 0;
    const int m = n - 1;
    // total order (w2, lo, hi) from the weight order the device is handed: every edge moves inside its run of equal weights
    std::vector<int> lo(m), hi(m);
    std::vector<double> w2(m);
    for (int i = 0; i < m; ++i) {
        const int pos = hd_tie_position(lo_in, hi_in, w2_in, m, i);
        lo[pos] = lo_in[i]; hi[pos] = hi_in[i]; w2[pos] = w2_in[i];
    }
    // adjacency, ascending rank per vertex
    std::vector<int> adj_off(n + 1, 0);
    for (int r = 0; r < m; ++r) { adj_off[lo[r] + 1]++; adj_off[hi[r] + 1]++; }
    for (int x = 0; x < n; ++x) adj_off[x + 1] += adj_off[x];
    std::vector<unsigned long long> adj(2 * (size_t)m);
    {
        std::vector<int> cur(adj_off.begin(), adj_off.end() - 1);
        for (int r = 0; r < m; ++r) {
            adj[cur[lo[r]]++] = ((unsigned long long)r << 32) | (unsigned)hi[r];
            adj[cur[hi[r]]++] = ((unsigned long long)r << 32) | (unsigned)lo[r];
        }
    }
    const int ncap = n / mcs + 2, ncl_cap = 2 * ncap + 1;
    std::vector<unsigned char> side(2 * (size_t)m), eflag(m), wins(ncl_cap), selected(ncl_cap), cand(ncl_cap);
    std::vector<int> kcnt(m), a(n, -1), uf(n), split_pos(m), S(ncap), nsv(1, 0), node(2 * ncap), sp_parent(ncap), sp_side(ncap), kid(2 * ncap),
        chainlen(ncl_cap, 0), npts(ncl_cap, 0), kw_parent(2 * ncap), kw_top(2 * ncap), nsub(ncap), tot(ncap), csize(ncl_cap),
        depth(ncap), pre(ncap), q(ncap), done(ncl_cap), sel_by_final(ncl_cap + 1), out_label(ncl_cap), ncl_out(1, 0);
    std::vector<unsigned> first(n);
    std::vector<double> death(ncl_cap, 0.0);
    std::vector<double> stab(ncl_cap), stab2(ncl_cap), out_death(ncl_cap), lam_split(ncap);
    std::vector<HdChainRec> crec(m);
    HdView v{};
    v.n = n; v.m = m; v.mcs = mcs; v.ncap = ncap; v.eps = eps;
    v.lo = lo.data(); v.hi = hi.data(); v.w2 = w2.data(); v.adj_off = adj_off.data(); v.adj = adj.data();
    v.side = side.data(); v.eflag = eflag.data(); v.kcnt = kcnt.data(); v.a = a.data(); v.uf = uf.data(); v.split_pos = split_pos.data();
    v.S = S.data(); v.ns = nsv.data(); v.first = first.data(); v.node = node.data(); v.sp_parent = sp_parent.data(); v.sp_side = sp_side.data();
    v.kid = kid.data(); v.crec = crec.data(); v.lam_split = lam_split.data(); v.chainlen = chainlen.data(); v.npts = npts.data(); v.death = death.data();
    v.kw_parent = kw_parent.data(); v.kw_top = kw_top.data(); v.nsub = nsub.data(); v.tot = tot.data();
    v.csize = csize.data(); v.depth = depth.data(); v.pre = pre.data(); v.q = q.data(); v.done = done.data(); v.stab = stab.data();
    v.stab2 = stab2.data(); v.wins = wins.data(); v.selected = selected.data(); v.cand = cand.data(); v.sel_by_final = sel_by_final.data();
    v.out_label = out_label.data(); v.out_death = out_death.data(); v.n_clusters = ncl_out.data(); v.labels = labels; v.probs = probs;
    std::vector<int> st_x(HD_MAX_MCS), st_i(HD_MAX_MCS);
    for (int i = 0; i < 2 * m; ++i) hd_side_count(v, i, st_x.data(), st_i.data(), 1);
    for (int i = 0; i < 2 * m; ++i) hd_side_assign(v, i, st_x.data(), st_i.data(), 1);
    for (int x = 0; x < n; ++x) if (a[x] < 0) return 2;                 // R2: every point leaves at exactly one chain node
    for (int x = 0; x < n; ++x) uf[x] = x;
    for (int r = m - 1; r >= 0; --r) hd_segment_union(v, r);            // (any order)
    for (int x = 0; x < n; ++x) hd_segment_flatten(v, x);
    for (int r = 0, acc = 0; r < m; ++r) { split_pos[r] = acc; acc += (eflag[r] & HD_SPLIT) ? 1 : 0; }
    for (int r = 0; r < m; ++r) hd_split_scatter(v, r);
    const int ns = nsv[0];
    if (ns > ncap - 1) return 3;
    if (n_splits) *n_splits = ns;
    for (int i = 0; i < 2 * ns; ++i) hd_split_nodes(v, i);
    hd_kruskal_splits(v, ns, kw_parent.data(), kw_top.data());
    const int ncl = 2 * ns + 1;
    std::vector<unsigned> ckey(m), crank(m);
    for (int r = 0; r < m; ++r) {
        crec[r] = hd_chain_rec(v, r, hd_chain_find(v, r));
        if (crec[r].c == 0) chainlen[0]++;
        ckey[r] = hd_chain_sortkey(crec[r]); crank[r] = (unsigned)r;
    }
    std::stable_sort(crank.begin(), crank.end(), [&](unsigned x, unsigned y) { return ckey[x] < ckey[y]; });
    {
        std::vector<unsigned> ks(m);
        for (int i = 0; i < m; ++i) ks[i] = ckey[crank[i]];
        ckey.swap(ks);
    }
    v.chain_key = ckey.data(); v.chain_rank = crank.data();
    for (int c = 1; c < ncl; ++c) hd_chain_stats(v, c);
    int total_sweeps = 0;
    auto relax = [&](auto&& body) {
        std::fill(done.begin(), done.end(), 0);
        for (int sweep = 0;; ++sweep) {
            int changed = 0;
            for (int k = 0; k < ns; ++k) changed |= body(k, sweep);
            ++total_sweeps;
            if (!changed) break;
        }
    };
    relax([&](int k, int sweep) { return hd_up_all(v, k, sweep); });
    relax([&](int k, int sweep) { return hd_down_order(v, k, sweep); });
    for (int k = 0; k < ns; ++k) {
        int cnt = 0;
        for (int j = 0; j < ns; ++j) cnt += hd_bfs_before(depth[j], pre[j], depth[k], pre[k]) ? 1 : 0;
        q[k] = cnt;
    }
    for (int c = 1; c < ncl; ++c) hd_select_eom(v, c);
    const bool use_eps = eps != 0.0 && ncl > 1;
    if (use_eps) {
        for (int c = 1; c < ncl; ++c) hd_eps_candidates(v, c);
        for (int c = 1; c < ncl; ++c) hd_eps_select(v, c);
    }
    for (int c = 0; c < ncl; ++c) hd_selected_by_final(v, c, use_eps);
    int acc = 0;
    for (int f = 0; f < ncl; ++f) { const int s = sel_by_final[f]; sel_by_final[f] = acc; acc += s; }
    if (n_clusters) *n_clusters = acc;
    for (int c = 0; c < ncl; ++c) hd_owner(v, c);
    for (int p = 0; p < n; ++p) hd_point(v, p);
    if (sweeps) *sweeps = total_sweeps;
    return 0;
}
