// HDBSCAN hierarchy stage ON THE DEVICE (round 6; SURVEY §8a row B2, kernel list K2d): from the minimum spanning tree of the
// mutual-reachability graph, as csrc/cluster.hip leaves it in device memory (edges sorted by weight), to flat labels and membership
// probabilities -- no copy of the tree to the host, no host thread between the tree and the clusters' first kernel.
//
// Replaces the tail of `cluster_model.fit(points_ref_wo_ground)` (src/vilgod/zero_shot_detector.py:248; hdbscan.HDBSCAN(
// min_cluster_size=15, cluster_selection_epsilon=0.15), tools/configs/preprocessor/waymo.yaml:10-15) and produces, bit for bit, what
// the host stage csrc/hdbscan_tree.cpp produces (tests/test_hierarchy.py).  The host stage is a sequential union-find over all n - 1
// edges; here every step is a rule with bounded local work per edge / point / cluster (hdbscan_device.inc: R1..R6, each checked against
// the host stage on the CPU by the test-suite's emulation of the same bodies):
//
//   k_hd_order                         the strict (w2, lo, hi) order the host stage uses: every edge to its place inside its run of
//                                      equal weights; the 2 (n - 1) half-edges
//   radix sort + k_hd_offsets          the half-edges by vertex (stable: by ascending rank inside a vertex)
//   k_hd_side<count / assign>          per (edge, side): vertices reached over lower-rank edges, at most min_cluster_size   (R1, R2)
//   k_hd_union / k_hd_flatten          lock-free union-find over the non-split edges: segments                             (R3)
//   scan + k_hd_scatter                the split edges, a few hundred, in rank order
//   k_hd_tree_a                        ONE lane, union-find in LDS: Kruskal over the splits on the segments = the cluster tree
//   k_hd_chain + radix sort            every chain node's cluster (a walk up the cluster tree); the chains grouped by cluster  (R4)
//   k_hd_chain_stats                   a wave per cluster: its chain by descending rank, summed in the library's order     (R6)
//   k_hd_tree_bc                       ONE workgroup: cluster sizes, the stabilities' last rows, excess of mass; dendrogram depths and
//                                      the library's BFS numbering (R5); epsilon; flat labels
// The cluster tree's arrays (a few hundred splits) are staged in LDS by every kernel that sweeps or walks the tree (HdCarve).
//   k_hd_points                        label and probability of every point
// The one-workgroup kernel sweeps the cluster tree level by level (as many sweeps, one barrier each, as it is high).  float64 throughout; compiled with
// -ffp-contract=off (the stability sums are the host stage's expressions, unfused).
#include <string.h>
#include <algorithm>
#include "common.h"
#include <rocprim/rocprim.hpp>

#include "vilgod_hip.h"
#define HD_DEVICE
#include "hdbscan_device.inc"

#define HD_NT 256                 // threads per workgroup of the per-element kernels
#define HD_TREE_LDS_INTS 30720    // dynamic LDS of k_hd_tree_a (120 KB): ten ints per split -> up to 3072 splits, else global memory

struct vg_hier {
    int max_points;
    char* slab;
    void* d_temp;
    size_t temp_bytes;
    int *lo, *hi;                  // the tree in (w2, lo, hi) order
    double* w2;
    unsigned *he_key, *he_key_s;   // half-edges: vertex, (rank << 32 | other endpoint)
    unsigned long long *he_val, *adj;
    int* adj_off;
    unsigned long long* bfs_key;
    long long* stamps;              // development build: 100 MHz time stamps of the one-workgroup kernels' phases (NULL in the product)
    unsigned *ckey, *ckey_s, *crank, *crank_s;   // chain nodes: sort key (cluster), rank
    HdView v;                      // the arrays of the stage (n, m, mcs, eps, inputs filled per call)
};

namespace {

struct HdSplitFlag {
    __device__ int operator()(unsigned char f) const { return (int)(f & HD_SPLIT); }
};

__global__ __launch_bounds__(HD_NT) void k_hd_order(int m, const int* __restrict__ lo_in, const int* __restrict__ hi_in, const double* __restrict__ w2_in,
                                                    int* __restrict__ lo, int* __restrict__ hi, double* __restrict__ w2,
                                                    unsigned* __restrict__ he_key, unsigned long long* __restrict__ he_val) {
    const int i = blockIdx.x * HD_NT + threadIdx.x;
    if (i >= m) return;
    const int r = hd_tie_position(lo_in, hi_in, w2_in, m, i);
    const int a = lo_in[i], b = hi_in[i];
    lo[r] = a; hi[r] = b; w2[r] = w2_in[i];
    he_key[2 * r] = (unsigned)a;     he_val[2 * r] = ((unsigned long long)r << 32) | (unsigned)b;
    he_key[2 * r + 1] = (unsigned)b; he_val[2 * r + 1] = ((unsigned long long)r << 32) | (unsigned)a;
}

__global__ __launch_bounds__(HD_NT) void k_hd_offsets(int n2, int n, const unsigned* __restrict__ key_s, int* __restrict__ adj_off) {
    const int i = blockIdx.x * HD_NT + threadIdx.x;
    if (i == 0) adj_off[n] = n2;
    if (i >= n2) return;
    if (i == 0 || key_s[i] != key_s[i - 1]) adj_off[key_s[i]] = i;          // (a tree: every vertex has an edge)
}

__global__ __launch_bounds__(HD_NT) void k_hd_init(HdView v) {
    const int i = blockIdx.x * HD_NT + threadIdx.x;
    if (i < v.n) v.uf[i] = i;
    if (i == 0) { *v.ns = 0; *v.n_clusters = 0; v.chainlen[0] = 0; }
}

template <bool ASSIGN>
__global__ __launch_bounds__(HD_NT) void k_hd_side(HdView v) {
    extern __shared__ int hd_stack[];                          // [2][mcs][HD_NT]
    const int i = blockIdx.x * HD_NT + threadIdx.x;
    if (i >= 2 * v.m) return;
    int* st_x = hd_stack + threadIdx.x;
    int* st_i = hd_stack + v.mcs * HD_NT + threadIdx.x;
    if (ASSIGN) hd_side_assign(v, i, st_x, st_i, HD_NT);
    else hd_side_count(v, i, st_x, st_i, HD_NT);
}

__global__ __launch_bounds__(HD_NT) void k_hd_union(HdView v) {
    const int r = blockIdx.x * HD_NT + threadIdx.x;
    if (r < v.m) hd_segment_union(v, r);
}
__global__ __launch_bounds__(HD_NT) void k_hd_flatten(HdView v) {
    const int x = blockIdx.x * HD_NT + threadIdx.x;
    if (x < v.n) hd_segment_flatten(v, x);
}
__global__ __launch_bounds__(HD_NT) void k_hd_scatter(HdView v) {
    const int r = blockIdx.x * HD_NT + threadIdx.x;
    if (r < v.m) hd_split_scatter(v, r);
}

// ONE workgroup: the union-find nodes of the splits' endpoints, then Kruskal over the splits by ONE lane -- every array it touches in
// LDS (a dependent global-memory access per step would cost a microsecond each), written out afterwards.
// The LDS form and the global-memory form are two instantiations of one body: with ONE body behind "pointer = LDS or global" every access
// is a FLAT instruction (the address space decided per access at run time), and a flat access to LDS costs several times a ds_read
// (round 6: 265 us of Kruskal for 682 splits at 2.4 GHz = 950 cycles per split).
#define HD_STAMP(I) if (stamps && threadIdx.x == 0) stamps[I] = (long long)wall_clock64();
__device__ __forceinline__ void hd_tree_a_body(const HdView& w, int ns, int* par, int* top, long long* stamps) {
    for (int i = threadIdx.x; i < 2 * ns; i += blockDim.x) hd_split_nodes(w, i);
    __syncthreads();
    HD_STAMP(1)
    // (ONE lane.  Measured and dropped, round 6: the same steps run wave-uniformly by 64 lanes -- node pairs of 64 splits prefetched into
    // registers, union by size -- took 312 us instead of 265 for 682 splits: the scalarising readfirstlanes cost more than the loads saved;
    // TWO lanes, one per endpoint (finds side by side, roots and sizes crossed by a DPP swap, union by size): 214 against 220 us.  A split
    // costs ~750 cycles = three dependent LDS round trips of a lone wave, however the work around them is arranged.)
    if (threadIdx.x == 0) hd_kruskal_splits(w, ns, par, top);
    __syncthreads();
    HD_STAMP(2)
}
__global__ __launch_bounds__(1024) void k_hd_tree_a(HdView v, long long* __restrict__ stamps) {
    extern __shared__ int lds[];
    const int t = threadIdx.x, T = blockDim.x;
    const int ns = *v.ns;
    if (ns == 0) return;
    HD_STAMP(0)
    if (10 * ns <= HD_TREE_LDS_INTS) {
        HdView w = v;
        w.node = lds; w.kid = lds + 6 * ns; w.sp_parent = lds + 8 * ns; w.sp_side = lds + 9 * ns;
        hd_tree_a_body(w, ns, lds + 2 * ns, lds + 4 * ns, stamps);
        for (int i = t; i < 2 * ns; i += T) v.kid[i] = lds[6 * ns + i];
        for (int i = t; i < ns; i += T) { v.sp_parent[i] = lds[8 * ns + i]; v.sp_side[i] = lds[9 * ns + i]; }
    } else
        hd_tree_a_body(v, ns, v.kw_parent, v.kw_top, stamps);
}

// LDS staging of the cluster tree's small arrays (a few hundred splits): the sweeps and walks over the tree are chains of dependent
// accesses -- ~100 ns each in LDS, a microsecond each in global memory.  A kernel carves its pieces out of its dynamic LDS, copies the
// inputs in, points a local HdView at them (the bodies of hdbscan_device.inc do not care where an array lives) and copies results out.
struct HdCarve {
    int* base;
    int used;
};
template <typename T>
__device__ __forceinline__ T* hd_carve(HdCarve& cv, const T* g, int count, bool copy_in) {
    cv.used = (cv.used + 1) & ~1;                                  // 8-byte alignment
    T* p = (T*)(cv.base + cv.used);
    cv.used += (int)(((size_t)count * sizeof(T) + 3) / 4);
    if (copy_in) for (int i = threadIdx.x; i < count; i += blockDim.x) p[i] = g[i];
    return p;
}
template <typename T>
__device__ __forceinline__ void hd_copy_out(T* g, const T* l, int count) {
    for (int i = threadIdx.x; i < count; i += blockDim.x) g[i] = l[i];
}

#define HD_CHAIN_LDS_INTS 12288     // k_hd_chain: S, sp_parent, sp_side of up to 4096 splits per workgroup (48 KB)
__device__ __forceinline__ void hd_chain_body(const HdView& v, const HdView& w, unsigned* __restrict__ ckey, unsigned* __restrict__ crank) {
    const int r = blockIdx.x * HD_NT + threadIdx.x;
    const int c = r < v.m ? hd_chain_find(w, r) : -1;
    if (r < v.m) {
        const HdChainRec rec = hd_chain_rec(v, r, c);
        v.crec[r] = rec;
        ckey[r] = hd_chain_sortkey(rec); crank[r] = (unsigned)r;
    }
    // of the root (the points no cluster holds: most chain nodes) only the chain length is read: one atomic per wave
    const unsigned long long roots = __ballot(c == 0);
    if (c == 0 && (int)(threadIdx.x & 63) == __ffsll((long long)roots) - 1) atomicAdd(&v.chainlen[0], __popcll(roots));
}
__global__ __launch_bounds__(HD_NT) void k_hd_chain(HdView v, unsigned* __restrict__ ckey, unsigned* __restrict__ crank) {
    extern __shared__ int lds[];
    const int ns = *v.ns;
    if (3 * ns <= HD_CHAIN_LDS_INTS) {                             // the walk up the cluster tree: its three arrays in LDS
        for (int i = threadIdx.x; i < ns; i += HD_NT) { lds[i] = v.S[i]; lds[ns + i] = v.sp_parent[i]; lds[2 * ns + i] = v.sp_side[i]; }
        __syncthreads();
        HdView w = v;
        w.S = lds; w.sp_parent = lds + ns; w.sp_side = lds + 2 * ns;
        hd_chain_body(v, w, ckey, crank);
    } else
        hd_chain_body(v, v, ckey, crank);
}

// a wave per cluster over its run of the sorted chain nodes: chain length, points, largest lambda, and the rows of its own chain summed
// in the library's order (descending rank)
__global__ __launch_bounds__(HD_NT) void k_hd_chain_stats(HdView v) {
    const int ns = *v.ns, ncl = 2 * ns + 1;
    const int lane = threadIdx.x & 63, nw = gridDim.x * (HD_NT / 64);
    for (int c = 1 + (int)((blockIdx.x * HD_NT + threadIdx.x) >> 6); c < ncl; c += nw) {
        const int b = hd_lower_bound_u32(v.chain_key, v.m, (unsigned)c), e = hd_lower_bound_u32(v.chain_key, v.m, (unsigned)c + 1u);
        const double birth = hd_birth(v, c);
        double s = 0.0;
        int np = 0;
        // 64 records per step, the next step's records requested before this step's additions; the additions strictly in order, one
        // chain node after the other out of the lanes (v_readlane with a constant lane: no loop counter, no LDS round trip per node)
        auto fetch = [&](int top) {
            const int i = top - lane;                              // lane 0: the highest rank of the step
            return i >= b ? v.crec[v.chain_rank[i]] : HdChainRec{-1, 0, 0.0};
        };
        HdChainRec nxt = fetch(e - 1);
        for (int top = e - 1; top >= b; top -= 64) {
            const HdChainRec rec = nxt;
            nxt = fetch(top - 64);
            const int cnt = min(64, top - b + 1);
            const double tv = (rec.lam - birth) * 1.0;
            const int tlo = __double2loint(tv), thi = __double2hiint(tv);
            np += rec.kc;
            if (cnt == 64 && __ballot(rec.kc != 1) == 0ull) {      // the common step: 64 nodes of one point each -- 64 additions, no branch
#pragma unroll
                for (int j = 0; j < 64; ++j) s += __hiloint2double(__builtin_amdgcn_readlane(thi, j), __builtin_amdgcn_readlane(tlo, j));
            } else {
#pragma unroll
                for (int j = 0; j < 64; ++j) {
                    if (j < cnt) {
                        const double tt = __hiloint2double(__builtin_amdgcn_readlane(thi, j), __builtin_amdgcn_readlane(tlo, j));
                        const int kk = __builtin_amdgcn_readlane(rec.kc, j);
                        if (kk == 1) s += tt;
                        else s = hd_stab_terms(s, tt, kk);
                    }
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) np += __shfl_xor(np, off);
        if (lane == 0) {
            v.chainlen[c] = e - b; v.npts[c] = np; v.stab[c] = s;
            v.death[c] = e > b ? v.crec[v.chain_rank[b]].lam : 0.0;
        }
    }
}

// ONE workgroup over the cluster tree, its arrays in LDS: bottom-up (sizes, the stabilities' last rows, excess of mass), top-down (dendrogram
// depths, preorder), the library's BFS numbering, the selection, cluster_selection_epsilon, flat labels.  A sweep = one barrier.
#define HD_SWEEPS(BODY)                                                                    \
    for (int i = t; i < ns; i += T) v.done[i] = 0;                                         \
    if (t == 0) s_flag[0] = 0;                                                             \
    __syncthreads();                                                                       \
    for (int sweep = 0;; ++sweep) {                                                        \
        if (t == 0) s_flag[(sweep + 1) % 3] = 0;                                           \
        int ch = 0;                                                                        \
        for (int i = t; i < ns; i += T) ch |= (BODY);                                      \
        if (ch) s_flag[sweep % 3] = 1;                                                     \
        __syncthreads();                                                                   \
        if (!s_flag[sweep % 3]) break;                                                     \
    }
#define HD_TREE_BC_LDS_INTS 38912      // 152 KB of dynamic LDS (+ the scan's 2 KB): up to ~1100 splits, else global memory
__device__ __forceinline__ void hd_tree_bc_body(const HdView& v, unsigned long long* bfs_key, int* n_clusters, int ns, int* s_flag, int* s_part,
                                                long long* stamps) {
    const int t = threadIdx.x, T = blockDim.x, ncl = 2 * ns + 1;
    HD_STAMP(5)
    HD_SWEEPS(hd_up_all(v, i, sweep))
    HD_STAMP(6)
    HD_SWEEPS(hd_down_order(v, i, sweep))
    HD_STAMP(7)
    // BFS position of a split = its place in (depth, preorder) order: a bitonic sort of depth << 40 | preorder << 20 | split over the next
    // power of two (a step = one barrier; counting "who is in front of me" was ns^2 / T reads per thread: 60-180 us for 682 splits)
    int P = 1;
    while (P < ns) P <<= 1;
    for (int i = t; i < P; i += T)
        bfs_key[i] = i < ns ? ((unsigned long long)(unsigned)v.depth[i] << (2 * HD_RANK_BITS)) | ((unsigned long long)(unsigned)v.pre[i] << HD_RANK_BITS) | (unsigned)i : ~0ull;
    __syncthreads();
    for (int k2 = 2; k2 <= P; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = t; i < P; i += T) {
                const int x = i ^ j;
                if (x > i) {
                    const unsigned long long a = bfs_key[i], b = bfs_key[x];
                    if ((a > b) == ((i & k2) == 0)) { bfs_key[i] = b; bfs_key[x] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = t; i < ns; i += T) v.q[(int)(bfs_key[i] & ((1u << HD_RANK_BITS) - 1u))] = i;
    __syncthreads();
    HD_STAMP(8)
    for (int c = 1 + t; c < ncl; c += T) hd_select_eom(v, c);
    __syncthreads();
    HD_STAMP(9)
    const bool use_eps = v.eps != 0.0 && ncl > 1;
    if (use_eps) {
        for (int c = 1 + t; c < ncl; c += T) hd_eps_candidates(v, c);
        __syncthreads();
        for (int c = 1 + t; c < ncl; c += T) hd_eps_select(v, c);
        __syncthreads();
    }
    HD_STAMP(10)
    for (int c = t; c < ncl; c += T) hd_selected_by_final(v, c, use_eps);
    __syncthreads();
    // exclusive scan of sel_by_final[0 .. ncl): a chunk per thread
    const int per = (ncl + T - 1) / T, b0 = t * per < ncl ? t * per : ncl, b1 = b0 + per < ncl ? b0 + per : ncl;
    int sum = 0;
    for (int f = b0; f < b1; ++f) sum += v.sel_by_final[f];
    s_part[t] = sum;
    __syncthreads();
    for (int off = 1; off < T; off <<= 1) {
        const int add = t >= off ? s_part[t - off] : 0;
        __syncthreads();
        s_part[t] += add;
        __syncthreads();
    }
    int run = s_part[t] - sum;
    for (int f = b0; f < b1; ++f) { const int sf = v.sel_by_final[f]; v.sel_by_final[f] = run; run += sf; }
    if (t == T - 1) *n_clusters = s_part[T - 1];
    __syncthreads();
    HD_STAMP(11)
    for (int c = t; c < ncl; c += T) hd_owner(v, c);
    __syncthreads();
    HD_STAMP(12)
}
__global__ __launch_bounds__(512) void k_hd_tree_bc(HdView gv, unsigned long long* __restrict__ g_bfs_key, long long* __restrict__ stamps) {
    extern __shared__ int lds[];
    const int ns = *gv.ns, ncl = 2 * ns + 1;
    __shared__ int s_flag[3];
    __shared__ int s_part[512];
    HD_STAMP(4)
    if (stamps && threadIdx.x == 0) stamps[14] = clock64();
    if (12 * ns + 11 * ncl + 64 <= HD_TREE_BC_LDS_INTS) {          // (two instantiations of the body: see k_hd_tree_a)
        HdView v = gv;
        HdCarve cv{lds, 0};
        v.kid = hd_carve(cv, gv.kid, 2 * ns, true); v.sp_parent = hd_carve(cv, gv.sp_parent, ns, true); v.sp_side = hd_carve(cv, gv.sp_side, ns, true);
        v.lam_split = hd_carve(cv, gv.lam_split, ns, true);
        v.tot = hd_carve(cv, gv.tot, ns, false); v.nsub = hd_carve(cv, gv.nsub, ns, false); v.depth = hd_carve(cv, gv.depth, ns, false);
        v.pre = hd_carve(cv, gv.pre, ns, false); v.done = hd_carve(cv, gv.done, ns, false); v.q = hd_carve(cv, gv.q, ns, false);
        v.npts = hd_carve(cv, gv.npts, ncl, true); v.chainlen = hd_carve(cv, gv.chainlen, ncl, true);
        v.csize = hd_carve(cv, gv.csize, ncl, false); v.stab = hd_carve(cv, gv.stab, ncl, true); v.stab2 = hd_carve(cv, gv.stab2, ncl, false);
        v.death = hd_carve(cv, gv.death, ncl, true); v.sel_by_final = hd_carve(cv, gv.sel_by_final, ncl + 1, false);
        v.wins = hd_carve(cv, gv.wins, ncl, false); v.selected = hd_carve(cv, gv.selected, ncl, false); v.cand = hd_carve(cv, gv.cand, ncl, false);
        // (free once the bottom-up sweeps are over: stab2 -> the BFS keys, stab -> out_death, csize -> out_label)
        v.out_death = v.stab; v.out_label = v.csize;
        __syncthreads();
        hd_tree_bc_body(v, (unsigned long long*)v.stab2, gv.n_clusters, ns, s_flag, s_part, stamps);
        hd_copy_out(gv.out_label, v.out_label, ncl); hd_copy_out(gv.out_death, v.out_death, ncl);
    } else
        hd_tree_bc_body(gv, g_bfs_key, gv.n_clusters, ns, s_flag, s_part, stamps);
    __syncthreads();
    HD_STAMP(13)
    if (stamps && threadIdx.x == 0) stamps[15] = clock64();
}
#undef HD_SWEEPS
#undef HD_STAMP

__global__ __launch_bounds__(HD_NT) void k_hd_points(HdView v) {
    const int p = blockIdx.x * HD_NT + threadIdx.x;
    if (p < v.n) hd_point(v, p);
}
__global__ __launch_bounds__(HD_NT) void k_hd_noise(int n, int* __restrict__ labels, double* __restrict__ probs, int* __restrict__ ncl) {
    const int p = blockIdx.x * HD_NT + threadIdx.x;
    if (p < n) { labels[p] = -1; probs[p] = 0.0; }
    if (p == 0 && ncl) *ncl = 0;
}

template <typename T>
T* hd_take(char*& p, size_t count) {
    T* r = (T*)p;
    p += (count * sizeof(T) + 255) & ~(size_t)255;
    return r;
}

VgPerDeviceOnce hd_tree_a_lds, hd_tree_bc_lds;

}  // namespace

extern "C" {

int vg_hier_create(vg_hier** out, int max_points) {
    if (!out || max_points < 2 || max_points > (1 << HD_RANK_BITS)) return VG_ERR_ARG;
    vg_hier* h = new vg_hier();
    h->max_points = max_points;
    const size_t n = (size_t)max_points, m = n, ncap = n / 2 + 2, ncl = 2 * ncap + 2;       // (min_cluster_size >= 2)
    // one slab: the first pass sizes it, the second hands the pieces out
    for (int pass = 0; pass < 2; ++pass) {
        char* p = pass ? h->slab : (char*)nullptr;
        HdView& v = h->v;
        h->lo = hd_take<int>(p, m); h->hi = hd_take<int>(p, m); h->w2 = hd_take<double>(p, m);
        h->he_key = hd_take<unsigned>(p, 2 * m); h->he_key_s = hd_take<unsigned>(p, 2 * m);
        h->he_val = hd_take<unsigned long long>(p, 2 * m); h->adj = hd_take<unsigned long long>(p, 2 * m);
        h->adj_off = hd_take<int>(p, n + 1);
        h->bfs_key = hd_take<unsigned long long>(p, 2 * ncap);
        h->ckey = hd_take<unsigned>(p, m); h->ckey_s = hd_take<unsigned>(p, m); h->crank = hd_take<unsigned>(p, m); h->crank_s = hd_take<unsigned>(p, m);
        v.side = hd_take<unsigned char>(p, 2 * m); v.eflag = hd_take<unsigned char>(p, m);
        v.kcnt = hd_take<int>(p, m); v.a = hd_take<int>(p, n); v.uf = hd_take<int>(p, n); v.split_pos = hd_take<int>(p, m);
        v.S = hd_take<int>(p, ncap); v.ns = hd_take<int>(p, 1); v.first = hd_take<unsigned>(p, n);
        v.node = hd_take<int>(p, 2 * ncap); v.sp_parent = hd_take<int>(p, ncap); v.sp_side = hd_take<int>(p, ncap);
        v.kid = hd_take<int>(p, 2 * ncap); v.crec = hd_take<HdChainRec>(p, m); v.lam_split = hd_take<double>(p, ncap);
        v.chainlen = hd_take<int>(p, ncl); v.npts = hd_take<int>(p, ncl); v.death = hd_take<double>(p, ncl);
        v.kw_parent = hd_take<int>(p, 2 * ncap); v.kw_top = hd_take<int>(p, 2 * ncap);
        v.nsub = hd_take<int>(p, ncap); v.tot = hd_take<int>(p, ncap); v.csize = hd_take<int>(p, ncl);
        v.depth = hd_take<int>(p, ncap); v.pre = hd_take<int>(p, ncap); v.q = hd_take<int>(p, ncap); v.done = hd_take<int>(p, ncl);
        v.stab = hd_take<double>(p, ncl); v.stab2 = hd_take<double>(p, ncl);
        v.wins = hd_take<unsigned char>(p, ncl); v.selected = hd_take<unsigned char>(p, ncl); v.cand = hd_take<unsigned char>(p, ncl);
        v.sel_by_final = hd_take<int>(p, ncl + 1); v.out_label = hd_take<int>(p, ncl); v.out_death = hd_take<double>(p, ncl);
        v.n_clusters = hd_take<int>(p, 1);
        if (!pass) {
            const size_t bytes = (size_t)(p - (char*)nullptr);
            if (hipMalloc((void**)&h->slab, bytes) != hipSuccess) { delete h; return VG_ERR_HIP; }
        }
    }
    size_t t3 = 0, t4 = 0, t5 = 0;
    (void)rocprim::radix_sort_pairs(nullptr, t3, h->he_key, h->he_key_s, h->he_val, h->adj, 2 * m, 0, HD_RANK_BITS);
    (void)rocprim::exclusive_scan(nullptr, t4, rocprim::make_transform_iterator(h->v.eflag, HdSplitFlag()), h->v.split_pos, 0, m, rocprim::plus<int>());
    (void)rocprim::radix_sort_pairs(nullptr, t5, h->ckey, h->ckey_s, h->crank, h->crank_s, m, 0, HD_CHAIN_KEY_BITS);
    h->temp_bytes = std::max(std::max(t3, t4), t5) + 256;
    if (hipMalloc(&h->d_temp, h->temp_bytes) != hipSuccess) { (void)hipFree(h->slab); delete h; return VG_ERR_HIP; }
    h->stamps = nullptr;
#ifdef VG_DEV
    if (hipMalloc((void**)&h->stamps, 16 * sizeof(long long)) != hipSuccess) h->stamps = nullptr;
#endif
    *out = h;
    return VG_OK;
}

#ifdef VG_DEV
/* development aid (tools/dev/vilgod_hip_dev.h): the phase stamps of the last call's one-workgroup kernels, 100 MHz ticks */
int vg_hier_stamps(vg_hier* h, int64_t* h_out16) {
    if (!h || !h->stamps || !h_out16) return VG_ERR_ARG;
    VG_CHECK(hipMemcpy(h_out16, h->stamps, 16 * sizeof(long long), hipMemcpyDeviceToHost));
    return VG_OK;
}
#endif

int vg_hier_destroy(vg_hier* h) {
    if (!h) return VG_ERR_ARG;
    if (h->stamps) (void)hipFree(h->stamps);
    (void)hipFree(h->slab);
    (void)hipFree(h->d_temp);
    delete h;
    return VG_OK;
}

int vg_hdbscan_tree_device(vg_hier* h, const int32_t* d_lo, const int32_t* d_hi, const double* d_w2, int n, int min_cluster_size, double eps,
                           int32_t* d_labels, double* d_probs, int32_t* d_n_clusters, void* stream) {
    if (!h || n < 0 || !d_labels || !d_probs || min_cluster_size < 2 || min_cluster_size > HD_MAX_MCS) return VG_ERR_ARG;
    if (n > h->max_points) return VG_ERR_CAPACITY;
    hipStream_t st = (hipStream_t)stream;
    if (n <= min_cluster_size) {
        if (n > 0 || d_n_clusters) hipLaunchKernelGGL(k_hd_noise, dim3(vg_div_up(std::max(n, 1), HD_NT)), dim3(HD_NT), 0, st, n, d_labels, d_probs, d_n_clusters);
        VG_LAUNCH_CHECK();
        return VG_OK;
    }
    if (!d_lo || !d_hi || !d_w2) return VG_ERR_ARG;
    const int m = n - 1;
    HdView v = h->v;
    v.n = n; v.m = m; v.mcs = min_cluster_size; v.ncap = n / min_cluster_size + 2; v.eps = eps;
    v.lo = h->lo; v.hi = h->hi; v.w2 = h->w2; v.adj_off = h->adj_off; v.adj = h->adj;
    v.labels = d_labels; v.probs = d_probs;
    const dim3 B(HD_NT);
    const int gm = vg_div_up(m, HD_NT), gn = vg_div_up(n, HD_NT), g2m = vg_div_up(2 * m, HD_NT);
    size_t tb;
    // ---- the (w2, lo, hi) order; adjacency by ascending rank ----
    hipLaunchKernelGGL(k_hd_order, dim3(gm), B, 0, st, m, d_lo, d_hi, d_w2, h->lo, h->hi, h->w2, h->he_key, h->he_val);
    tb = h->temp_bytes;
    VG_CHECK(rocprim::radix_sort_pairs(h->d_temp, tb, h->he_key, h->he_key_s, h->he_val, h->adj, (size_t)(2 * m), 0, HD_RANK_BITS, st));
    hipLaunchKernelGGL(k_hd_offsets, dim3(g2m), B, 0, st, 2 * m, n, h->he_key_s, h->adj_off);
    hipLaunchKernelGGL(k_hd_init, dim3(gn), B, 0, st, v);
    // ---- R1, R2 ----
    const size_t stack_bytes = (size_t)2 * min_cluster_size * HD_NT * sizeof(int);
    hipLaunchKernelGGL((k_hd_side<false>), dim3(g2m), B, stack_bytes, st, v);
    hipLaunchKernelGGL((k_hd_side<true>), dim3(g2m), B, stack_bytes, st, v);
    // ---- R3 ----
    hipLaunchKernelGGL(k_hd_union, dim3(gm), B, 0, st, v);
    hipLaunchKernelGGL(k_hd_flatten, dim3(gn), B, 0, st, v);
    tb = h->temp_bytes;
    VG_CHECK(rocprim::exclusive_scan(h->d_temp, tb, rocprim::make_transform_iterator(v.eflag, HdSplitFlag()), v.split_pos, 0, (size_t)m,
                                     rocprim::plus<int>(), st));
    hipLaunchKernelGGL(k_hd_scatter, dim3(gm), B, 0, st, v);
    if (vg_max_dynamic_lds((const void*)k_hd_tree_a, HD_TREE_LDS_INTS * 4, hd_tree_a_lds) != VG_OK) return VG_ERR_HIP;
    hipLaunchKernelGGL(k_hd_tree_a, dim3(1), dim3(1024), HD_TREE_LDS_INTS * 4, st, v, h->stamps);
    // ---- R4 .. R6 ----
    hipLaunchKernelGGL(k_hd_chain, dim3(gm), B, HD_CHAIN_LDS_INTS * 4, st, v, h->ckey, h->crank);
    tb = h->temp_bytes;
    VG_CHECK(rocprim::radix_sort_pairs(h->d_temp, tb, h->ckey, h->ckey_s, h->crank, h->crank_s, (size_t)m, 0, HD_CHAIN_KEY_BITS, st));
    v.chain_key = h->ckey_s; v.chain_rank = h->crank_s;
    hipLaunchKernelGGL(k_hd_chain_stats, dim3(512), B, 0, st, v);
    if (vg_max_dynamic_lds((const void*)k_hd_tree_bc, HD_TREE_BC_LDS_INTS * 4, hd_tree_bc_lds) != VG_OK) return VG_ERR_HIP;
    hipLaunchKernelGGL(k_hd_tree_bc, dim3(1), dim3(512), HD_TREE_BC_LDS_INTS * 4, st, v, h->bfs_key, h->stamps);
    hipLaunchKernelGGL(k_hd_points, dim3(gn), B, 0, st, v);
    if (d_n_clusters) VG_CHECK(hipMemcpyAsync(d_n_clusters, v.n_clusters, sizeof(int), hipMemcpyDeviceToDevice, st));
    VG_LAUNCH_CHECK();
    return VG_OK;
}

}  // extern "C"
