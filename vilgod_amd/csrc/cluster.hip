// Spatial clustering on gfx950 (kernels K2a-K2c): the GPU half of HDBSCAN.
//
// Replaces the work hidden in `cluster_model.fit(points_ref_wo_ground)` (src/vilgod/zero_shot_detector.py:248;
// hdbscan.HDBSCAN(min_cluster_size=15, cluster_selection_epsilon=0.15), tools/configs/preprocessor/waymo.yaml:10-15):
//   K2a  k_cl_bbox / k_cl_codes / radix sort / k_cl_cell_marks + scan
//        points -> 0.4 m cells, Morton order (6 interleaved bits + 3 high bits of x,y), dense cell-start table
//        (any 2^l-aligned cube of cells is ONE contiguous range of the sorted points)
//   K2b  k_cl_core        exact k-NN core distance: pruned depth-first walk of the implicit octree (nearest child
//                         first, box-distance pruning against the current k-th distance, nodes with <= 48 points
//                         scanned directly), float64 distances, 16-entry register-resident sorted list
//   K2c  k_cl_b_*         exact minimum spanning tree of the mutual-reachability graph, Boruvka rounds with the
//                         same walk ("nearest point of another component"): subtrees owned entirely by the
//                         query's component are skipped through per-round purity tables (levels 0..3), and
//                         per-component upper bounds published with atomicMin prune the rest of the component;
//                         edges are ordered by the STRICT total order
//                         (w2, pair d2, min id, max id) so the tree is unique and equals the oracle's
//        radix sort of the n-1 edges by weight
// The sequential hierarchy stage (K2d) is csrc/hdbscan_tree.cpp on the host.
//
// All distances are float64 with d2 = (dx*dx + dy*dy) + dz*dz on exactly converted float32 coordinates
// (what the library computes after its float64 conversion); compiled with -ffp-contract=off.
#include <string.h>
#include <vector>
#include <algorithm>
#include <cstring>
#include <math.h>
#include <algorithm>
#include "common.h"
#include "vilgod_hip.h"
#include <rocprim/rocprim.hpp>

#define CL_CELL 0.4
#define CL_NX 512
#define CL_NY 512
#define CL_NZ 64
#define CL_LB 6                       // interleaved bits per axis
#define CL_NCODES (1 << 24)           // 9 + 9 + 6 bits
#define CL_LMAX 6                     // coarsest Morton-contiguous level (25.6 m cubes): 8 x 8 x 1 roots
#define CL_PUR_LEVELS 4               // purity tables for levels 0..3 (0.4 .. 3.2 m); levels 4..6 were measured: no gain
#ifndef CL_LEAF
#define CL_LEAF 48                    // nodes with at most this many points are scanned instead of subdivided
#endif
#define CL_FIRST_BATCH 6      // Boruvka rounds queued before the first host read of the edge counter (development build: VG_CLUSTER_FIRST_BATCH)
#define CL_STACK 44      // DFS stack entries per thread (LDS): the walk never holds more than 1 + 7 * CL_LMAX = 43 nodes (a pop precedes every push of
                         // <= 8 children, level-0 nodes are never expanded); 44 KB per 256-thread block -> three blocks per CU instead of two
#define CL_K 16                       // neighbours kept (k-th other point = entry k, entry 0 is the point itself)

struct ClGrid {
    double ox, oy, oz;                // world coordinate of cell (0,0,0)'s min corner
    unsigned int kmin[3], kmax[3];    // ordered-int bbox keys (scratch for the reduction)
    double inf;                       // +infinity, set by the host.  Kernels that keep a wave-uniform float64 in SGPRs read it from
                                      // here: ROCm 7.2's gfx950 back end materialises a uniform `double x = INFINITY` as
                                      // `s_mov_b64 s[..], 0x7ff0000000000000`, which gfx9 cannot encode (32-bit literals only; the
                                      // assembler rejects the line, the direct object emission truncates it to 0.0)
};

struct vg_cluster {
    int max_points;
    ClGrid* d_grid;
    unsigned int *d_code, *d_code_s;      // Morton codes (unsorted / sorted)
    int *d_perm_in, *d_perm;              // identity / sorted -> original index
    float4* d_spts;                       // sorted points (x,y,z,e): e = 4th clustering coordinate (0 in 3-D)
    float* d_st;                          // sorted 5th clustering coordinate (5-D only)
    int grid_n;                           // points in the grid currently built (vg_cluster_grid / vg_cluster_mst_nd)
    int* d_cell_start;                    // [CL_NCODES+1] reversed min-scan layout, see cl_cell_start()
    int* d_cell_comp;                     // purity tables, levels 0..3 back to back: component id if pure, else -1
    unsigned int* d_cell_e;               // same layout: (min, max) of the 4th coordinate in the node as two fp16 (dim >= 4)
    double* d_core2;                      // [n] sorted order
    int *d_comp, *d_parent, *d_parent2;   // Boruvka components (sorted index space)
    unsigned long long *d_best_w, *d_best_e;
    int *d_sel_a, *d_sel_b;
    unsigned long long *d_pt_w, *d_pt_key, *d_pt_d, *d_best_d;   // pt_d / best_d: squared pair distance (tie-break after w)
    int4* d_aux;                          // per sorted point {core2 (2 words), component, 5th coordinate}: what a leaf scan needs next to spts
    double* d_pt_lb;                      // per point: a lower bound of the weight of ANY edge leaving its component (grows over the rounds)
    int* d_pt_b;
    int* d_counter;                       // [0] number of MST edges emitted
    int *d_mst_a, *d_mst_b;               // original ids
    unsigned long long *d_mst_w, *d_mst_w_s;
    int *d_mst_idx, *d_mst_idx_s;
    void* d_temp;
    size_t temp_bytes;
    int* h_counter;                       // pinned
    int* d_dbg;                           // VG_CLUSTER_DEBUG=1: points scanned per thread in the last search round
    unsigned int* d_entries;
    int* d_csize;                          // points per component, kept at the component's root (sorted index space)
    unsigned long long* d_giant;           // this round's largest component: size << 32 | root (0 until the first round's kernels ran)
    int *d_far, *d_far_flag;  // cooperative kernels: (node, 64-query) work list of the frame; queries handed on to the next phase
};

// Consecutive hardware workgroup ids are dealt round-robin to the 8 XCDs (observed placement, speed only): logical block = a contiguous run
// of the Morton order per XCD, so that the table lines and points a region's queries share are served by ONE L2 instead of eight
// (round 5: Sigma k_cl_b_search 2 354 -> 2 280 us per MST, two interleaved pairs of traces).
__device__ __forceinline__ int cl_xcd_block(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, s = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + s;
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int cl_spread6(unsigned int v) {   // 6 bits -> every third bit
    v &= 63u;
    v = (v | (v << 8)) & 0x300Fu;
    v = (v | (v << 4)) & 0x30C3u;
    v = (v | (v << 2)) & 0x9249u;
    return v;
}
__device__ __forceinline__ unsigned int cl_code(int cx, int cy, int cz) {
    unsigned int lo = cl_spread6(cx) | (cl_spread6(cy) << 1) | (cl_spread6(cz) << 2);
    unsigned int hi = ((unsigned int)cx >> 6) | (((unsigned int)cy >> 6) << 3);
    return (hi << 18) | lo;
}
__device__ __forceinline__ void cl_cell_of(const ClGrid& g, double x, double y, double z, int& cx, int& cy, int& cz) {
    cx = (int)floor((x - g.ox) * (1.0 / CL_CELL));
    cy = (int)floor((y - g.oy) * (1.0 / CL_CELL));
    cz = (int)floor((z - g.oz) * (1.0 / CL_CELL));
    cx = cx < 0 ? 0 : (cx > CL_NX - 1 ? CL_NX - 1 : cx);
    cy = cy < 0 ? 0 : (cy > CL_NY - 1 ? CL_NY - 1 : cy);
    cz = cz < 0 ? 0 : (cz > CL_NZ - 1 ? CL_NZ - 1 : cz);
}
// cell_start is stored reversed (index NCODES - code) so that the "first point with code >= c" table is a
// forward inclusive min-scan.
__device__ __forceinline__ int cl_start(const int* __restrict__ cs, unsigned int code) { return cs[CL_NCODES - code]; }
// The same table per coarser level l = 1 .. CL_LMAX, stored FORWARD and compact right behind the cell table: entry c of level l =
// first sorted point whose code >> 3l is >= c (c = 0 .. NCODES >> 3l, the last one = n).  The eight children of a node (nine
// consecutive entries: they include the node's own range) and the 64 grand-children of a block are then neighbours in memory,
// where the cell table has them 4 * 8^l bytes apart (k_cl_levels fills them, 16 us per frame).
__device__ __host__ __forceinline__ size_t cl_lvl_off(int l) {      // offset of level l (>= 1) behind the cell table
    return ((size_t)CL_NCODES - ((size_t)CL_NCODES >> (3 * (l - 1)))) / 7 + (size_t)(l - 1);
}
#define CL_LVL_TOTAL (cl_lvl_off(CL_LMAX + 1))
// first sorted point of node c (= code >> 3l) of level l; c + 1 gives the node's end
__device__ __forceinline__ int cl_start_l(const int* __restrict__ cs, int l, unsigned int c) {
    return l == 0 ? cs[CL_NCODES - c] : cs[(size_t)(CL_NCODES + 1) + cl_lvl_off(l) + c];
}
__global__ void k_cl_levels(int* __restrict__ cs) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= CL_LVL_TOTAL) return;
    int l = 1;
    while (l < CL_LMAX && idx >= cl_lvl_off(l + 1)) ++l;
    const size_t c = idx - cl_lvl_off(l);
    cs[(size_t)(CL_NCODES + 1) + idx] = cs[(size_t)CL_NCODES - (c << (3 * l))];
}

__global__ void k_cl_bbox(const float* __restrict__ pts, int n, int stride, ClGrid* g) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float v = pts[(size_t)i * stride + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        mn[a] = vg_wave_min(mn[a]);
        mx[a] = vg_wave_max(mx[a]);
    }
    // one set of atomics per BLOCK (64 blocks): 7.4 k atomics of 1 236 waves on the same cache line took 87 us
    __shared__ float red[4][6];
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[threadIdx.x >> 6][a] = mn[a]; red[threadIdx.x >> 6][3 + a] = mx[a]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        float lo = red[0][a], hi = red[0][3 + a];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { lo = fminf(lo, red[w][a]); hi = fmaxf(hi, red[w][3 + a]); }
        atomicMin(&g->kmin[a], vg_fkey(lo));
        atomicMax(&g->kmax[a], vg_fkey(hi));
    }
}

__global__ void k_cl_grid(ClGrid* g) {
    if (threadIdx.x != 0) return;
    double lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = vg_fkey_inv(g->kmin[a]);
        hi[a] = vg_fkey_inv(g->kmax[a]);
    }
    const double ext[3] = {CL_NX * CL_CELL, CL_NY * CL_CELL, CL_NZ * CL_CELL};
    double o[3];
    for (int a = 0; a < 3; ++a) {
        // centre the data in the grid when it fits, otherwise anchor at the minimum (outliers clamp to border cells)
        double span = hi[a] - lo[a];
        double start = span < ext[a] ? lo[a] - 0.5 * (ext[a] - span) : lo[a];
        o[a] = floor(start / CL_CELL) * CL_CELL;
    }
    g->ox = o[0]; g->oy = o[1]; g->oz = o[2];
}

__global__ void k_cl_codes(const float* __restrict__ pts, int n, int stride, const ClGrid* __restrict__ g,
                           unsigned int* __restrict__ code, int* __restrict__ perm) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int cx, cy, cz;
    cl_cell_of(*g, pts[(size_t)i * stride], pts[(size_t)i * stride + 1], pts[(size_t)i * stride + 2], cx, cy, cz);
    code[i] = cl_code(cx, cy, cz);
    perm[i] = i;
}

__global__ void k_cl_gather(const float* __restrict__ pts, int n, int stride, int dim, const int* __restrict__ perm,
                            const unsigned int* __restrict__ code_s, float4* __restrict__ spts, float* __restrict__ st,
                            int* __restrict__ cell_start_rev) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = pts + (size_t)perm[i] * stride;
    spts[i] = make_float4(p[0], p[1], p[2], dim >= 4 ? p[3] : 0.f);
    if (dim >= 5) st[i] = p[4];
    if (i == 0 || code_s[i] != code_s[i - 1]) cell_start_rev[CL_NCODES - code_s[i]] = i;
}

__global__ void k_cl_fill_int(int* p, int v, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---------------------------------------------------------------------------------------------
// squared distance in float64, summed left to right over the DIM coordinates: ((((dx2 + dy2) + dz2) + de2) + dt2)
template <int DIM>
__device__ __forceinline__ double cl_d2(double ax, double ay, double az, double ae, double at, const float4& b,
                                        const float* __restrict__ st, int j) {
    double dx = ax - (double)b.x, dy = ay - (double)b.y, dz = az - (double)b.z;
    double d2 = (dx * dx + dy * dy) + dz * dz;
    if (DIM >= 4) { const double de = ae - (double)b.w; d2 = d2 + de * de; }
    if (DIM >= 5) { const double dt = at - (double)st[j]; d2 = d2 + dt * dt; }
    return d2;
}

// squared distance from q to the nearest face of the level-l block [b-1, b+2) (faces clipped by the grid do not count)
__device__ __forceinline__ double cl_block_radius2(const ClGrid& g, double qx, double qy, double qz, int bx, int by,
                                                   int bz, int l) {
    const double s = CL_CELL * (double)(1 << l);
    double r = INFINITY;
    const int nb[3] = {CL_NX >> l, CL_NY >> l, CL_NZ >> l};
    const int b[3] = {bx, by, bz};
    const double q[3] = {qx, qy, qz}, o[3] = {g.ox, g.oy, g.oz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // a face only bounds the search if cells exist beyond it (points outside the grid are clamped INTO border cells)
        if (b[a] - 2 >= 0) r = fmin(r, q[a] - (o[a] + (double)(b[a] - 1) * s));
        if (b[a] + 2 < nb[a]) r = fmin(r, (o[a] + (double)(b[a] + 2) * s) - q[a]);
    }
    if (r < 0) r = 0;
    return isinf(r) ? INFINITY : r * r;
}

// ---- pruned depth-first traversal of the implicit octree ---------------------------------------------------
// A node = (level l, cell coordinates at that level); its points are the contiguous range
// [start(code), start(code + 8^l)) of the Morton-sorted array.  Stack entries pack (l, x, y, z) in 32 bits and live
// in LDS, interleaved by thread (bank-conflict free).
__device__ __forceinline__ unsigned int cl_pack(int l, int x, int y, int z) {
    return ((unsigned int)l << 27) | ((unsigned int)x << 18) | ((unsigned int)y << 9) | (unsigned int)z;
}
__device__ __forceinline__ void cl_unpack(unsigned int e, int& l, int& x, int& y, int& z) {
    l = (int)(e >> 27); x = (int)((e >> 18) & 511u); y = (int)((e >> 9) & 511u); z = (int)(e & 511u);
}
// squared distance from q to the axis-aligned box of node (l,x,y,z); 0 inside.  Border nodes extend to infinity on
// the outside (points beyond the grid are clamped INTO border cells).
__device__ __forceinline__ double cl_box_d2(const ClGrid& g, double qx, double qy, double qz, int l, int x, int y, int z) {
    const double s = CL_CELL * (double)(1 << l);
    const int nb[3] = {CL_NX >> l, CL_NY >> l, CL_NZ >> l};
    const int c[3] = {x, y, z};
    const double q[3] = {qx, qy, qz}, o[3] = {g.ox, g.oy, g.oz};
    double d2 = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double lo = o[a] + (double)c[a] * s, hi = lo + s;
        double d = 0.0;
        if (q[a] < lo && c[a] > 0) d = lo - q[a];
        else if (q[a] > hi && c[a] < nb[a] - 1) d = q[a] - hi;
        d2 += d * d;
    }
    return d2;
}

__device__ __forceinline__ size_t cl_pur_off_fwd(int l) {   // offset of level l inside the per-node tables (= cl_pur_off)
    size_t o = 0;
    for (int i = 0; i < l; ++i) o += (size_t)CL_NCODES >> (3 * i);
    return o;
}
// ---- range of the 4th clustering coordinate per octree node (levels 0..CL_PUR_LEVELS-1), for the 4-/5-D searches -----
// The grid prunes on x,y,z only; in the two-frame input the 4th coordinate (entropy score, 0..1) often separates a
// point from ALL its spatial neighbours, so the xyz bound alone lets the search wander through thousands of points that
// are close in space and far in 5-D.  Packed as two fp16: min rounded DOWN (low half), max rounded UP (high half).
__device__ __forceinline__ unsigned int cl_pack_e(float mn, float mx) {
    __half hmn = __float2half_rd(mn), hmx = __float2half_ru(mx);
    return (unsigned int)__half_as_ushort(hmn) | ((unsigned int)__half_as_ushort(hmx) << 16);
}
__device__ __forceinline__ void cl_unpack_e(unsigned int v, float& mn, float& mx) {
    mn = __half2float(__ushort_as_half((unsigned short)(v & 0xFFFFu)));
    mx = __half2float(__ushort_as_half((unsigned short)(v >> 16)));
}
// squared gap between qe and the node's [min, max] range (0 inside), shaved by 2^-30 so that rounding cannot lift the
// sum above the true minimum
__device__ __forceinline__ double cl_e_gap2(unsigned int packed, double qe) {
    float mn, mx;
    cl_unpack_e(packed, mn, mx);
    double d = 0.0;
    if (qe < (double)mn) d = (double)mn - qe;
    else if (qe > (double)mx) d = qe - (double)mx;
    d *= (1.0 - 9.313225746154785e-10);
    return d * d;
}
__global__ void k_cl_e_leaf(int n, const unsigned int* __restrict__ code_s, const float4* __restrict__ spts,
                            unsigned int* __restrict__ cell_e) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned int c = code_s[i];
    if (i > 0 && code_s[i - 1] == c) return;
    float mn = spts[i].w, mx = mn;
    for (int j = i + 1; j < n && code_s[j] == c; ++j) {
        const float e = spts[j].w;
        mn = fminf(mn, e);
        mx = fmaxf(mx, e);
    }
    cell_e[c] = cl_pack_e(mn, mx);
}
__global__ void k_cl_e_up(int n, int l, const unsigned int* __restrict__ code_s, const int* __restrict__ cs,
                          unsigned int* __restrict__ cell_e) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned int key = code_s[i] >> (3 * l);
    if (i > 0 && (code_s[i - 1] >> (3 * l)) == key) return;
    const unsigned int* below = cell_e + cl_pur_off_fwd(l - 1);
    float mn = INFINITY, mx = -INFINITY;
    for (unsigned int ch = 0; ch < 8; ++ch) {
        const unsigned int ck = key * 8 + ch;
        const unsigned int c0 = ck << (3 * (l - 1));
        if (cl_start_l(cs, l - 1, c0 >> (3 * (l - 1))) == cl_start_l(cs, l - 1, (c0 >> (3 * (l - 1))) + 1u)) continue;   // empty child: entry not written
        float a, b;
        cl_unpack_e(below[ck], a, b);
        mn = fminf(mn, a);
        mx = fmaxf(mx, b);
    }
    cell_e[cl_pur_off_fwd(l) + key] = cl_pack_e(mn, mx);
}

#ifdef VG_DEV      // the per-point tree walk of rounds 1-2 (superseded by the cooperative kernels below in round 3): A/B aid, development build only
template <int DIM>
__global__ __launch_bounds__(256) void k_cl_core(const float4* __restrict__ spts, const float* __restrict__ stt, int n,
                                                 const ClGrid* __restrict__ gp, const int* __restrict__ cs,
                                                 const unsigned int* __restrict__ cell_e, int k,
                                                 double* __restrict__ core2, int* __restrict__ dbg_scan) {
    __shared__ unsigned int stack[CL_STACK * 256];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int scanned = 0;                                  // VG_CLUSTER_DEBUG: pair distances evaluated by this point
    unsigned int* st = stack + threadIdx.x;
    const ClGrid g = *gp;
    const float4 qf = spts[i];
    const double qx = qf.x, qy = qf.y, qz = qf.z, qe = qf.w, qt = DIM >= 5 ? (double)stt[i] : 0.0;
    int cx, cy, cz;
    cl_cell_of(g, qx, qy, qz, cx, cy, cz);
    double h[CL_K];
#pragma unroll
    for (int j = 0; j < CL_K; ++j) h[j] = INFINITY;
    const int rx0 = cx >> CL_LMAX, ry0 = cy >> CL_LMAX;
    const int nrx = CL_NX >> CL_LMAX, nry = CL_NY >> CL_LMAX;
    for (int rr = 0; rr < nrx * nry; ++rr) {
        // own root first, then the others (pruned by distance)
        int rx = rr % nrx, ry = rr / nrx;
        if (rr == 0) { rx = rx0; ry = ry0; }
        else if (rx == rx0 && ry == ry0) { rx = 0; ry = 0; }
        int sp = 0;
        st[0] = cl_pack(CL_LMAX, rx, ry, 0);
        sp = 1;
        while (sp > 0) {
            int l, x, y, z;
            cl_unpack(st[(--sp) * 256], l, x, y, z);
            const double nb2 = cl_box_d2(g, qx, qy, qz, l, x, y, z);
            if (nb2 >= h[k]) continue;                                       // cannot lower the k-th distance (ALU only)
            const unsigned int c0 = cl_code(x << l, y << l, z << l);
            const int j0 = cl_start_l(cs, l, c0 >> (3 * l)), j1 = cl_start_l(cs, l, (c0 >> (3 * l)) + 1u);
            if (j0 == j1) continue;
            if (DIM >= 4 && l < CL_PUR_LEVELS && nb2 + cl_e_gap2(cell_e[cl_pur_off_fwd(l) + (c0 >> (3 * l))], qe) >= h[k]) continue;
            if (l == 0 || j1 - j0 <= CL_LEAF) {
                scanned += j1 - j0;
                for (int jb = j0; jb < j1; jb += 4) {                       // four independent loads per trip
                    float4 pj[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) pj[u] = spts[jb + u < j1 ? jb + u : j1 - 1];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int j = jb + v;
                        if (j >= j1) continue;
                        double d2 = cl_d2<DIM>(qx, qy, qz, qe, qt, pj[v], stt, j);
                        if (d2 < h[CL_K - 1]) {
#pragma unroll
                            for (int u = 0; u < CL_K; ++u)
                                if (d2 < h[u]) { double tmp = h[u]; h[u] = d2; d2 = tmp; }
                        }
                    }
                }
                continue;
            }
            // children, nearest octant LAST on the stack -> popped first
            const int l1 = l - 1;
            const int ox = ((cx >> l1) > 2 * x) ? 1 : 0, oy = ((cy >> l1) > 2 * y) ? 1 : 0, oz = ((cz >> l1) > 2 * z) ? 1 : 0;
            const int near = ox | (oy << 1) | (oz << 2);
            // the 8 children are consecutive Morton ranges: their 9 boundaries are loaded together (independent loads), so
            // empty octants -- most of them in LiDAR data -- are never pushed and never cost a pop with two dependent loads
            int bnd[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) bnd[c] = cl_start_l(cs, l1, (c0 >> (3 * l1)) + (unsigned int)c);
            unsigned int occupied = 0;                    // bit c: child c holds points (static indices only: no scratch)
#pragma unroll
            for (int c = 0; c < 8; ++c) occupied |= (bnd[c] != bnd[c + 1]) ? (1u << c) : 0u;
            for (int c = 7; c >= 0; --c) {
                const int ch = c ^ near;
                if (!((occupied >> ch) & 1u)) continue;
                const int nx = 2 * x + (ch & 1), ny = 2 * y + ((ch >> 1) & 1), nz = 2 * z + ((ch >> 2) & 1);
                if (cl_box_d2(g, qx, qy, qz, l1, nx, ny, nz) >= h[k]) continue;
                if (sp < CL_STACK) st[(sp++) * 256] = cl_pack(l1, nx, ny, nz);
            }
        }
    }
    core2[i] = h[k];
    if (dbg_scan) dbg_scan[i] = scanned;
}
#endif  // VG_DEV

// ---------------------------------------------------------------------------------------------
// Cooperative exact k-NN core distances (north_star: "LDS-staged radius-neighbour / HDBSCAN core-distance kernel over
// Morton-sorted points with coalesced HBM reads").  Two phases, both exact, both bit-identical to the walk above (the same
// float64 pair distances; the k-th smallest value does not depend on the order in which the pairs are met):
//   A  k_cl_core_blk   one WAVE per (0.8 m block, 64 queries): the queries of a level-1 node are 64 consecutive sorted points,
//                      lane = query.  The 27 neighbouring level-1 nodes are 27 contiguous ranges of the sorted array: they are
//                      read with coalesced loads (one point per lane), staged in LDS as float64 and scanned by every lane
//                      (broadcast reads) into its register top-16 (min / max chain, entered only when some lane improves).  A
//                      neighbour node is skipped when its box is at least as far as every lane's current k-th distance.  A
//                      lane is DONE when its k-th distance does not exceed the distance to the shell's outer faces (nothing
//                      outside the 2.4 m cube can be closer); the others -- isolated points, ~8 % of a LiDAR frame -- go on a
//                      list with their k-th distance so far as an upper bound.
//   B  k_cl_core_far   one WAVE per listed query, lane = candidate: the level-L 3x3x3 shell that covers the bound (L = 2..6) is
//                      expanded to its 27 x 64 level-(L-2) sub-nodes lane-parallel (range + box distance per lane), the
//                      sub-nodes nearer than the running k-th distance are streamed 64 points at a time (coalesced), and the
//                      wave keeps ONE sorted list of the smallest distances across lanes 0..15 (insert = two DPP shifts, max,
//                      min).  No per-lane dependent walk: the slowest lane of the old kernel (an isolated point opening ~100
//                      nodes, each behind two dependent reads of the 64 MB cell table) no longer sets the launch time.
__constant__ signed char CLB_ORDER[27][3] = {
    {0, 0, 0},
    {-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1},
    {-1, -1, 0}, {1, -1, 0}, {-1, 1, 0}, {1, 1, 0}, {-1, 0, -1}, {1, 0, -1}, {-1, 0, 1}, {1, 0, 1}, {0, -1, -1}, {0, 1, -1}, {0, -1, 1}, {0, 1, 1},
    {-1, -1, -1}, {1, -1, -1}, {-1, 1, -1}, {1, 1, -1}, {-1, -1, 1}, {1, -1, 1}, {-1, 1, 1}, {1, 1, 1}};
#define CLB_TILE 128                  // candidates staged per pass (two coalesced loads per lane)

// (h, v) <- (min(h, v), max(h, v)) for non-NaN doubles
__device__ __forceinline__ void cl_minmax(double& h, double& v) {
    double lo, hi;
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(h), "v"(v));
    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(h), "v"(v));
    h = lo;
    v = hi;
}

// Work list of the cooperative kernels.  A work item = up to 64 consecutive sorted points = the points of ONE octree node,
// chosen top-down from level 3 (3.2 m): a node that holds at most 64 points is one item (lane utilisation: a LiDAR frame's
// 0.8 m nodes hold 9 points on average), a fuller node is split into its children, level-0 cells (0.4 m) are cut into chunks of
// 64.  The item's candidates are the 27 nodes around it AT ITS LEVEL, so sparse regions get a wide shell (more queries end in
// the cooperative phase) and dense surfaces near the sensor a narrow one (a 0.8 m node there holds up to ~600 points with
// ~4 000 in its shell; the heaviest item sets the launch time).  entry = first query index | level << 30.
#define CLB_TOP 1                     // coarsest work-item level (levels 2 / 3 were measured: their shells next to dense objects hold
                                      // 10-20 k candidates and the heaviest item sets the launch time: 832 us instead of 267)
#define CLB_SPLIT 96                  // a level-1 node with more points than this is handled cell by cell
#define CLB_SHELL_CAP 4096            // ... and so is one whose level + 2 ancestor (a 4 x 4 x 4 block that holds most of its shell) is crowded
__global__ void k_cl_blocks(int n, const unsigned int* __restrict__ code_s, const int* __restrict__ cs,
                            unsigned int* __restrict__ entries, int* __restrict__ counters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned int code = code_s[i], prev = i > 0 ? code_s[i - 1] : ~code;
    if (prev == code) return;                             // not the first point of any node
    // the node that holds point i at every level up to CLB_TOP + 2: its range (independent reads of the cell table)
    int j0[CLB_TOP + 3], j1[CLB_TOP + 3];
#pragma unroll
    for (int lv = 0; lv <= CLB_TOP + 2; ++lv) {
        const unsigned int kk = code >> (3 * lv);
        j0[lv] = cl_start_l(cs, lv, kk);
        j1[lv] = cl_start_l(cs, lv, kk + 1u);
    }
    // a node "fits" when it fills at most one wave and its surroundings are not crowded (the item's cost is its shell's
    // population); the work item of point i's branch is the coarsest node that fits (cells always do, in chunks of 64)
    bool fits[CLB_TOP + 2];
    fits[0] = true;
#pragma unroll
    for (int lv = 1; lv <= CLB_TOP; ++lv) fits[lv] = (j1[lv] - j0[lv]) <= CLB_SPLIT && (j1[lv + 2] - j0[lv + 2]) <= CLB_SHELL_CAP;
    fits[CLB_TOP + 1] = false;
#pragma unroll
    for (int lv = 0; lv <= CLB_TOP; ++lv) {
        if (j0[lv] != i) break;                           // i does not start this node, nor any coarser one
        if (!fits[lv] || fits[lv + 1]) continue;
        const int cnt = j1[lv] - j0[lv];
        const int chunks = (cnt + 63) >> 6;
        const int base = atomicAdd(&counters[2], chunks);
        for (int c = 0; c < chunks; ++c) entries[base + c] = (unsigned int)(i + (c << 6)) | ((unsigned int)lv << 30);
    }
}

// Staging plan of one pass of a cooperative kernel: the candidates of up to 27 neighbour nodes, CONCATENATED, so that one
// pass (one round trip to memory for the whole wave) covers a whole neighbourhood when it holds <= CLB_TILE points -- most of
// them do.  seg[t] = {first sorted index, points taken, offset in the tile}; filled by wave-uniform code.
struct ClbPlan {
    int start[27], take[27], off[28];
};
// sorted index of the tile's f-th candidate
__device__ __forceinline__ int clb_locate(const ClbPlan& p, int nseg, int f) {
    int t = 0;
    for (int u = 1; u < nseg; ++u) t += (f >= p.off[u]) ? 1 : 0;
    return p.start[t] + (f - p.off[t]);
}

template <int DIM>
__global__ __launch_bounds__(64) void k_cl_core_blk(const float4* __restrict__ spts, const float* __restrict__ stt, int n,
                                                    const ClGrid* __restrict__ gp, const int* __restrict__ cs,
                                                    const unsigned int* __restrict__ code_s, const unsigned int* __restrict__ entries,
                                                    int* __restrict__ counters, int k, double* __restrict__ core2,
                                                    int* __restrict__ far_list, int* __restrict__ dbg_scan) {
    __shared__ int bnd[27][2];
    __shared__ ClbPlan plan;
    __shared__ float4 tile[CLB_TILE];                 // x, y, z, 4th coordinate
    __shared__ float tile_t[DIM >= 5 ? CLB_TILE : 1]; // 5th coordinate
    const int lane = threadIdx.x;
    const ClGrid g = *gp;
    const int n_entries = counters[2];
    for (int e = blockIdx.x; e < n_entries; e += gridDim.x) {
        const int lv = (int)(entries[e] >> 30), i0 = (int)(entries[e] & 0x3FFFFFFFu);   // node level (0 .. 3), first query
        const unsigned int key = code_s[i0] >> (3 * lv);
        const int iend = min(cl_start_l(cs, lv, key + 1u), i0 + 64);
        const int i = i0 + lane;
        const bool active = i < iend;
        const float4 qf = spts[active ? i : i0];
        const float qtf = DIM >= 5 ? stt[active ? i : i0] : 0.f;
        const double qx = qf.x, qy = qf.y, qz = qf.z, qe = qf.w, qt = qtf;
        int bx, by, bz;                                   // the node's coordinates at its level (wave-uniform: from its first point)
        {
            const float4 f0 = spts[i0];
            cl_cell_of(g, (double)f0.x, (double)f0.y, (double)f0.z, bx, by, bz);
            bx >>= lv; by >>= lv; bz >>= lv;
        }
        __syncthreads();                                  // the previous entry's reads of bnd / plan / tile are over
        if (lane < 27) {
            const int nx = bx + CLB_ORDER[lane][0], ny = by + CLB_ORDER[lane][1], nz = bz + CLB_ORDER[lane][2];
            int j0 = 0, j1 = 0;
            if (nx >= 0 && ny >= 0 && nz >= 0 && nx < (CL_NX >> lv) && ny < (CL_NY >> lv) && nz < (CL_NZ >> lv)) {
                const unsigned int c0 = cl_code(nx << lv, ny << lv, nz << lv);
                j0 = cl_start_l(cs, lv, c0 >> (3 * lv));
                j1 = cl_start_l(cs, lv, (c0 >> (3 * lv)) + 1u);
            }
            bnd[lane][0] = j0;
            bnd[lane][1] = j1;
        }
        __syncthreads();
        double h[CL_K];
#pragma unroll
        for (int j = 0; j < CL_K; ++j) h[j] = INFINITY;
        float thr = INFINITY;                             // float32 screen: a candidate whose float32 distance exceeds it cannot enter the list
        int scanned = 0;
        int t_next = 0, cur = bnd[0][0];                  // next neighbour node to plan, next point of it
        while (t_next < 27) {
            // ---- plan one pass (wave-uniform) ----
            int nseg = 0, tot = 0;
            while (t_next < 27 && tot < CLB_TILE) {
                const int j1 = bnd[t_next][1];
                bool need = cur < j1;
                if (need) {
                    // the node's box against every lane's k-th distance so far (same rule as the walk: >= cannot lower it)
                    const double nb2 = cl_box_d2(g, qx, qy, qz, lv, bx + CLB_ORDER[t_next][0], by + CLB_ORDER[t_next][1], bz + CLB_ORDER[t_next][2]);
                    need = __any(active && nb2 < h[k]);
                }
                if (need) {
                    const int take = min(j1 - cur, CLB_TILE - tot);
                    if (lane == 0) { plan.start[nseg] = cur; plan.take[nseg] = take; plan.off[nseg] = tot; }
                    ++nseg; tot += take; cur += take;
                    if (cur < j1) break;                  // tile full: the rest of this node in the next pass
                }
                ++t_next;
                if (t_next < 27) cur = bnd[t_next][0];
            }
            if (tot == 0) break;
            __syncthreads();
            // ---- stage: coalesced within each node's range ----
#pragma unroll
            for (int r = 0; r < CLB_TILE / 64; ++r) {
                const int f = lane + 64 * r;
                if (f < tot) {
                    const int j = clb_locate(plan, nseg, f);
                    tile[f] = spts[j];
                    if (DIM >= 5) tile_t[f] = stt[j];
                }
            }
            __syncthreads();
            scanned += tot;
            // ---- scan, four candidates per trip.  Screen in float32 first: the float64 distance of the (exactly converted)
            // float32 coordinates differs from this float32 evaluation by a few ulp, the threshold carries a 1e-5 margin, so
            // a candidate the screen drops is farther than the lane's 16th distance and the exact chain never misses one ----
            for (int c0 = 0; c0 < tot; c0 += 4) {
                float4 p[4];
                float sd[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    p[v] = tile[min(c0 + v, tot - 1)];
                    const float dx = qf.x - p[v].x, dy = qf.y - p[v].y, dz = qf.z - p[v].z;
                    float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    if (DIM >= 4) { const float de = qf.w - p[v].w; d = fmaf(de, de, d); }
                    if (DIM >= 5) { const float dt = qtf - tile_t[min(c0 + v, tot - 1)]; d = fmaf(dt, dt, d); }
                    sd[v] = c0 + v < tot ? d : INFINITY;
                }
                if (!__any(fminf(fminf(sd[0], sd[1]), fminf(sd[2], sd[3])) <= thr)) continue;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    if (!__any(sd[v] <= thr)) continue;
                    const double dx = qx - (double)p[v].x, dy = qy - (double)p[v].y, dz = qz - (double)p[v].z;
                    double x = (dx * dx + dy * dy) + dz * dz;
                    if (DIM >= 4) { const double de = qe - (double)p[v].w; x = x + de * de; }
                    if (DIM >= 5) { const double dt = qt - (double)tile_t[min(c0 + v, tot - 1)]; x = x + dt * dt; }
                    if (c0 + v >= tot) x = INFINITY;
                    if (!__any(x < h[CL_K - 1])) continue;
                    // sorted insert as a min / max chain.  Raw v_min_f64 / v_max_f64: the operands are squared distances or
                    // +inf, never NaN, so the canonicalising v_max_f64 x, x that fmin / fmax put in front of every operand in
                    // IEEE mode (half of the chain's instructions) is not needed.  The lower half of the list is entered only
                    // when some lane's value belongs there.
                    if (__any(x < h[CL_K / 2 - 1])) {
#pragma unroll
                        for (int u = 0; u < CL_K / 2; ++u) cl_minmax(h[u], x);
                    }
#pragma unroll
                    for (int u = CL_K / 2; u < CL_K; ++u) cl_minmax(h[u], x);
                    thr = (float)h[CL_K - 1] * 1.00001f + 1e-30f;       // (float)(+inf) stays +inf
                }
            }
        }
        // nothing outside the 3 x 3 x 3 shell is nearer than its outer faces
        const double r2 = cl_block_radius2(g, qx, qy, qz, bx, by, bz, lv);
        const bool done = h[k] <= r2;
        if (active) {
            core2[i] = h[k];                              // final, or an upper bound for phase B
            if (dbg_scan) dbg_scan[i] = scanned;
        }
        const unsigned long long far = __ballot(active && !done);
        if (far) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&counters[3], __popcll(far));
            base = __shfl(base, 0);
            if (active && !done) far_list[base + __popcll(far & ((1ull << lane) - 1ull))] = i;
        }
    }
}

// value of lane l - 1 within the 16-lane row (0 for the row's first lane): the neighbour a sorted insert needs
__device__ __forceinline__ double cl_row_shr1(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xFFFFFFFFll), 0x111, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x111, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double cl_readlane_d(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), l);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <int DIM>
__global__ __launch_bounds__(64) void k_cl_core_far(const float4* __restrict__ spts, const float* __restrict__ stt, int n,
                                                    const ClGrid* __restrict__ gp, const int* __restrict__ cs,
                                                    const int* __restrict__ far_list, const int* __restrict__ counters, int k,
                                                    double* __restrict__ core2, int* __restrict__ dbg_scan, int force_level) {
    const int lane = threadIdx.x;
    const ClGrid g = *gp;
    const int n_far = counters[3];
    for (int u = blockIdx.x; u < n_far; u += gridDim.x) {
        const int i = far_list[u];
        const float4 qf = spts[i];
        const double qx = qf.x, qy = qf.y, qz = qf.z, qe = qf.w, qt = DIM >= 5 ? (double)stt[i] : 0.0;
        int cx, cy, cz;
        cl_cell_of(g, qx, qy, qz, cx, cy, cz);
        const double bound = core2[i];                    // phase A's k-th distance: the true one is not larger
        int L = 2;
        while (L < CL_LMAX && !(bound <= cl_block_radius2(g, qx, qy, qz, cx >> L, cy >> L, cz >> L, L))) ++L;
        if (isinf(bound)) L = 3;
        if (force_level) L = force_level;
        double hs = g.inf;                                // lanes 0..15: the smallest distances so far, ascending
        double T = g.inf;                                 // = lane k's entry (wave-uniform; see ClGrid::inf)
        int scanned = 0;
        for (;;) {
            hs = g.inf; T = g.inf;
            auto stream = [&](int j0, int j1) {
#ifdef VG_DEV
                if (dbg_scan && u == 0 && lane == 0) { const int w = atomicAdd(&dbg_scan[n + 1], 2); if (w < 200) { dbg_scan[n + 100 + w] = j0; dbg_scan[n + 100 + w + 1] = j1; } }
#endif
                for (int base = j0; base < j1; base += 64) {
                    const int j = base + lane;
                    double d2 = INFINITY;
                    if (j < j1) d2 = cl_d2<DIM>(qx, qy, qz, qe, qt, spts[j], stt, j);
                    scanned += min(64, j1 - base);
                    unsigned long long pm = __ballot(d2 < T);
                    while (pm) {
                        const int b = __ffsll((long long)pm) - 1;
                        pm &= pm - 1;
                        const double v = cl_readlane_d(d2, b);
                        if (v < T) {
                            double mx, prev = cl_row_shr1(hs);
                            asm("v_max_f64 %0, %1, %2" : "=v"(mx) : "v"(prev), "v"(v));
                            asm("v_min_f64 %0, %1, %2" : "=v"(hs) : "v"(mx), "v"(hs));
                            T = cl_readlane_d(hs, k);
                        }
                    }
                }
            };
            if (L > CL_LMAX) {                            // the grid's roots did not settle it: every point (exact for any input)
                stream(0, n);
                break;
            }
            const int BX = cx >> L, BY = cy >> L, BZ = cz >> L;
            const int l2 = L - 2;
            for (int t = 0; t < 27; ++t) {
                const int nx = BX + CLB_ORDER[t][0], ny = BY + CLB_ORDER[t][1], nz = BZ + CLB_ORDER[t][2];
                if (nx < 0 || ny < 0 || nz < 0 || nx >= (CL_NX >> L) || ny >= (CL_NY >> L) || nz >= (CL_NZ >> L)) continue;
                if (cl_box_d2(g, qx, qy, qz, L, nx, ny, nz) >= T) continue;
                // the node's 64 level-(L-2) sub-nodes, one per lane (Morton order: consecutive ranges)
                const unsigned int c0 = cl_code(nx << L, ny << L, nz << L);
                const int s0 = cl_start_l(cs, l2, (c0 >> (3 * l2)) + (unsigned int)lane);
                const int s1 = cl_start_l(cs, l2, (c0 >> (3 * l2)) + (unsigned int)lane + 1u);
                const int sx = (nx << 2) | (((lane >> 3) & 1) << 1) | (lane & 1);
                const int sy = (ny << 2) | (((lane >> 4) & 1) << 1) | ((lane >> 1) & 1);
                const int sz = (nz << 2) | (((lane >> 5) & 1) << 1) | ((lane >> 2) & 1);
                const double sb2 = cl_box_d2(g, qx, qy, qz, l2, sx, sy, sz);
                unsigned long long sm = __ballot(s0 != s1 && sb2 < T);
                while (sm) {
                    const int b = __ffsll((long long)sm) - 1;
                    sm &= sm - 1;
                    if (cl_readlane_d(sb2, b) >= T) continue;
                    stream(__builtin_amdgcn_readlane(s0, b), __builtin_amdgcn_readlane(s1, b));
                }
            }
            if (T <= cl_block_radius2(g, qx, qy, qz, BX, BY, BZ, L)) break;
            ++L;
        }
        if (lane == 0) {
            core2[i] = T;
            if (dbg_scan) dbg_scan[i] += scanned;
        }
#ifdef VG_DEV
        if (dbg_scan && u == 0 && lane < 16) dbg_scan[n + 16 + lane] = __float_as_int((float)hs);
        if (dbg_scan && u == 0 && lane == 0) { dbg_scan[n + 2] = i; dbg_scan[n + 3] = L; dbg_scan[n + 4] = __float_as_int((float)T); dbg_scan[n + 5] = __float_as_int((float)bound); }
#endif
    }
}

// ---------------------------------------------------------------------------------------------
// Boruvka
#define CL_NONE 0xFFFFFFFFFFFFFFFFull   // 'no candidate' (sorts after every weight, +inf included)

__global__ void k_cl_b_init(int n, int* __restrict__ comp, int* __restrict__ counter, int* __restrict__ pt_b,
                            double* __restrict__ pt_lb, const double* __restrict__ core2, const float* __restrict__ stt,
                            int4* __restrict__ aux, int* __restrict__ csize, unsigned long long* __restrict__ giant) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *giant = 0ull;
    if (i < n) {
        comp[i] = i; pt_b[i] = -1; pt_lb[i] = 0.0; csize[i] = 1;
        const long long cb = __double_as_longlong(core2[i]);
        aux[i] = make_int4((int)(cb & 0xFFFFFFFFll), (int)(cb >> 32), i, stt ? __float_as_int(stt[i]) : 0);
    }
    if (i == 0) counter[1] = 0;
    if (i == 0) counter[0] = 0;
}

// A point's candidate of the previous round stays its minimum foreign edge as long as the other end is still foreign
// (components only grow): keep it, and publish its weight as the component's bound BEFORE the searches start, so the
// points that do have to search again (their candidate was absorbed) prune against a tight bound from the first node on.
// Without this every late round -- few, large components -- started all its traversals unbounded (3-4 ms per round).
__global__ void k_cl_b_seed(int n, const int* __restrict__ comp, int* __restrict__ pt_b,
                            const unsigned long long* __restrict__ pt_w, unsigned long long* __restrict__ best_w, const int* __restrict__ flags) {
    if (flags[1]) return;                 // the tree was complete before this round: queued ahead without a host read
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    const int b = pt_b[a];
    if (b < 0) return;
    const int c = comp[a];
    if (comp[b] == c) pt_b[a] = -1;                 // absorbed: search again
    else atomicMin(&best_w[c], pt_w[a]);
}

// THE LARGEST COMPONENT DOES NOT SEARCH (round 5).  Boruvka needs, per round, the minimum edge leaving a component only from the components
// that are going to merge along it; any subset of the components may sit a round out -- every edge the others pick is still the minimum
// edge leaving its component, hence an edge of the (unique, strict order) MST, and as long as one other component exists the round makes
// progress.  The component that sits out is the largest one (if it holds at least n / 8 points): in the late rounds it owns most of the
// frame's points, its nearest foreign structure is metres away, and its boundary points were the longest walks of the launch -- while the
// edge that joins it to a neighbour is found from the neighbour's side anyway.  Sizes live at the roots (k_cl_b_init: 1; k_cl_b_compress adds
// an absorbed root's size to its new root); the maximum over the roots is taken here, one atomic per workgroup.
__global__ void k_cl_b_round_init(int n, const int* __restrict__ comp, unsigned long long* __restrict__ best_w,
                                  unsigned long long* __restrict__ best_d, unsigned long long* __restrict__ best_e,
                                  int* __restrict__ sel_a, const int* __restrict__ flags, const int* __restrict__ csize,
                                  unsigned long long* __restrict__ giant, const unsigned int* __restrict__ code_s,
                                  int* __restrict__ cell_comp) {
    if (flags[1]) return;                 // the tree was complete before this round: queued ahead without a host read
    __shared__ unsigned long long blk_max[4];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) const_cast<int*>(flags)[4] = 0;   // queries the cooperative search hands on to the walk this round
    unsigned long long mine = 0ull;
    if (i < n) {
        best_w[i] = CL_NONE;
        best_d[i] = ~0ull;
        best_e[i] = ~0ull;
        sel_a[i] = -1;
        if (comp[i] == i) mine = ((unsigned long long)(unsigned int)csize[i] << 32) | (unsigned int)i;
        // purity tables, first half: the first point of every node of levels 0 .. 3 writes its component; k_cl_b_purity (a later launch)
        // overwrites the entry with -1 where two neighbours inside the node disagree
        {
            const unsigned int code = code_s[i], prev = i > 0 ? code_s[i - 1] : ~code;
            size_t off = 0;
#pragma unroll
            for (int l = 0; l < CL_PUR_LEVELS; ++l) {
                if (i == 0 || (prev >> (3 * l)) != (code >> (3 * l))) cell_comp[off + (code >> (3 * l))] = comp[i];
                off += (size_t)CL_NCODES >> (3 * l);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(mine, o);
        mine = other > mine ? other : mine;
    }
    if ((threadIdx.x & 63) == 0) blk_max[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long m = blk_max[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = blk_max[w] > m ? blk_max[w] : m;
        if (m) atomicMax(giant, m);
    }
}
// does component c sit this round out?
__device__ __forceinline__ bool cl_sits_out(unsigned long long giant, int c, int min_size) {
    return (int)(giant >> 32) >= min_size && (int)(giant & 0xFFFFFFFFull) == c;
}

__device__ __forceinline__ size_t cl_pur_off(int l) {   // offset of level l inside the purity tables
    size_t o = 0;
    for (int i = 0; i < l; ++i) o += (size_t)CL_NCODES >> (3 * i);
    return o;
}

// Purity tables, levels 0 .. 3: entry = the component that owns every point of the node, or -1.  All points of a node agree iff every two
// neighbours in the sorted order agree: k_cl_b_round_init wrote the first point's component at every level, a thread whose point differs
// from its predecessor writes -1 at every level at which the two share a node (same value from every writer).  (Rounds 1-4: the first point
// of a cell walked the whole cell -- 29.5 us per launch, 80 at worst -- and three more launches built levels 1 .. 3 from the level below:
// 0.33 ms and 28 launches per frame; now one compare per thread and one launch per round.)
__global__ void k_cl_b_purity(int n, const unsigned int* __restrict__ code_s, const int* __restrict__ comp,
                              int* __restrict__ cell_comp, const int* __restrict__ flags) {
    if (flags[1]) return;                 // the tree was complete before this round: queued ahead without a host read
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || i == 0) return;
    if (comp[i - 1] == comp[i]) return;
    const unsigned int c = code_s[i], p = code_s[i - 1];
    size_t off = 0;
#pragma unroll
    for (int l = 0; l < CL_PUR_LEVELS; ++l) {
        if ((p >> (3 * l)) == (c >> (3 * l))) cell_comp[off + (c >> (3 * l))] = -1;
        off += (size_t)CL_NCODES >> (3 * l);
    }
}

__device__ __forceinline__ unsigned long long cl_edge_key(int oa, int ob) {
    unsigned int lo = oa < ob ? oa : ob, hi = oa < ob ? ob : oa;
    return ((unsigned long long)lo << 32) | hi;
}

// NT: threads per workgroup.  The walk is bound by its longest dependent chain, not by a CU's resources, and its workgroup's size decides
// how many CUs a launch OCCUPIES: a CU that holds any wave of this kernel cannot take a projection-GEMM tile of another frame's ViT pass
// (that workgroup needs all 160 KB of LDS and every vector register), and the dispatcher deals workgroups round-robin over the CUs:
// 309 workgroups of 256 threads sit on all 256 CUs for the launch's ~300 us, 155 of 512 threads on 155 (two waves per SIMD either way:
// 172 registers, 88 KB of stack per CU).  FAR: the far_list / far_flag entry paths of the cooperative search (development build).
// Measured and dropped (round 5): leaf CHILDREN scanned from their parent's visit.  A walk is a chain of dependent round trips -- one per
// internal node, two per leaf (its own range, then its points) -- and the parent's trip already brings every child's range and purity, so
// in near-to-far order the leaf children in front of the first internal child were scanned right there: the same visits in the same order,
// one round trip less each.  Bit-identical, and SLOWER: Sigma search 2 290 -> 2 610 us per MST (211 instead of 172 registers; what counts is
// the WAVE's chain: before, all lanes that sat at a leaf in one loop iteration shared its round trip; now a lane scanning five leaf children
// inside one iteration keeps the other 63 lanes waiting five trips).
// Measured and dropped (round 5), second attempt at the same chain: a leaf's points requested ONE ITERATION LATE, beside the next node's tables
// (a "pending" range per lane; stale bounds prune less, never wrongly): one round trip per iteration instead of two, bit-identical, 187
// registers -- and 8 % slower (Sigma search 2 330 -> 2 520 us, two interleaved pairs of traces).  Together with the first attempt this says
// the walk is NOT bound by its memory round trips: with ~1.2 waves per SIMD a wave pays for the INSTRUCTIONS of every path some lane takes
// in an iteration (float64 box tests of eight children, the sixteen-point screen + exact chain), and both variants added instructions.
// Measured and dropped (round 4): rounds >= 2 over a compacted list of the points that still have to walk (a few thousand; one 6 us
// launch builds it): bit-identical, and the rounds take as long as before (332 vs 316 us: a round lasts as long as its longest walks,
// however few waves carry them) while the pipeline's frames/s do not move (65.1 vs 65.2) -- the walks do not keep GEMM tiles waiting.
template <int DIM, int NT = 256, bool FAR = false>
__global__ __launch_bounds__(NT) void k_cl_b_search(const float4* __restrict__ spts, const float* __restrict__ stt, int n,
                                                     const ClGrid* __restrict__ gp, const int* __restrict__ cs,
                                                     const int* __restrict__ cell_comp, const unsigned int* __restrict__ cell_e,
                                                     const int* __restrict__ perm,
                                                     const double* __restrict__ core2, const int* __restrict__ comp,
                                                     const int4* __restrict__ aux,
                                                     unsigned long long* __restrict__ best_w,
                                                     unsigned long long* __restrict__ pt_w,
                                                     unsigned long long* __restrict__ pt_d,
                                                     unsigned long long* __restrict__ pt_key, int* __restrict__ pt_b,
                                                     double* __restrict__ pt_lb, int* __restrict__ dbg_scan, const int* __restrict__ flags,
                                                     const int* __restrict__ far_list, const int* __restrict__ far_flag,
                                                     const unsigned long long* __restrict__ giant, int sit_min, int xcd_order) {
    if (flags[1]) return;                 // the tree was complete before this round: queued ahead without a host read
    __shared__ unsigned int stack[CL_STACK * NT];
    // far_list != NULL: only the queries the cooperative search (k_cl_b_search_blk) could not finish inside its shell, each
    // starting from the best edge it found there (pt_w / pt_d / pt_key / pt_b hold it; pt_b = -1: none)
    // (far_flag instead of far_list: the same queries, but each in the lane its point index maps to -- 64 hard walks packed into
    // one wave diverge and run one after the other, spread over all waves they run side by side)
    int a = (xcd_order ? cl_xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x) * blockDim.x + threadIdx.x;
    if (!FAR) { far_list = nullptr; far_flag = nullptr; }
    if (far_list) {
        if (a >= flags[4]) return;
        a = far_list[a];
    }
    if (a >= n) return;
    if (far_flag) {
        if (!far_flag[a]) return;
        far_list = far_flag;                             // from here on: "start from the edge the point holds"
    }
    int scanned = 0;
    if (dbg_scan && !far_list) dbg_scan[a] = 0;
    unsigned int* st = stack + threadIdx.x;
    const ClGrid g = *gp;
    const float4 qf = spts[a];
    const float qtf = DIM >= 5 ? stt[a] : 0.f;
    const double qx = qf.x, qy = qf.y, qz = qf.z, qe = qf.w, qt = qtf;
    if (!far_list && pt_b[a] >= 0) return;           // candidate of an earlier round still valid (k_cl_b_seed)
    const int ca = comp[a];
    if (cl_sits_out(*giant, ca, sit_min)) {          // the largest component does not search this round (k_cl_b_round_init)
        pt_w[a] = CL_NONE; pt_d[a] = ~0ull; pt_key[a] = ~0ull; pt_b[a] = -1;
        return;
    }
    const double core_a = core2[a];
    // Every edge that leaves a's component from a weighs at least lb_a: its mutual-reachability weight is >= core_a, and
    // >= the bound carried over from earlier rounds (a's own minimum foreign edge then, or the winning weight of its
    // component when a was pruned) -- the foreign set only shrinks.  A point whose bound already exceeds the component's
    // published best cannot supply the component's edge: interior points of large components drop out at once.
    const double lb_a = fmax(core_a, pt_lb[a]);
    {
        const unsigned long long cb0 = __hip_atomic_load(&best_w[ca], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cb0 != CL_NONE && lb_a > __longlong_as_double((long long)cb0)) {
            pt_w[a] = CL_NONE; pt_d[a] = ~0ull; pt_key[a] = ~0ull; pt_b[a] = -1;
            return;
        }
    }
    const int oa = perm[a];
    int cx, cy, cz;
    cl_cell_of(g, qx, qy, qz, cx, cy, cz);
    // edges are ordered by (w, d2, original ids): mutual-reachability weight, then the pair's own squared distance.
    // w = core_a for EVERY neighbour inside a's core ball whose core distance is not larger, so ties on w are the
    // rule; breaking them by d2 lets the traversal stop at the nearest such neighbour (nodes farther than bd2 cannot
    // win a tie) instead of scanning the whole ball for the smallest id.
    double bw = INFINITY;          // best weight found by this point
    double bd2 = INFINITY;         // ... and that edge's squared pair distance
    double cbest = INFINITY;       // best weight published for the whole component (read again now and then)
    int since_refresh = 64;        // ... first with the first node
    unsigned long long bkey = ~0ull;   // ... and its id key -- computed only when an exact (w, d2) tie asks for it, and at the end
    bool bkey_valid = true;
    int bb = -1;
    if (far_list && pt_b[a] >= 0) {
        bw = __longlong_as_double((long long)pt_w[a]);
        bd2 = __longlong_as_double((long long)pt_d[a]);
        bkey = pt_key[a];
        bb = pt_b[a];
    }
    // The slowest WAVE sets the launch time (1 234 waves on 1 024 SIMDs: nothing else hides a wave's stalls), and a wave pays for
    // whatever any of its lanes does (per-thread stamps, round 3: the median wave of round 1 runs 180 us in which its slowest
    // lane waits 6 us for node reads and spends 15-45 us in leaf scans -- the rest is other lanes' scans).  Hence:
    //  * a node costs ONE memory round trip: the ranges of its eight children (nine consecutive entries of the level below: they
    //    include the node's own range), its purity entry, its 4th-coordinate range and -- every 8th node and after a leaf scan --
    //    the component's published bound are requested together;
    //  * a leaf scan loads sixteen points per trip, each as two 16-byte records (coordinates; core distance, component, 5th
    //    coordinate): no second trip for the survivors' core distances, none for ids (read only on an exact tie);
    //  * a point of the own component, or farther than the best edge so far, is dropped by a float32 screen.
    const int rx0 = cx >> CL_LMAX, ry0 = cy >> CL_LMAX;
    const int nrx = CL_NX >> CL_LMAX, nry = CL_NY >> CL_LMAX;
    for (int rr = 0; rr < nrx * nry; ++rr) {
        int rx = rr % nrx, ry = rr / nrx;
        if (rr == 0) { rx = rx0; ry = ry0; }
        else if (rx == rx0 && ry == ry0) { rx = 0; ry = 0; }
        int sp = 0;
        st[0] = cl_pack(CL_LMAX, rx, ry, 0);
        sp = 1;
        while (sp > 0) {
            int l, x, y, z;
            cl_unpack(st[(--sp) * NT], l, x, y, z);
            // every edge into this node weighs at least lb; it must be able to tie or beat both bounds (ALU only)
            const double nd2 = cl_box_d2(g, qx, qy, qz, l, x, y, z);
            const double lb = fmax(lb_a, nd2);
            if (lb > bw || lb > cbest || (lb == bw && nd2 > bd2)) continue;
            const unsigned int c = cl_code(x << l, y << l, z << l) >> (3 * l);      // the node's number at its level
            // ---- the node's reads, all in flight together ----
            int bnd[9];
            {
                // children ranges: level l - 1, entries 8c .. 8c + 8 (cells: the reversed table); a cell reads its own two ends
                const long long base = l == 0 ? (long long)CL_NCODES - c
                                     : l == 1 ? (long long)CL_NCODES - 8ll * c
                                              : (long long)(CL_NCODES + 1) + (long long)cl_lvl_off(l - 1) + 8ll * c;
                const long long dir = l <= 1 ? -1 : 1;
#pragma unroll
                for (int u = 0; u < 9; ++u) {
                    long long at = base + dir * u;
                    bnd[u] = cs[at < 0 ? 0 : at];
                }
            }
            const bool has_pur = l < CL_PUR_LEVELS;
            const size_t po = has_pur ? cl_pur_off(l) + c : 0;
            const int pure = cell_comp[po];
            const unsigned int erange = DIM >= 4 ? cell_e[po] : 0u;
            // (the bound: in a late round a whole frame's threads belong to two or three components, and a read per node queues up
            // behind the atomics on those few words)
            const bool refresh = ++since_refresh >= 8;
            unsigned long long cb = 0;
            if (refresh) { cb = __hip_atomic_load(&best_w[ca], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); since_refresh = 0; }
            // ---- use them ----
            const int j0 = bnd[0], j1 = l == 0 ? bnd[1] : bnd[8];
            if (refresh) cbest = cb == CL_NONE ? INFINITY : __longlong_as_double((long long)cb);
            if (lb_a > cbest) { sp = 0; rr = nrx * nry; break; }      // this point can no longer win: stop
            if (j0 == j1) continue;
            if (has_pur && pure == ca) continue;         // all ours
            if (DIM >= 4 && has_pur) {
                const double ed2 = nd2 + cl_e_gap2(erange, qe);
                const double elb = fmax(lb_a, ed2);
                if (elb > bw || elb > cbest || (elb == bw && ed2 > bd2)) continue;
            }
            if (l == 0 || j1 - j0 <= CL_LEAF) {
                bool improved = false;
                scanned += j1 - j0;
                // the float64 distance of the (exactly converted) float32 coordinates differs from the float32 evaluation by a few
                // ulp, the threshold carries a 1e-5 margin: a point the screen drops has d2 > bw and could neither win nor tie
                float thr = (float)bw * 1.00001f + 1e-30f;               // (float)(+inf) stays +inf
                for (int jb = j0; jb < j1; jb += 16) {
                    float4 pj[16];
                    int4 xj[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int j = jb + u < j1 ? jb + u : j1 - 1;
                        pj[u] = spts[j];
                        xj[u] = aux[j];
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const float fx = qf.x - pj[u].x, fy = qf.y - pj[u].y, fz = qf.z - pj[u].z;
                        float sd = fmaf(fz, fz, fmaf(fy, fy, fx * fx));
                        if (DIM >= 4) { const float fe = qf.w - pj[u].w; sd = fmaf(fe, fe, sd); }
                        if (DIM >= 5) { const float ft = qtf - __int_as_float(xj[u].w); sd = fmaf(ft, ft, sd); }
                        if (!(sd <= thr) || xj[u].z == ca || jb + u >= j1) continue;
                        // (the survivors update the best edge by selects: inside this branch every further branch is paid by the
                        // whole wave; only an exact (w, d2) tie -- rare -- goes on to compare ids)
                        const int j = jb + u;
                        const double dx = qx - (double)pj[u].x, dy = qy - (double)pj[u].y, dz = qz - (double)pj[u].z;
                        double d2 = (dx * dx + dy * dy) + dz * dz;
                        if (DIM >= 4) { const double de = qe - (double)pj[u].w; d2 = d2 + de * de; }
                        if (DIM >= 5) { const double dt = qt - (double)__int_as_float(xj[u].w); d2 = d2 + dt * dt; }
                        const double cj2 = __longlong_as_double(((long long)xj[u].y << 32) | (unsigned int)xj[u].x);
                        const double w = fmax(fmax(d2, core_a), cj2);
                        const bool better = w < bw || (w == bw && d2 < bd2);
                        if (w == bw && d2 == bd2) {      // same weight, same pair distance: the smaller id pair
                            if (!bkey_valid) { bkey = cl_edge_key(oa, perm[bb]); bkey_valid = true; }
                            const unsigned long long key = cl_edge_key(oa, perm[j]);
                            if (key < bkey) { bkey = key; bb = j; }
                        }
                        bw = better ? w : bw;
                        bd2 = better ? d2 : bd2;
                        bb = better ? j : bb;
                        bkey_valid = bkey_valid && !better;
                        improved = improved || better;
                        thr = (float)bw * 1.00001f + 1e-30f;
                    }
                }
                if (improved) atomicMin(&best_w[ca], (unsigned long long)__double_as_longlong(bw));
                since_refresh = 64;
                continue;
            }
            // ---- children: nearest one pushed last.  Sibling boxes share their faces: two distances per axis serve all eight; the
            // push is a store + a conditional count (no branch per child) ----
            const int l1 = l - 1;
            const int ox = ((cx >> l1) > 2 * x) ? 1 : 0, oy = ((cy >> l1) > 2 * y) ? 1 : 0, oz = ((cz >> l1) > 2 * z) ? 1 : 0;
            const int near = ox | (oy << 1) | (oz << 2);
            double ad[3][2];
            {
                const double s1 = CL_CELL * (double)(1 << l1);
                const int nbx[3] = {CL_NX >> l1, CL_NY >> l1, CL_NZ >> l1};
                const int c2[3] = {2 * x, 2 * y, 2 * z};
                const double q3[3] = {qx, qy, qz}, o3[3] = {g.ox, g.oy, g.oz};
#pragma unroll
                for (int ax = 0; ax < 3; ++ax)
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb) {     // the same expressions as cl_box_d2, per axis
                        const int cc = c2[ax] + hb;
                        const double lo = o3[ax] + (double)cc * s1, hi = lo + s1;
                        double d = 0.0;
                        if (q3[ax] < lo && cc > 0) d = lo - q3[ax];
                        else if (q3[ax] > hi && cc < nbx[ax] - 1) d = q3[ax] - hi;
                        ad[ax][hb] = d;
                    }
            }
            // child u of the near-first order is octant u ^ near: its range [bnd[u ^ near], bnd[(u ^ near) + 1]) comes out of three conditional
            // exchange stages over the two boundary arrays (48 selects) instead of a 7-deep select chain per child and end (112), and the
            // squared axis distances are taken once per axis half (round 5: the walk is bound by its instruction count)
            int blo[8], bhi[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) { blo[v] = bnd[v]; bhi[v] = bnd[v + 1]; }
#pragma unroll
            for (int bit = 0; bit < 3; ++bit) {
                const bool sw = (near >> bit) & 1;
#pragma unroll
                for (int v = 0; v < 8; ++v)
                    if (!((v >> bit) & 1)) {
                        const int w_ = v | (1 << bit);
                        const int a0 = blo[v], a1 = blo[w_], h0 = bhi[v], h1 = bhi[w_];
                        blo[v] = sw ? a1 : a0; blo[w_] = sw ? a0 : a1;
                        bhi[v] = sw ? h1 : h0; bhi[w_] = sw ? h0 : h1;
                    }
            }
            // (after the exchanges position u holds octant u ^ near; the same for the axis distances: index 0 = the near half)
            const double ax0 = ox ? ad[0][1] : ad[0][0], ax1 = ox ? ad[0][0] : ad[0][1];
            const double ay0 = oy ? ad[1][1] : ad[1][0], ay1 = oy ? ad[1][0] : ad[1][1];
            const double az0 = oz ? ad[2][1] : ad[2][0], az1 = oz ? ad[2][0] : ad[2][1];
            const double sx[2] = {ax0 * ax0, ax1 * ax1}, sy[2] = {ay0 * ay0, ay1 * ay1}, sz[2] = {az0 * az0, az1 * az1};
#pragma unroll
            for (int u = 7; u >= 0; --u) {
                const int ch = u ^ near;
                const int hx = ch & 1, hy = (ch >> 1) & 1, hz = (ch >> 2) & 1;
                double cd2 = 0.0;
                cd2 += sx[u & 1];
                cd2 += sy[(u >> 1) & 1];
                cd2 += sz[(u >> 2) & 1];
                const double clb = fmax(lb_a, cd2);
                const int lo_i = blo[u], hi_i = bhi[u];
                const bool keep = lo_i != hi_i && !(clb > bw || clb > cbest || (clb == bw && cd2 > bd2)) && sp < CL_STACK;
                st[sp * NT] = cl_pack(l1, 2 * x + hx, 2 * y + hy, 2 * z + hz);      // (slot sp is free: written, kept only if counted)
                sp += keep ? 1 : 0;
            }
        }
    }
    if (bb >= 0 && !bkey_valid) bkey = cl_edge_key(oa, perm[bb]);
    if (dbg_scan) dbg_scan[a] = (far_list ? dbg_scan[a] : 0) + scanned;
    if (bb >= 0 && bw <= cbest) {
        pt_w[a] = (unsigned long long)__double_as_longlong(bw);
        pt_d[a] = (unsigned long long)__double_as_longlong(bd2);
        pt_key[a] = bkey;
        pt_b[a] = bb;
        pt_lb[a] = bw;                       // a's true minimum foreign edge: a lower bound from now on
    } else {
        if (cbest < INFINITY && cbest > pt_lb[a]) pt_lb[a] = cbest;      // everything a left unexplored weighs more than this
        pt_w[a] = CL_NONE;
        pt_d[a] = ~0ull;
        pt_key[a] = ~0ull;
        pt_b[a] = -1;
    }
}

#ifdef VG_DEV      // measured slower than the walk alone (LAB_NOTES.md section 3, round 3): kept for the A/B tools only, untested in the product
// Cooperative nearest-foreign search of a Boruvka round (same layout as k_cl_core_blk: one wave per (0.8 m node, 64 queries),
// lane = query, the 27 neighbour nodes staged in LDS by coalesced loads: coordinates as float64, component id, squared core
// distance, original id).  A lane keeps the best edge (w, d2, key) it has met under the walk's strict order.  Everything outside
// the shell is at least r away (r = distance to the shell's outer faces), so its edges weigh >= r^2: a lane whose best edge --
// or whose component's published bound -- is lighter than that is finished; the others start the tree walk from what they found.
// A neighbour node is skipped when it is pure and owned by the component of every searching lane, or when it is farther than
// every searching lane's bounds.
template <int DIM>
__global__ __launch_bounds__(64) void k_cl_b_search_blk(const float4* __restrict__ spts, const float* __restrict__ stt, int n,
                                                        const ClGrid* __restrict__ gp, const int* __restrict__ cs,
                                                        const unsigned int* __restrict__ code_s, const unsigned int* __restrict__ entries,
                                                        const int* __restrict__ cell_comp, const int* __restrict__ perm,
                                                        const double* __restrict__ core2, const int* __restrict__ comp,
                                                        unsigned long long* __restrict__ best_w, unsigned long long* __restrict__ pt_w,
                                                        unsigned long long* __restrict__ pt_d, unsigned long long* __restrict__ pt_key,
                                                        int* __restrict__ pt_b, double* __restrict__ pt_lb, int* __restrict__ dbg_scan,
                                                        int* __restrict__ flags, int* __restrict__ far_list, int* __restrict__ far_flag) {
    if (flags[1]) return;
    __shared__ int bnd[27][3];                        // range + purity of the 27 neighbour nodes
    __shared__ ClbPlan plan;
    __shared__ double tile[DIM + 1][CLB_TILE];        // coordinates, squared core distance
    __shared__ int tile_i[2][CLB_TILE];               // component, original id
    const int lane = threadIdx.x;
    const ClGrid g = *gp;
    const int n_entries = flags[2];
    for (int e = blockIdx.x; e < n_entries; e += gridDim.x) {
        const int lv = (int)(entries[e] >> 30), i0 = (int)(entries[e] & 0x3FFFFFFFu);   // node level (0 .. 3), first query
        const unsigned int key1 = code_s[i0] >> (3 * lv);
        const int iend = min(cl_start_l(cs, lv, key1 + 1u), i0 + 64);
        const int a = i0 + lane;
        const bool active = a < iend;
        const int aa = active ? a : i0;
        bool searching = active && pt_b[aa] < 0;      // a candidate of an earlier round that is still foreign is kept (k_cl_b_seed)
        if (far_flag && active) far_flag[a] = 0;
        if (!__any(searching)) continue;
        const float4 qf = spts[aa];
        const double qx = qf.x, qy = qf.y, qz = qf.z, qe = qf.w, qt = DIM >= 5 ? (double)stt[aa] : 0.0;
        const int ca = comp[aa];
        const double core_a = core2[aa];
        const double lb_a = fmax(core_a, pt_lb[aa]);
        double cbest = g.inf;
        {
            const unsigned long long cb0 = __hip_atomic_load(&best_w[ca], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cb0 != CL_NONE) cbest = __longlong_as_double((long long)cb0);
        }
        if (searching && lb_a > cbest) {              // cannot supply its component's edge (same exit as the walk)
            pt_w[a] = CL_NONE; pt_d[a] = ~0ull; pt_key[a] = ~0ull; pt_b[a] = -1;
            searching = false;
        }
        if (!__any(searching)) continue;
        const int oa = perm[aa];
        int bx, by, bz;
        {
            const float4 f0 = spts[i0];
            cl_cell_of(g, (double)f0.x, (double)f0.y, (double)f0.z, bx, by, bz);
            bx >>= lv; by >>= lv; bz >>= lv;
        }
        __syncthreads();
        if (lane < 27) {
            const int nx = bx + CLB_ORDER[lane][0], ny = by + CLB_ORDER[lane][1], nz = bz + CLB_ORDER[lane][2];
            int j0 = 0, j1 = 0, pure = -1;
            if (nx >= 0 && ny >= 0 && nz >= 0 && nx < (CL_NX >> lv) && ny < (CL_NY >> lv) && nz < (CL_NZ >> lv)) {
                const unsigned int c0 = cl_code(nx << lv, ny << lv, nz << lv);
                j0 = cl_start_l(cs, lv, c0 >> (3 * lv));
                j1 = cl_start_l(cs, lv, (c0 >> (3 * lv)) + 1u);
                if (j0 != j1) pure = cell_comp[cl_pur_off(lv) + (c0 >> (3 * lv))];
            }
            bnd[lane][0] = j0; bnd[lane][1] = j1; bnd[lane][2] = pure;
        }
        __syncthreads();
        double bw = g.inf, bd2 = g.inf;
        unsigned long long bkey = ~0ull;
        int bb = -1, scanned = 0;
        int t_next = 0, cur = bnd[0][0];
        while (t_next < 27) {
            int nseg = 0, tot = 0;
            while (t_next < 27 && tot < CLB_TILE) {
                const int j1 = bnd[t_next][1], pure = bnd[t_next][2];
                bool need = cur < j1;
                if (need) {
                    const double nd2 = cl_box_d2(g, qx, qy, qz, lv, bx + CLB_ORDER[t_next][0], by + CLB_ORDER[t_next][1], bz + CLB_ORDER[t_next][2]);
                    const double lb = fmax(lb_a, nd2);
                    need = __any(searching && pure != ca && !(lb > bw || lb > cbest || (lb == bw && nd2 > bd2)));
                }
                if (need) {
                    const int take = min(j1 - cur, CLB_TILE - tot);
                    if (lane == 0) { plan.start[nseg] = cur; plan.take[nseg] = take; plan.off[nseg] = tot; }
                    ++nseg; tot += take; cur += take;
                    if (cur < j1) break;
                }
                ++t_next;
                if (t_next < 27) cur = bnd[t_next][0];
            }
            if (tot == 0) break;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < CLB_TILE / 64; ++r) {
                const int f = lane + 64 * r;
                if (f < tot) {
                    const int j = clb_locate(plan, nseg, f);
                    const float4 p = spts[j];
                    tile[0][f] = (double)p.x; tile[1][f] = (double)p.y; tile[2][f] = (double)p.z;
                    if (DIM >= 4) tile[3 < DIM ? 3 : 0][f] = (double)p.w;
                    if (DIM >= 5) tile[4 < DIM ? 4 : 0][f] = (double)stt[j];
                    tile[DIM][f] = core2[j];
                    tile_i[0][f] = comp[j];
                    tile_i[1][f] = j;
                }
            }
            __syncthreads();
            scanned += tot;
            for (int c = 0; c < tot; ++c) {
                if (tile_i[0][c] == ca) continue;
                const double dx = qx - tile[0][c], dy = qy - tile[1][c], dz = qz - tile[2][c];
                double d2 = (dx * dx + dy * dy) + dz * dz;
                if (DIM >= 4) { const double de = qe - tile[3 < DIM ? 3 : 0][c]; d2 = d2 + de * de; }
                if (DIM >= 5) { const double dt = qt - tile[4 < DIM ? 4 : 0][c]; d2 = d2 + dt * dt; }
                if (d2 > bw) continue;
                const double w = fmax(fmax(d2, core_a), tile[DIM][c]);
                if (w > bw || (w == bw && d2 > bd2)) continue;
                const int j = tile_i[1][c];
                const unsigned long long key = cl_edge_key(oa, perm[j]);   // (rare: only for edges that tie or win)
                if (w < bw || d2 < bd2 || key < bkey) { bw = w; bd2 = d2; bkey = key; bb = j; }
            }
        }
        if (searching) {
            if (bb >= 0) atomicMin(&best_w[ca], (unsigned long long)__double_as_longlong(bw));
            const unsigned long long cb = __hip_atomic_load(&best_w[ca], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cbest = cb == CL_NONE ? g.inf : __longlong_as_double((long long)cb);
        }
        const double r2 = cl_block_radius2(g, qx, qy, qz, bx, by, bz, lv);
        const bool done = fmin(bw, cbest) < r2;           // strictly lighter than anything outside the shell
        if (searching) {
            if (dbg_scan) dbg_scan[a] = scanned;
            if (done && !(bb >= 0 && bw <= cbest)) {
                if (cbest < g.inf && cbest > pt_lb[a]) pt_lb[a] = cbest;
                pt_w[a] = CL_NONE; pt_d[a] = ~0ull; pt_key[a] = ~0ull; pt_b[a] = -1;
            } else {                                      // final (done), or the walk's starting point
                pt_w[a] = bb >= 0 ? (unsigned long long)__double_as_longlong(bw) : CL_NONE;
                pt_d[a] = bb >= 0 ? (unsigned long long)__double_as_longlong(bd2) : ~0ull;
                pt_key[a] = bkey;
                pt_b[a] = bb;
                if (done) pt_lb[a] = bw;
            }
        }
        const unsigned long long far = __ballot(searching && !done);
        if (far) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&flags[4], __popcll(far));
            base = __shfl(base, 0);
            if (searching && !done) far_list[base + __popcll(far & ((1ull << lane) - 1ull))] = a;
            if (far_flag && searching && !done) far_flag[a] = 1;
        }
    }
}

// per component: smallest w (atomicMin in the search), then smallest d2 among those, then smallest id key among those
#endif  // VG_DEV
__global__ void k_cl_b_select_d(int n, const int* __restrict__ comp, const unsigned long long* __restrict__ best_w,
                                const unsigned long long* __restrict__ pt_w, const unsigned long long* __restrict__ pt_d,
                                unsigned long long* __restrict__ best_d, const int* __restrict__ flags) {
    if (flags[1]) return;                 // the tree was complete before this round: queued ahead without a host read
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    const int c = comp[a];
    if (pt_w[a] != CL_NONE && pt_w[a] == best_w[c]) atomicMin(&best_d[c], pt_d[a]);
}
__global__ void k_cl_b_select(int n, const int* __restrict__ comp, const unsigned long long* __restrict__ best_w,
                              const unsigned long long* __restrict__ best_d, const unsigned long long* __restrict__ pt_w,
                              const unsigned long long* __restrict__ pt_d, const unsigned long long* __restrict__ pt_key,
                              unsigned long long* __restrict__ best_e, const int* __restrict__ flags) {
    if (flags[1]) return;                 // the tree was complete before this round: queued ahead without a host read
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    const int c = comp[a];
    if (pt_w[a] != CL_NONE && pt_w[a] == best_w[c] && pt_d[a] == best_d[c]) atomicMin(&best_e[c], pt_key[a]);
}
__global__ void k_cl_b_pick(int n, const int* __restrict__ comp, const unsigned long long* __restrict__ best_w,
                            const unsigned long long* __restrict__ best_d, const unsigned long long* __restrict__ best_e,
                            const unsigned long long* __restrict__ pt_w, const unsigned long long* __restrict__ pt_d,
                            const unsigned long long* __restrict__ pt_key, const int* __restrict__ pt_b,
                            int* __restrict__ sel_a, int* __restrict__ sel_b, const int* __restrict__ flags,
                            const unsigned long long* __restrict__ giant, int sit_min) {
    if (flags[1]) return;                 // the tree was complete before this round: queued ahead without a host read
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    const int c = comp[a];
    // (the component that sits out picks nothing: some of its points still hold candidates of earlier rounds, valid edges but not
    // necessarily the component's minimum since the others did not search)
    if (cl_sits_out(*giant, c, sit_min)) return;
    if (pt_w[a] != CL_NONE && pt_w[a] == best_w[c] && pt_d[a] == best_d[c] && pt_key[a] == best_e[c]) {
        sel_a[c] = a;
        sel_b[c] = pt_b[a];
    }
}
// Every component root c with a pick hooks onto the component its edge leads to: parent(c) = comp[sel_b[c]] (c itself without a pick).  The
// two roots of a mutual pair chose the same edge: the smaller id becomes the root, its partner emits the edge.  (Until round 5 the parents were
// written by a launch of their own, k_cl_b_link; they are a function of the picks, which are complete when this kernel starts.)
__global__ void k_cl_b_emit(int n, const int* __restrict__ comp, const int* __restrict__ sel_a, const int* __restrict__ sel_b,
                            const unsigned long long* __restrict__ best_w, const int* __restrict__ perm,
                            int* __restrict__ parent2, int* __restrict__ counter, int* __restrict__ mst_a,
                            int* __restrict__ mst_b, unsigned long long* __restrict__ mst_w, const int* __restrict__ flags) {
    if (flags[1]) return;                 // the tree was complete before this round: queued ahead without a host read
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    if (comp[c] != c) return;
    auto parent_of = [&](int r) { return sel_a[r] >= 0 ? comp[sel_b[r]] : r; };
    const int p = parent_of(c);
    if (p == c) { parent2[c] = c; return; }
    const bool mutual = parent_of(p) == c;
    if (mutual && c < p) {
        parent2[c] = c;            // the smaller id of a mutual pair becomes the root; its partner emits the edge
        return;
    }
    parent2[c] = p;
    int k = atomicAdd(counter, 1);
    mst_a[k] = perm[sel_a[c]];
    mst_b[k] = perm[sel_b[c]];
    mst_w[k] = best_w[c];
}

__global__ void k_cl_b_compress(int n, int* __restrict__ comp, const int* __restrict__ parent2, int* __restrict__ flags, int4* __restrict__ aux,
                                int* __restrict__ csize, unsigned long long* __restrict__ giant) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) flags[1] = flags[0] >= n - 1;    // read by every kernel of the NEXT round (stream order): rounds are queued in
                                                // batches without a host read in between, a round after the last one is a no-op
    if (i == 0) *giant = 0ull;                   // the next round's k_cl_b_round_init takes the maximum again
    if (i >= n) return;
    const int old = comp[i];
    int r = old;
    while (parent2[r] != r) r = parent2[r];
    // an absorbed root hands its point count to the new root (only new roots are written, and a new root never reads its own count here)
    if (old == i && r != i) atomicAdd(&csize[r], csize[i]);
    comp[i] = r;
    aux[i].z = r;
}

#ifdef VG_DEV
// Experiment (round 5, VG_CLUSTER_SEEDSIM=1): what would round 1 of the Boruvka search cost if every point whose best edge weighs exactly its
// own core distance -- the points a pass over the k-NN shell could seed without a walk -- were answered beforehand?  After round 1's search
// the candidates of all OTHER points are dropped and the search is launched again: that second launch walks only the points a seed could
// not serve (same results; its duration in the kernel trace is the residual walk).
__global__ void k_cl_seedsim_drop(int n, const double* __restrict__ core2, const unsigned long long* __restrict__ pt_w, int* __restrict__ pt_b,
                                  int* __restrict__ counter) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    const bool seedable = pt_b[a] >= 0 && pt_w[a] == (unsigned long long)__double_as_longlong(core2[a]);
    if (!seedable) { pt_b[a] = -1; atomicAdd(&counter[5], 1); }
}
#endif

__global__ void k_cl_iota(int n, int* p) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

__global__ void k_cl_unsort_core(int n, const int* __restrict__ perm, const double* __restrict__ core2_s,
                                 double* __restrict__ core2_orig) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) core2_orig[perm[i]] = core2_s[i];
}

__global__ void k_cl_gather_edges(int m, const int* __restrict__ idx_s, const int* __restrict__ a,
                                  const int* __restrict__ b, int* __restrict__ lo, int* __restrict__ hi) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    int e = idx_s[i];
    int x = a[e], y = b[e];
    lo[i] = x < y ? x : y;
    hi[i] = x < y ? y : x;
}

// ---------------------------------------------------------------------------------------------
// Fixed-radius queries against the grid of a TARGET point set (entropy scores / two-frame clustering).
// float32 arithmetic in the order the reference's CUDA ops compile to under nvcc's default FMA contraction:
//   d2 = fma(dz, dz, fma(dy, dy, dx * dx))      (pcdet ball_query_kernel_stack; pytorch3d KNearestNeighbor dist += diff*diff)
__device__ __forceinline__ float cl_d2_f32(float qx, float qy, float qz, const float4& b) {
    const float dx = qx - b.x, dy = qy - b.y, dz = qz - b.z;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

// counts[i] = min(cap, #{ target j : d2(q_i, t_j) < r2 })  -- pointcloud_utils.py:74-107 (ball_query + count_nonzero)
// squared distance (float, slightly UNDER-estimated) from q to cell (x,y,z) of the grid: a cell is skipped only when even
// this lower bound exceeds the radius.  Border cells extend to infinity outwards (points beyond the grid are clamped into them).
__device__ __forceinline__ float cl_cell_d2_lb(const ClGrid& g, float qx, float qy, float qz, int x, int y, int z) {
    const float q[3] = {qx, qy, qz};
    const double o[3] = {g.ox, g.oy, g.oz};
    const int c[3] = {x, y, z}, nb[3] = {CL_NX, CL_NY, CL_NZ};
    float d2 = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = (float)(o[a] + (double)c[a] * CL_CELL), hi = (float)(o[a] + (double)(c[a] + 1) * CL_CELL);
        float d = 0.f;
        if (q[a] < lo && c[a] > 0) d = lo - q[a];
        else if (q[a] > hi && c[a] < nb[a] - 1) d = q[a] - hi;
        d2 += d * d;
    }
    return d2 * 0.9999f - 1e-6f;
}

__global__ __launch_bounds__(256) void k_cl_ball_count(const float* __restrict__ q, int nq, int qstride,
                                                       const float4* __restrict__ spts, const ClGrid* __restrict__ gp,
                                                       const int* __restrict__ cs, float r2, int reach, int cap,
                                                       int* __restrict__ counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const ClGrid g = *gp;
    const float qx = q[(size_t)i * qstride], qy = q[(size_t)i * qstride + 1], qz = q[(size_t)i * qstride + 2];
    int cx, cy, cz;
    cl_cell_of(g, qx, qy, qz, cx, cy, cz);
    int cnt = 0;
    for (int z = max(cz - reach, 0); z <= min(cz + reach, CL_NZ - 1); ++z)
        for (int y = max(cy - reach, 0); y <= min(cy + reach, CL_NY - 1); ++y)
            for (int x = max(cx - reach, 0); x <= min(cx + reach, CL_NX - 1); ++x) {
                if (cl_cell_d2_lb(g, qx, qy, qz, x, y, z) >= r2) continue;          // no point of this cell can be inside
                const unsigned int c = cl_code(x, y, z);
                const int j0 = cl_start(cs, c), j1 = cl_start(cs, c + 1u);
                for (int j = j0; j < j1; ++j) cnt += cl_d2_f32(qx, qy, qz, spts[j]) < r2 ? 1 : 0;
                if (cnt >= cap) { counts[i] = cap; return; }                        // the reference stops at nsample hits
            }
    counts[i] = cnt < cap ? cnt : cap;
}

// nearest target point with d2 <= r2max: idx (ORIGINAL target index, lowest index among equidistant ones) and d2, or
// (-1, +inf)  -- pointcloud_utils.py:496-513 (knn K=1 + distance gate)
__global__ __launch_bounds__(256) void k_cl_nearest(const float* __restrict__ q, int nq, int qstride,
                                                    const float4* __restrict__ spts, const int* __restrict__ perm,
                                                    const ClGrid* __restrict__ gp, const int* __restrict__ cs, float r2max,
                                                    int reach, int* __restrict__ idx, float* __restrict__ d2out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const ClGrid g = *gp;
    const float qx = q[(size_t)i * qstride], qy = q[(size_t)i * qstride + 1], qz = q[(size_t)i * qstride + 2];
    int cx, cy, cz;
    cl_cell_of(g, qx, qy, qz, cx, cy, cz);
    float best = INFINITY;
    int bi = -1;
    auto scan = [&](int x, int y, int z) {
        const unsigned int c = cl_code(x, y, z);
        const int j0 = cl_start(cs, c), j1 = cl_start(cs, c + 1u);
        for (int j = j0; j < j1; ++j) {
            const float d2 = cl_d2_f32(qx, qy, qz, spts[j]);
            if (d2 > r2max) continue;
            const int o = perm[j];
            if (d2 < best || (d2 == best && o < bi)) { best = d2; bi = o; }
        }
    };
    if (reach >= 0) scan(cx, cy, cz);                 // the own cell first: its hit prunes most of the others
    for (int z = max(cz - reach, 0); z <= min(cz + reach, CL_NZ - 1); ++z)
        for (int y = max(cy - reach, 0); y <= min(cy + reach, CL_NY - 1); ++y)
            for (int x = max(cx - reach, 0); x <= min(cx + reach, CL_NX - 1); ++x) {
                if (x == cx && y == cy && z == cz) continue;
                const float lb = cl_cell_d2_lb(g, qx, qy, qz, x, y, z);
                if (lb > r2max || lb > best) continue;        // equal distances still compete on the index: strict >
                scan(x, y, z);
            }
    idx[i] = bi;
    d2out[i] = best;
}

// ---------------------------------------------------------------------------------------------
struct MinOp {
    __device__ __host__ int operator()(int a, int b) const { return a < b ? a : b; }
};

extern "C" {

int vg_cluster_create(vg_cluster** out, int max_points) {
    if (!out || max_points <= 0) return VG_ERR_ARG;
    vg_cluster* h = new vg_cluster();
    memset(h, 0, sizeof(*h));
    h->max_points = max_points;
    const size_t n = (size_t)max_points;
    VG_CHECK(hipMalloc(&h->d_grid, sizeof(ClGrid)));
    VG_CHECK(hipMalloc(&h->d_code, 4 * n));
    VG_CHECK(hipMalloc(&h->d_code_s, 4 * n));
    VG_CHECK(hipMalloc(&h->d_perm_in, 4 * n));
    VG_CHECK(hipMalloc(&h->d_perm, 4 * n));
    VG_CHECK(hipMalloc(&h->d_spts, sizeof(float4) * n));
    VG_CHECK(hipMalloc(&h->d_st, 4 * n));
    VG_CHECK(hipMalloc(&h->d_cell_start, 4 * ((size_t)(CL_NCODES + 1) + CL_LVL_TOTAL)));
    {
        size_t tot = 0;
        for (int l = 0; l < CL_PUR_LEVELS; ++l) tot += (size_t)CL_NCODES >> (3 * l);
        VG_CHECK(hipMalloc(&h->d_cell_comp, 4 * tot));
        VG_CHECK(hipMalloc(&h->d_cell_e, 4 * tot));
    }
    VG_CHECK(hipMalloc(&h->d_core2, 8 * n));
    VG_CHECK(hipMalloc(&h->d_comp, 4 * n));
    VG_CHECK(hipMalloc(&h->d_aux, 16 * n));
    VG_CHECK(hipMalloc(&h->d_parent, 4 * n));
    VG_CHECK(hipMalloc(&h->d_parent2, 4 * n));
    VG_CHECK(hipMalloc(&h->d_best_w, 8 * n));
    VG_CHECK(hipMalloc(&h->d_best_e, 8 * n));
    VG_CHECK(hipMalloc(&h->d_sel_a, 4 * n));
    VG_CHECK(hipMalloc(&h->d_sel_b, 4 * n));
    VG_CHECK(hipMalloc(&h->d_pt_w, 8 * n));
    VG_CHECK(hipMalloc(&h->d_pt_key, 8 * n));
    VG_CHECK(hipMalloc(&h->d_pt_d, 8 * n));
    VG_CHECK(hipMalloc(&h->d_pt_lb, 8 * n));
    VG_CHECK(hipMalloc(&h->d_best_d, 8 * n));
    VG_CHECK(hipMalloc(&h->d_pt_b, 4 * n));
    VG_CHECK(hipMalloc(&h->d_counter, 64));
    VG_CHECK(hipMalloc(&h->d_csize, 4 * n));
    VG_CHECK(hipMalloc(&h->d_giant, 8));
    VG_CHECK(hipMalloc(&h->d_entries, 4 * n));
    VG_CHECK(hipMalloc(&h->d_far, 4 * n));
    VG_CHECK(hipMalloc(&h->d_far_flag, 4 * n));
    VG_CHECK(hipMemset(h->d_far_flag, 0, 4 * n));
    VG_CHECK(hipMalloc(&h->d_mst_a, 4 * n));
    VG_CHECK(hipMalloc(&h->d_mst_b, 4 * n));
    VG_CHECK(hipMalloc(&h->d_mst_w, 8 * n));
    VG_CHECK(hipMalloc(&h->d_mst_w_s, 8 * n));
    VG_CHECK(hipMalloc(&h->d_mst_idx, 4 * n));
    VG_CHECK(hipMalloc(&h->d_mst_idx_s, 4 * n));
    VG_CHECK(hipHostMalloc((void**)&h->h_counter, 64));
    h->d_dbg = nullptr;
    if (getenv("VG_CLUSTER_DEBUG")) VG_CHECK(hipMalloc(&h->d_dbg, 4 * n));
    size_t t1 = 0, t2 = 0, t3 = 0;
    VG_CHECK(rocprim::radix_sort_pairs(nullptr, t1, h->d_code, h->d_code_s, h->d_perm_in, h->d_perm, n, 0, 24));
    VG_CHECK(rocprim::radix_sort_pairs(nullptr, t2, h->d_mst_w, h->d_mst_w_s, h->d_mst_idx, h->d_mst_idx_s, n, 0, 64));
    VG_CHECK(rocprim::inclusive_scan(nullptr, t3, h->d_cell_start, h->d_cell_start, (size_t)(CL_NCODES + 1), MinOp()));
    h->temp_bytes = std::max(t1, std::max(t2, t3)) + 256;
    VG_CHECK(hipMalloc(&h->d_temp, h->temp_bytes));
    *out = h;
    return VG_OK;
}

void vg_cluster_destroy(vg_cluster* h) {
    if (!h) return;
    void* ptrs[] = {h->d_grid, h->d_code, h->d_code_s, h->d_perm_in, h->d_perm, h->d_spts, h->d_st, h->d_cell_start, h->d_cell_comp, h->d_cell_e,
                    h->d_core2, h->d_comp, h->d_parent, h->d_parent2, h->d_best_w, h->d_best_e, h->d_sel_a, h->d_sel_b,
                    h->d_pt_w, h->d_pt_key, h->d_pt_d, h->d_pt_lb, h->d_best_d, h->d_pt_b, h->d_counter, h->d_mst_a, h->d_mst_b, h->d_mst_w, h->d_mst_w_s,
                    h->d_mst_idx, h->d_mst_idx_s, h->d_temp, h->d_entries, h->d_far, h->d_far_flag, h->d_aux, h->d_csize, h->d_giant};
    for (void* p : ptrs) (void)hipFree(p);
    (void)hipHostFree(h->h_counter);
    delete h;
}

}  // extern "C"

// bbox -> grid origin -> Morton codes -> radix sort -> sorted points + dense cell-start table
static int cl_build_grid(vg_cluster* h, const float* d_points, int n, int stride, int dim, hipStream_t st) {
    const int nb = vg_div_up(n, 256);
    ClGrid g0;
    memset(&g0, 0, sizeof(g0));
    for (int a = 0; a < 3; ++a) { g0.kmin[a] = 0xFFFFFFFFu; g0.kmax[a] = 0u; }
    g0.inf = INFINITY;
    VG_CHECK(hipMemcpyAsync(h->d_grid, &g0, sizeof(g0), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_cl_bbox, dim3(std::min(nb, 64)), dim3(256), 0, st, d_points, n, stride, h->d_grid);
    hipLaunchKernelGGL(k_cl_grid, dim3(1), dim3(64), 0, st, h->d_grid);
    hipLaunchKernelGGL(k_cl_codes, dim3(nb), dim3(256), 0, st, d_points, n, stride, h->d_grid, h->d_code, h->d_perm_in);
    size_t tb = h->temp_bytes;
    VG_CHECK(rocprim::radix_sort_pairs(h->d_temp, tb, h->d_code, h->d_code_s, h->d_perm_in, h->d_perm, (size_t)n, 0, 24, st));
    {
        size_t tot = (size_t)CL_NCODES + 1;
        hipLaunchKernelGGL(k_cl_fill_int, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, h->d_cell_start, n, tot);
    }
    hipLaunchKernelGGL(k_cl_gather, dim3(nb), dim3(256), 0, st, d_points, n, stride, dim, h->d_perm, h->d_code_s, h->d_spts,
                       h->d_st, h->d_cell_start);
    tb = h->temp_bytes;
    VG_CHECK(rocprim::inclusive_scan(h->d_temp, tb, h->d_cell_start, h->d_cell_start, (size_t)(CL_NCODES + 1), MinOp(), st));
    hipLaunchKernelGGL(k_cl_levels, dim3((unsigned)((CL_LVL_TOTAL + 255) / 256)), dim3(256), 0, st, h->d_cell_start);
    VG_LAUNCH_CHECK();
    h->grid_n = n;
    return VG_OK;
}

template <int DIM>
static void cl_launch_core(vg_cluster* h, int n, int k, hipStream_t st) {
#ifdef VG_DEV
    static const int walk = getenv("VG_CLUSTER_CORE_WALK") ? atoi(getenv("VG_CLUSTER_CORE_WALK")) : 0;   // 1: the per-point tree walk (A/B aid, development build)
    const bool no_far = getenv("VG_CLUSTER_CORE_NOFAR") != nullptr;
    const int far_level = getenv("VG_CLUSTER_FAR_LEVEL") ? atoi(getenv("VG_CLUSTER_FAR_LEVEL")) : 0;
#else
    constexpr int walk = 0, far_level = 0;
    constexpr bool no_far = false;
#endif
#ifdef VG_DEV
    if (walk) {
        hipLaunchKernelGGL((k_cl_core<DIM>), dim3(vg_div_up(n, 256)), dim3(256), 0, st, h->d_spts, h->d_st, n, h->d_grid,
                           h->d_cell_start, h->d_cell_e, k, h->d_core2, h->d_dbg);
    } else
#endif
    {
        // counters[2] = work-list entries of phase A (kept for the Boruvka rounds), counters[3] = queries left for phase B
        (void)hipMemsetAsync(h->d_counter + 2, 0, 8, st);
        hipLaunchKernelGGL(k_cl_blocks, dim3(vg_div_up(n, 256)), dim3(256), 0, st, n, h->d_code_s, h->d_cell_start, h->d_entries, h->d_counter);
        hipLaunchKernelGGL((k_cl_core_blk<DIM>), dim3(std::min(n, 16384)), dim3(64), 0, st, h->d_spts, h->d_st, n, h->d_grid, h->d_cell_start,
                           h->d_code_s, h->d_entries, h->d_counter, k, h->d_core2, h->d_far, h->d_dbg);
        if (!no_far)
            hipLaunchKernelGGL((k_cl_core_far<DIM>), dim3(std::min(n, 8192)), dim3(64), 0, st, h->d_spts, h->d_st, n, h->d_grid, h->d_cell_start,
                               h->d_far, h->d_counter, k, h->d_core2, h->d_dbg, far_level);
    }
    if (h->d_dbg) {
        // VG_CLUSTER_DEBUG=1: pairs evaluated / pairs needed: the exact answer needs n * k distances
        std::vector<int> sc(n);
        int cnt[4] = {0, 0, 0, 0};
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(sc.data(), h->d_dbg, 4 * (size_t)n, hipMemcpyDeviceToHost);
        (void)hipMemcpy(cnt, h->d_counter, 16, hipMemcpyDeviceToHost);
        long long tot = 0;
        for (int v : sc) tot += v;
        std::sort(sc.begin(), sc.end());
        fprintf(stderr, "[cluster dbg] core distances: n %d, k %d, pair distances evaluated %lld = %.1f x the %lld needed (n*k); per point median %d, p99 %d, max %d"
                        "; cooperative: %d (node, 64-query) work items, %d queries left to the far phase\n",
                n, k, tot, (double)tot / ((double)n * k), (long long)n * k, sc[n / 2], sc[(size_t)n * 99 / 100], sc[n - 1], walk ? 0 : cnt[2], walk ? 0 : cnt[3]);
    }
}
// smallest size at which the largest component sits a round out (k_cl_b_round_init): an eighth of the points
static int cl_sit_min(int n) {
#ifdef VG_DEV
    const char* e = getenv("VG_CLUSTER_SITOUT");           // A/B aid (development build), read per call: tools switch it inside one process
    if (e && atoi(e) == 0) return 0x7FFFFFFF;
#endif
    return std::max(2, n / 8);
}
// (sit_min, xcd_order: read ONCE per MST by the caller and handed to every launch of every round -- a development switch that a tool flips
// between a round's search and pick launches would let pick accept candidates of a component that did not search)
template <int DIM>
static void cl_launch_search(vg_cluster* h, int n, hipStream_t st, int sit_min, int xcd_order) {
#ifdef VG_DEV
    // VG_CLUSTER_SEARCH_NT (A/B aid, development build): threads per workgroup of the walk, 256, 512 or 768 (see k_cl_b_search)
    const char* nt_env = getenv("VG_CLUSTER_SEARCH_NT");          // (read per launch: the A/B tool switches it inside one process)
    const int nt = nt_env ? atoi(nt_env) : 512;
#else
    constexpr int nt = 512;
#endif
#ifdef VG_DEV
    // VG_CLUSTER_SEARCH_MODE (A/B aid, development build): 0 = the tree walk alone, 1 = cooperative search, leftovers walk as a compacted
    // list, 2 = cooperative search, leftovers walk in place.  Measured on 150k-point frames the cooperative search + the leftover walks
    // take longer than the walk alone (LAB_NOTES.md section 3)
    static const int mode = getenv("VG_CLUSTER_SEARCH_MODE") ? atoi(getenv("VG_CLUSTER_SEARCH_MODE")) : 0;
    static const int core_walk = getenv("VG_CLUSTER_CORE_WALK") ? atoi(getenv("VG_CLUSTER_CORE_WALK")) : 0;  // (no work list then)
    if (mode != 0 && !core_walk) {
        hipLaunchKernelGGL((k_cl_b_search_blk<DIM>), dim3(std::min(n, 16384)), dim3(64), 0, st, h->d_spts, h->d_st, n, h->d_grid, h->d_cell_start,
                           h->d_code_s, h->d_entries, h->d_cell_comp, h->d_perm, h->d_core2, h->d_comp, h->d_best_w, h->d_pt_w, h->d_pt_d, h->d_pt_key,
                           h->d_pt_b, h->d_pt_lb, h->d_dbg, h->d_counter, h->d_far, mode == 2 ? h->d_far_flag : (int*)nullptr);
        // the queries left over walk the tree from the edge they hold
        hipLaunchKernelGGL((k_cl_b_search<DIM, 256, true>), dim3(vg_div_up(n, 256)), dim3(256), 0, st, h->d_spts, h->d_st, n, h->d_grid,
                           h->d_cell_start, h->d_cell_comp, h->d_cell_e, h->d_perm, h->d_core2, h->d_comp, h->d_aux, h->d_best_w, h->d_pt_w, h->d_pt_d,
                           h->d_pt_key, h->d_pt_b, h->d_pt_lb, h->d_dbg, h->d_counter, mode == 1 ? (const int*)h->d_far : (const int*)nullptr,
                           mode == 2 ? (const int*)h->d_far_flag : (const int*)nullptr, h->d_giant, sit_min, 0);
        return;
    }
#endif
#ifdef VG_DEV
    if (nt == 768) {      // three waves per SIMD (<= 168 registers, 132 KB of stack): 103 instead of 155 workgroups for 79k points (A/B, round 6)
        hipLaunchKernelGGL((k_cl_b_search<DIM, 768>), dim3(vg_div_up(n, 768)), dim3(768), 0, st, h->d_spts, h->d_st, n, h->d_grid,
                           h->d_cell_start, h->d_cell_comp, h->d_cell_e, h->d_perm, h->d_core2, h->d_comp, h->d_aux, h->d_best_w, h->d_pt_w, h->d_pt_d,
                           h->d_pt_key, h->d_pt_b, h->d_pt_lb, h->d_dbg, h->d_counter, (const int*)nullptr, (const int*)nullptr, h->d_giant, sit_min, xcd_order);
        return;
    }
#endif
    if (nt == 512)
        hipLaunchKernelGGL((k_cl_b_search<DIM, 512>), dim3(vg_div_up(n, 512)), dim3(512), 0, st, h->d_spts, h->d_st, n, h->d_grid,
                           h->d_cell_start, h->d_cell_comp, h->d_cell_e, h->d_perm, h->d_core2, h->d_comp, h->d_aux, h->d_best_w, h->d_pt_w, h->d_pt_d,
                           h->d_pt_key, h->d_pt_b, h->d_pt_lb, h->d_dbg, h->d_counter, (const int*)nullptr, (const int*)nullptr, h->d_giant, sit_min, xcd_order);
    else
        hipLaunchKernelGGL((k_cl_b_search<DIM, 256>), dim3(vg_div_up(n, 256)), dim3(256), 0, st, h->d_spts, h->d_st, n, h->d_grid,
                           h->d_cell_start, h->d_cell_comp, h->d_cell_e, h->d_perm, h->d_core2, h->d_comp, h->d_aux, h->d_best_w, h->d_pt_w, h->d_pt_d,
                           h->d_pt_key, h->d_pt_b, h->d_pt_lb, h->d_dbg, h->d_counter, (const int*)nullptr, (const int*)nullptr, h->d_giant, sit_min, xcd_order);
}

extern "C" {

/* Builds the 0.4 m cell grid over a TARGET point set (x,y,z = first three columns) for vg_cluster_ball_count /
 * vg_cluster_nearest.  The grid stays valid until the next vg_cluster_grid / vg_cluster_mst[_nd] on this handle. */
int vg_cluster_grid(vg_cluster* h, const float* d_points, int n, int stride, void* stream) {
    if (!h || n < 0 || stride < 3) return VG_ERR_ARG;
    if (n > h->max_points) return VG_ERR_CAPACITY;
    h->grid_n = 0;
    if (n == 0) return VG_OK;
    if (!d_points) return VG_ERR_ARG;
    return cl_build_grid(h, d_points, n, stride, 3, (hipStream_t)stream);
}

/* d_counts[i] = min(cap, number of grid points t with d2(q_i, t) < r2), float32 d2 = fma(dz,dz,fma(dy,dy,dx*dx))
 * (pcdet ball_query + the count of pointcloud_utils.py:74-107; a query point that is itself in the target set counts). */
int vg_cluster_ball_count(vg_cluster* h, const float* d_query, int nq, int qstride, float r2, int cap, int32_t* d_counts,
                          void* stream) {
    if (!h || nq < 0 || qstride < 3 || !(r2 > 0.f) || cap < 0) return VG_ERR_ARG;
    if (nq == 0) return VG_OK;
    if (!d_query || !d_counts) return VG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (h->grid_n == 0) { VG_CHECK(hipMemsetAsync(d_counts, 0, 4 * (size_t)nq, st)); return VG_OK; }
    const int reach = (int)ceil(sqrt((double)r2) / CL_CELL);
    if (reach > 8) return VG_ERR_ARG;
    hipLaunchKernelGGL(k_cl_ball_count, dim3(vg_div_up(nq, 256)), dim3(256), 0, st, d_query, nq, qstride, h->d_spts, h->d_grid,
                       h->d_cell_start, r2, reach, cap, d_counts);
    VG_LAUNCH_CHECK();
    return VG_OK;
}

/* nearest grid point with float32 d2 <= max_d2: d_idx = its index in the point array given to vg_cluster_grid (lowest
 * index among equidistant points) and d_d2, or -1 / +inf (pointcloud_utils.py:496-513: knn K=1 + squared-distance gate). */
int vg_cluster_nearest(vg_cluster* h, const float* d_query, int nq, int qstride, float max_d2, int32_t* d_idx, float* d_d2,
                       void* stream) {
    if (!h || nq < 0 || qstride < 3 || !(max_d2 > 0.f)) return VG_ERR_ARG;
    if (nq == 0) return VG_OK;
    if (!d_query || !d_idx || !d_d2) return VG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int reach = h->grid_n ? (int)ceil(sqrt((double)max_d2) / CL_CELL) : 0;
    if (reach > 8) return VG_ERR_ARG;
    hipLaunchKernelGGL(k_cl_nearest, dim3(vg_div_up(nq, 256)), dim3(256), 0, st, d_query, nq, qstride, h->d_spts, h->d_perm,
                       h->d_grid, h->d_cell_start, max_d2, h->grid_n ? reach : -1, d_idx, d_d2);
    VG_LAUNCH_CHECK();
    return VG_OK;
}

/* Exact core distances + exact MST of the mutual reachability graph.  SYNCHRONOUS on `stream` (one small
 * device->host counter read per Boruvka round).
 *   d_points [n,stride] f32; dim = 3, 4 or 5 leading columns form the clustering space (x,y,z first: the cell grid and
 *   all pruning bounds use x,y,z only -- a lower bound of the dim-D distance; 5-D = the two-frame input of
 *   zero_shot_detector.py:232-237: x,y,z,entropy,0.1*frame).  Distances: float64, summed left to right.
 *   k = min_samples (<= 15): core = distance to the k-th nearest OTHER point
 *   d_core2  [n] f64 squared core distances, ORIGINAL point order (may be NULL)
 *   d_mst_lo/hi [n-1] int32 original point ids (lo < hi), d_mst_w2 [n-1] f64 squared weights, sorted ascending by
 *   weight (equal weights in unspecified order: vg_hdbscan_tree_host callers sort ties by (lo,hi)).
 *   h_rounds: number of Boruvka rounds (diagnostic, may be NULL). */
int vg_cluster_mst_nd(vg_cluster* h, const float* d_points, int n, int stride, int dim, int k, double* d_core2,
                      int32_t* d_mst_lo, int32_t* d_mst_hi, double* d_mst_w2, int32_t* h_rounds, void* stream) {
    if (!h || !d_points || n < 0 || dim < 3 || dim > 5 || stride < dim || k < 1 || k >= CL_K) return VG_ERR_ARG;
    if (n > h->max_points) return VG_ERR_CAPACITY;
    if (h_rounds) *h_rounds = 0;
    if (n < 2) return VG_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nb = vg_div_up(n, 256);
    {
        const int rc = cl_build_grid(h, d_points, n, stride, dim, st);
        if (rc != VG_OK) return rc;
    }
    if (dim >= 4) {
        hipLaunchKernelGGL(k_cl_e_leaf, dim3(nb), dim3(256), 0, st, n, h->d_code_s, h->d_spts, h->d_cell_e);
        for (int l = 1; l < CL_PUR_LEVELS; ++l)
            hipLaunchKernelGGL(k_cl_e_up, dim3(nb), dim3(256), 0, st, n, l, h->d_code_s, h->d_cell_start, h->d_cell_e);
    }
    if (dim == 3) cl_launch_core<3>(h, n, k, st);
    else if (dim == 4) cl_launch_core<4>(h, n, k, st);
    else cl_launch_core<5>(h, n, k, st);
    if (d_core2) hipLaunchKernelGGL(k_cl_unsort_core, dim3(nb), dim3(256), 0, st, n, h->d_perm, h->d_core2, d_core2);
    VG_LAUNCH_CHECK();
    // ---- Boruvka ----
    hipLaunchKernelGGL(k_cl_b_init, dim3(nb), dim3(256), 0, st, n, h->d_comp, h->d_counter, h->d_pt_b, h->d_pt_lb, h->d_core2,
                       dim >= 5 ? h->d_st : (const float*)nullptr, h->d_aux, h->d_csize, h->d_giant);
    // Rounds are queued in batches WITHOUT a host read in between: every round's kernels start with `if (flags[1]) return`,
    // and k_cl_b_compress sets flags[1] once the n - 1 edges are out, so a round queued after the last needed one costs a
    // dozen empty launches.  The edge count of every round is copied to its own pinned slot; the host reads them once per batch
    // (first batch: CL_FIRST_BATCH rounds -- 150k-point frames need 6-8 -- then one round at a time).
    int rounds = 0, edges = 0, needed = 0;
    int* const flags = h->d_counter;
    int first_batch = CL_FIRST_BATCH;
#ifdef VG_DEV
    if (getenv("VG_CLUSTER_FIRST_BATCH")) first_batch = std::max(1, std::min(12, atoi(getenv("VG_CLUSTER_FIRST_BATCH"))));
#endif
    const int sit_min = cl_sit_min(n);             // development switches: one reading per MST
    int xcd_order = 1;
#ifdef VG_DEV
    if (getenv("VG_CLUSTER_XCD_ORDER")) xcd_order = atoi(getenv("VG_CLUSTER_XCD_ORDER"));
#endif
    auto one_round = [&](int r) {
        hipLaunchKernelGGL(k_cl_b_round_init, dim3(nb), dim3(256), 0, st, n, h->d_comp, h->d_best_w, h->d_best_d, h->d_best_e, h->d_sel_a, flags, h->d_csize, h->d_giant, h->d_code_s, h->d_cell_comp);
        if (r > 1) hipLaunchKernelGGL(k_cl_b_seed, dim3(nb), dim3(256), 0, st, n, h->d_comp, h->d_pt_b, h->d_pt_w, h->d_best_w, flags);
        hipLaunchKernelGGL(k_cl_b_purity, dim3(nb), dim3(256), 0, st, n, h->d_code_s, h->d_comp, h->d_cell_comp, flags);
#ifdef VG_DEV
        // VG_CLUSTER_BOUNDSIM (round 6, VERDICT r5 task 6: would a component-level search pay?).  What a workgroup-per-component search adds
        // to this walk is a component bound every point sees AT ONCE (here: published through best_w by atomicMin, read at the start and every
        // 8th node).  Simulated before being built: every round's search is launched a SECOND time with best_w already holding each
        // component's final minimum (candidates and lower bounds restored to their state in front of the first launch) -- no shared bound
        // can prune more than the true one from the first node on.  The second launches' durations are the floor of that family.
        static const bool boundsim = getenv("VG_CLUSTER_BOUNDSIM") != nullptr;
        static int* sim_b = nullptr; static double* sim_lb = nullptr; static int sim_n = 0;
        if (boundsim) {
            if (sim_n < n) { (void)hipFree(sim_b); (void)hipFree(sim_lb); (void)hipMalloc(&sim_b, (size_t)n * 4); (void)hipMalloc(&sim_lb, (size_t)n * 8); sim_n = n; }
            (void)hipMemcpyAsync(sim_b, h->d_pt_b, (size_t)n * 4, hipMemcpyDeviceToDevice, st);
            (void)hipMemcpyAsync(sim_lb, h->d_pt_lb, (size_t)n * 8, hipMemcpyDeviceToDevice, st);
            if (dim == 3) cl_launch_search<3>(h, n, st, sit_min, xcd_order);
            else if (dim == 4) cl_launch_search<4>(h, n, st, sit_min, xcd_order);
            else cl_launch_search<5>(h, n, st, sit_min, xcd_order);
            // best_w now holds every component's minimum: search again from the same candidates / lower bounds (this launch's outputs are the
            // ones the round goes on with: the same edges -- tests still pass with the switch on)
            (void)hipMemcpyAsync(h->d_pt_b, sim_b, (size_t)n * 4, hipMemcpyDeviceToDevice, st);
            (void)hipMemcpyAsync(h->d_pt_lb, sim_lb, (size_t)n * 8, hipMemcpyDeviceToDevice, st);
        }
#endif
        if (dim == 3) cl_launch_search<3>(h, n, st, sit_min, xcd_order);
        else if (dim == 4) cl_launch_search<4>(h, n, st, sit_min, xcd_order);
        else cl_launch_search<5>(h, n, st, sit_min, xcd_order);
#ifdef VG_DEV
        if (r == 1 && getenv("VG_CLUSTER_SEEDSIM")) {
            (void)hipMemsetAsync(h->d_counter + 5, 0, 4, st);
            hipLaunchKernelGGL(k_cl_seedsim_drop, dim3(nb), dim3(256), 0, st, n, h->d_core2, h->d_pt_w, h->d_pt_b, h->d_counter);
            hipLaunchKernelGGL(k_cl_b_round_init, dim3(nb), dim3(256), 0, st, n, h->d_comp, h->d_best_w, h->d_best_d, h->d_best_e, h->d_sel_a, flags, h->d_csize, h->d_giant, h->d_code_s, h->d_cell_comp);
            hipLaunchKernelGGL(k_cl_b_seed, dim3(nb), dim3(256), 0, st, n, h->d_comp, h->d_pt_b, h->d_pt_w, h->d_best_w, flags);
            hipLaunchKernelGGL(k_cl_b_purity, dim3(nb), dim3(256), 0, st, n, h->d_code_s, h->d_comp, h->d_cell_comp, flags);   // (round_init rewrote the pure values)
            if (dim == 3) cl_launch_search<3>(h, n, st, sit_min, xcd_order);          // walks only the points that were not seedable
            else if (dim == 4) cl_launch_search<4>(h, n, st, sit_min, xcd_order);
            else cl_launch_search<5>(h, n, st, sit_min, xcd_order);
            int left = 0;
            (void)hipMemcpyAsync(&left, h->d_counter + 5, 4, hipMemcpyDeviceToHost, st);
            (void)hipStreamSynchronize(st);
            fprintf(stderr, "[cluster seedsim] round 1: %d of %d points have no edge at their own core distance (the second k_cl_b_search launch walks only these)\n", left, n);
        }
#endif
        hipLaunchKernelGGL(k_cl_b_select_d, dim3(nb), dim3(256), 0, st, n, h->d_comp, h->d_best_w, h->d_pt_w, h->d_pt_d, h->d_best_d, flags);
        hipLaunchKernelGGL(k_cl_b_select, dim3(nb), dim3(256), 0, st, n, h->d_comp, h->d_best_w, h->d_best_d, h->d_pt_w, h->d_pt_d,
                           h->d_pt_key, h->d_best_e, flags);
        hipLaunchKernelGGL(k_cl_b_pick, dim3(nb), dim3(256), 0, st, n, h->d_comp, h->d_best_w, h->d_best_d, h->d_best_e, h->d_pt_w,
                           h->d_pt_d, h->d_pt_key, h->d_pt_b, h->d_sel_a, h->d_sel_b, flags, h->d_giant, sit_min);
        hipLaunchKernelGGL(k_cl_b_emit, dim3(nb), dim3(256), 0, st, n, h->d_comp, h->d_sel_a, h->d_sel_b, h->d_best_w,
                           h->d_perm, h->d_parent2, h->d_counter, h->d_mst_a, h->d_mst_b, h->d_mst_w, flags);
        hipLaunchKernelGGL(k_cl_b_compress, dim3(nb), dim3(256), 0, st, n, h->d_comp, h->d_parent2, flags, h->d_aux, h->d_csize, h->d_giant);
        return hipMemcpyAsync(h->h_counter + ((r - 1) & 15), h->d_counter, 4, hipMemcpyDeviceToHost, st);
    };
    while (edges < n - 1) {
        const int batch = (rounds == 0 && !h->d_dbg) ? first_batch : 1;
        const int first = rounds + 1;
        for (int b = 0; b < batch; ++b) {
            if (++rounds > 64) {
                fprintf(stderr, "[vilgod_hip] vg_cluster_mst: Boruvka did not converge (%d of %d edges)\n", edges, n - 1);
                return VG_ERR_HIP;
            }
            VG_CHECK(one_round(rounds));
        }
        VG_CHECK(hipStreamSynchronize(st));
        for (int r = first; r <= rounds; ++r) {
            const int e = h->h_counter[(r - 1) & 15];
            if (h->d_dbg) {
                std::vector<int> sc(n);
                (void)hipMemcpy(sc.data(), h->d_dbg, 4 * (size_t)n, hipMemcpyDeviceToHost);
                std::sort(sc.begin(), sc.end());
                long long tot = 0; int active = 0;
                for (int v : sc) { tot += v; active += v > 0; }
                fprintf(stderr, "[cluster dbg] round %d: edges %d -> %d, searching threads %d, scanned points: total %lld, median %d, p99 %d, p99.9 %d, max %d\n",
                        r, edges, e, active, tot, sc[n / 2], sc[(size_t)n * 99 / 100], sc[(size_t)n * 999 / 1000], sc[n - 1]);
            }
            if (e == edges && !needed) {
                fprintf(stderr, "[vilgod_hip] vg_cluster_mst: no progress in round %d (%d of %d edges)\n", r, e, n - 1);
                return VG_ERR_HIP;
            }
            edges = e;
            if (edges >= n - 1 && !needed) needed = r;
        }
    }
    rounds = needed;
    if (h_rounds) *h_rounds = rounds;
    // ---- sort edges by weight ----
    const int m = n - 1;
    hipLaunchKernelGGL(k_cl_iota, dim3(vg_div_up(m, 256)), dim3(256), 0, st, m, h->d_mst_idx);
    size_t tb = h->temp_bytes;
    VG_CHECK(rocprim::radix_sort_pairs(h->d_temp, tb, h->d_mst_w, h->d_mst_w_s, h->d_mst_idx, h->d_mst_idx_s, (size_t)m, 0, 64, st));
    if (d_mst_lo && d_mst_hi)
        hipLaunchKernelGGL(k_cl_gather_edges, dim3(vg_div_up(m, 256)), dim3(256), 0, st, m, h->d_mst_idx_s, h->d_mst_a, h->d_mst_b,
                           d_mst_lo, d_mst_hi);
    if (d_mst_w2) VG_CHECK(hipMemcpyAsync(d_mst_w2, h->d_mst_w_s, 8 * (size_t)m, hipMemcpyDeviceToDevice, st));
    VG_LAUNCH_CHECK();
    return VG_OK;
}

int vg_cluster_mst(vg_cluster* h, const float* d_points, int n, int stride, int k, double* d_core2, int32_t* d_mst_lo,
                   int32_t* d_mst_hi, double* d_mst_w2, int32_t* h_rounds, void* stream) {
    return vg_cluster_mst_nd(h, d_points, n, stride, 3, k, d_core2, d_mst_lo, d_mst_hi, d_mst_w2, h_rounds, stream);
}

}  // extern "C"
