// Per-cluster reductions for gfx950.
//
//   k_ref_transform    points_ref = T_ref * [p,1] rounded to float32 (lidar_frame.py:66-69, pointcloud_utils.py:21-46)
//   k_plane_hyp/...    C2: ground plane by RANSAC (lidar_frame.py:96-109 -> pointcloud_utils.fit_plane :375-387 ->
//                      pyransac3d Plane.fit): same algorithm, but the three sample indices of every iteration
//                      come from a counter-based hash (seed, iteration) instead of python's `random`
//                      (un-vendored library + global RNG state: PARITY UNPINNED)
//   k_cluster_filter   B4 + C1: per cluster n, z extent, signed plane distances -> the three active validity
//                      filters (cluster_utils.py:14-15, 48-49, 51-60; objects.py:158-181)
//   k_cluster_box      E1: 2-D convex hull (gift wrapping with exact float64 orientation tests) + minimum-area
//                      rectangle over the hull edges + the box assembly of zero_shot_detector.py:451-461.
//                      Deviation (documented): ALL hull edges are tried; the reference drops the closing edge of
//                      qhull's vertex cycle (pointcloud_utils.py:329-330), whose start vertex is an artefact of qhull.
#include <string.h>
#include <math.h>
#include "common.h"
#include "vilgod_hip.h"

// ---------------------------------------------------------------------------------------------
__global__ void k_ref_transform(const float* __restrict__ src, int n, int stride, const double* __restrict__ T,
                                float* __restrict__ dst) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = src + (size_t)i * stride;
    float* q = dst + (size_t)i * stride;
    double x = p[0], y = p[1], z = p[2];
#pragma unroll
    for (int r = 0; r < 3; ++r) q[r] = (float)(((T[r * 4] * x + T[r * 4 + 1] * y) + T[r * 4 + 2] * z) + T[r * 4 + 3]);
    for (int c = 3; c < stride; ++c) q[c] = p[c];
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long vg_mix64(unsigned long long z) {   // splitmix64 finaliser
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// sample ids of iteration `it`: draw j = 0,1,2,... from the hash stream, skipping repeats (random.sample semantics:
// three distinct indices)
__device__ void vg_sample3(unsigned long long seed, int it, int n, int s[3]) {
    int got = 0;
    for (unsigned int j = 0; got < 3; ++j) {
        unsigned long long h = vg_mix64(seed * 0x100000001B3ull + ((unsigned long long)it << 20) + j);
        int v = (int)(h % (unsigned long long)n);
        bool dup = false;
        for (int t = 0; t < got; ++t) dup |= (s[t] == v);
        if (!dup) s[got++] = v;
    }
}

// one block per hypothesis.  idx: optional index list into pts (NULL = identity).  plane_out[it] = {a,b,c,d}, count_out[it]
__global__ __launch_bounds__(256) void k_plane_hyp(const float* __restrict__ pts, int stride, const int* __restrict__ idx,
                                                   int n, double thresh, unsigned long long seed,
                                                   double* __restrict__ plane_out, int* __restrict__ count_out) {
    __shared__ double pl[4];
    __shared__ int cnt[4];
    // grid = (iterations, PLANE_SPLIT): the inlier count of a hypothesis is summed over PLANE_SPLIT blocks (integer atomics into the
    // zeroed count array: exact, order-free) -- 100 blocks alone leave more than half of the 256 CUs idle for ~140 us
    const int it = blockIdx.x, part = blockIdx.y, nparts = gridDim.y;
    if (threadIdx.x == 0) {
        int s[3];
        vg_sample3(seed, it, n, s);
        double P[3][3];
        for (int t = 0; t < 3; ++t) {
            const float* p = pts + (size_t)(idx ? idx[s[t]] : s[t]) * stride;
            P[t][0] = p[0]; P[t][1] = p[1]; P[t][2] = p[2];
        }
        double A[3] = {P[1][0] - P[0][0], P[1][1] - P[0][1], P[1][2] - P[0][2]};
        double B[3] = {P[2][0] - P[0][0], P[2][1] - P[0][1], P[2][2] - P[0][2]};
        double C[3] = {A[1] * B[2] - A[2] * B[1], A[2] * B[0] - A[0] * B[2], A[0] * B[1] - A[1] * B[0]};
        double nrm = sqrt((C[0] * C[0] + C[1] * C[1]) + C[2] * C[2]);
        C[0] /= nrm; C[1] /= nrm; C[2] /= nrm;
        pl[0] = C[0]; pl[1] = C[1]; pl[2] = C[2];
        pl[3] = -((C[0] * P[1][0] + C[1] * P[1][1]) + C[2] * P[1][2]);
    }
    __syncthreads();
    const double a = pl[0], b = pl[1], c = pl[2], d = pl[3];
    const double inv = sqrt((a * a + b * b) + c * c);
    int local = 0;
    for (int i = part * 256 + threadIdx.x; i < n; i += 256 * nparts) {
        const float* p = pts + (size_t)(idx ? idx[i] : i) * stride;
        double dist = ((((a * (double)p[0] + b * (double)p[1]) + c * (double)p[2]) + d)) / inv;
        if (fabs(dist) <= thresh) local++;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&count_out[it], cnt[0] + cnt[1] + cnt[2] + cnt[3]);
        if (part == 0)
            for (int k = 0; k < 4; ++k) plane_out[it * 4 + k] = pl[k];
    }
}

// first hypothesis with the strictly largest inlier count (pyransac3d keeps the first best); writes best plane
__global__ void k_plane_best(const double* __restrict__ planes, const int* __restrict__ counts, int iters,
                             double* __restrict__ best_plane, int* __restrict__ best_count) {
    if (threadIdx.x != 0) return;
    int b = -1, bc = 0;
    for (int i = 0; i < iters; ++i)
        if (counts[i] > bc) { bc = counts[i]; b = i; }
    for (int k = 0; k < 4; ++k) best_plane[k] = b >= 0 ? planes[b * 4 + k] : 0.0;
    best_count[0] = bc;
}

__global__ void k_plane_inliers(const float* __restrict__ pts, int stride, const int* __restrict__ idx, int n,
                                const double* __restrict__ plane, double thresh, unsigned char* __restrict__ flags) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = plane[0], b = plane[1], c = plane[2], d = plane[3];
    const double inv = sqrt((a * a + b * b) + c * c);
    const float* p = pts + (size_t)(idx ? idx[i] : i) * stride;
    double dist = ((((a * (double)p[0] + b * (double)p[1]) + c * (double)p[2]) + d)) / inv;
    flags[i] = fabs(dist) <= thresh ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// stats[c] = {n, zmin, zmax, dmin, dmax, height}; valid[c] per the three active, required, `and` filters.
__global__ __launch_bounds__(256) void k_cluster_filter(const float* __restrict__ pts, int stride,
                                                        const int* __restrict__ index, const int* __restrict__ seg_off,
                                                        const double* __restrict__ plane, int min_points, int max_points,
                                                        double max_min_height, double min_max_height, double min_height,
                                                        double max_height, float* __restrict__ stats,
                                                        unsigned char* __restrict__ valid) {
    __shared__ float rz[8];
    __shared__ double rd[8];
    const int c = blockIdx.x;
    const int p0 = seg_off[c], n = seg_off[c + 1] - p0;
    const double a = plane[0], b = plane[1], cc = plane[2], d = plane[3];
    const double inv = sqrt((a * a + b * b) + cc * cc);
    float zmin = INFINITY, zmax = -INFINITY;
    double dmin = INFINITY, dmax = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float* p = pts + (size_t)index[p0 + i] * stride;
        zmin = fminf(zmin, p[2]);
        zmax = fmaxf(zmax, p[2]);
        double dist = (((a * (double)p[0] + b * (double)p[1]) + cc * (double)p[2]) + d) / inv;
        dmin = fmin(dmin, dist);
        dmax = fmax(dmax, dist);
    }
    zmin = vg_wave_min(zmin);
    zmax = vg_wave_max(zmax);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        dmin = fmin(dmin, __shfl_xor(dmin, o));
        dmax = fmax(dmax, __shfl_xor(dmax, o));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { rz[w] = zmin; rz[4 + w] = zmax; rd[w] = dmin; rd[4 + w] = dmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) {
            rz[0] = fminf(rz[0], rz[k]); rz[4] = fmaxf(rz[4], rz[4 + k]);
            rd[0] = fmin(rd[0], rd[k]); rd[4] = fmax(rd[4], rd[4 + k]);
        }
        const float height = rz[4] - rz[0];                   // objects.py:112-114 (float32)
        const bool ok_n = n >= min_points && n <= max_points;                        // cluster_utils.py:14-15
        const bool ok_plane = rd[0] <= max_min_height && rd[4] >= min_max_height;    // :58-60
        const bool ok_h = (double)height >= min_height && (double)height <= max_height;   // :48-49
        valid[c] = (ok_n && ok_plane && ok_h) ? 1 : 0;
        float* s = stats + (size_t)c * 6;
        s[0] = (float)n; s[1] = rz[0]; s[2] = rz[4]; s[3] = (float)rd[0]; s[4] = (float)rd[4]; s[5] = height;
    }
}

// ---------------------------------------------------------------------------------------------
#define BOX_MAX_HULL 512
// exact sign of the orientation of (a,b,c) for float32 inputs: differences and their products are exact in float64
__device__ __forceinline__ double vg_orient(double ax, double ay, double bx, double by, double cx, double cy) {
    return (bx - ax) * (cy - ay) - (by - ay) * (cx - ax);
}

// box[c] = {cx, cy, cz, l, w, h, rz} (float64, ref frame); aux[c] = {n_hull, area, degenerate}
__global__ __launch_bounds__(256) void k_cluster_box(const float* __restrict__ pts, int stride,
                                                     const int* __restrict__ index, const int* __restrict__ seg_off,
                                                     double* __restrict__ box, float* __restrict__ aux) {
    __shared__ double hx[BOX_MAX_HULL], hy[BOX_MAX_HULL];
    __shared__ double ang[BOX_MAX_HULL];
    __shared__ double red_v[4];
    __shared__ double red_d[4];
    __shared__ int red_i[4];
    __shared__ float red_z[8];
    __shared__ int sh_cur, sh_start;
    __shared__ double sh_sum[2];
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int p0 = seg_off[c], n = seg_off[c + 1] - p0;
#define PX(i) ((double)pts[(size_t)index[p0 + (i)] * stride])
#define PY(i) ((double)pts[(size_t)index[p0 + (i)] * stride + 1])
    // z extent + mean xy (degenerate fallback) + start vertex: lowest y, then lowest x
    float zmin = INFINITY, zmax = -INFINITY;
    double sx = 0, sy = 0, by = INFINITY, bx = INFINITY;
    int bi = -1;
    for (int i = tid; i < n; i += 256) {
        float z = pts[(size_t)index[p0 + i] * stride + 2];
        zmin = fminf(zmin, z);
        zmax = fmaxf(zmax, z);
        double x = PX(i), y = PY(i);
        sx += x; sy += y;
        if (y < by || (y == by && x < bx)) { by = y; bx = x; bi = i; }
    }
    zmin = vg_wave_min(zmin);
    zmax = vg_wave_max(zmax);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sx += __shfl_xor(sx, o);
        sy += __shfl_xor(sy, o);
        double oy = __shfl_xor(by, o), ox = __shfl_xor(bx, o);
        int oi = __shfl_xor(bi, o);
        if (oi >= 0 && (bi < 0 || oy < by || (oy == by && (ox < bx || (ox == bx && oi < bi))))) { by = oy; bx = ox; bi = oi; }
    }
    if (lane == 0) { red_z[wv] = zmin; red_z[4 + wv] = zmax; red_v[wv] = by; red_d[wv] = bx; red_i[wv] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < 4; ++k) {
            red_z[0] = fminf(red_z[0], red_z[k]);
            red_z[4] = fmaxf(red_z[4], red_z[4 + k]);
            if (red_i[k] >= 0 && (red_i[0] < 0 || red_v[k] < red_v[0] || (red_v[k] == red_v[0] && (red_d[k] < red_d[0] ||
                (red_d[k] == red_d[0] && red_i[k] < red_i[0]))))) { red_v[0] = red_v[k]; red_d[0] = red_d[k]; red_i[0] = red_i[k]; }
        }
        sh_start = red_i[0];
        sh_cur = red_i[0];
    }
    // NOTE: sx, sy partial sums are per wave; finish them through shared memory
    __syncthreads();
    if (lane == 0) { red_v[wv] = sx; red_d[wv] = sy; }
    __syncthreads();
    if (tid == 0) { sh_sum[0] = (red_v[0] + red_v[1]) + (red_v[2] + red_v[3]); sh_sum[1] = (red_d[0] + red_d[1]) + (red_d[2] + red_d[3]); }
    __syncthreads();
    zmin = red_z[0];
    zmax = red_z[4];
    // ---- gift wrapping (counter-clockwise): next = the point with no other point to its right; farthest on ties ----
    bool degenerate = n < 3;
    int hn = 0;                       // hull vertices so far: counted by every thread (uniform), so the loop exit never reads a
                                      // shared counter that thread 0 may already be advancing for the next iteration
    while (!degenerate) {
        const int cur = sh_cur;
        const double cx0 = PX(cur), cy0 = PY(cur);
        if (tid == 0 && hn < BOX_MAX_HULL) { hx[hn] = cx0; hy[hn] = cy0; }
        hn++;
        int best = -1;
        double bxx = 0, byy = 0, bd2 = -1;
        for (int i = tid; i < n; i += 256) {
            double x = PX(i), y = PY(i);
            if (x == cx0 && y == cy0) continue;           // the current vertex itself and its duplicates
            if (best < 0) { best = i; bxx = x; byy = y; bd2 = (x - cx0) * (x - cx0) + (y - cy0) * (y - cy0); continue; }
            double o = vg_orient(cx0, cy0, bxx, byy, x, y);
            double d2 = (x - cx0) * (x - cx0) + (y - cy0) * (y - cy0);
            if (o < 0 || (o == 0 && d2 > bd2)) { best = i; bxx = x; byy = y; bd2 = d2; }
        }
        // combine across the block with the same rule
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            int oi = __shfl_xor(best, o);
            double ox = __shfl_xor(bxx, o), oy = __shfl_xor(byy, o), od = __shfl_xor(bd2, o);
            if (oi >= 0) {
                if (best < 0) { best = oi; bxx = ox; byy = oy; bd2 = od; }
                else {
                    double orr = vg_orient(cx0, cy0, bxx, byy, ox, oy);
                    if (orr < 0 || (orr == 0 && (od > bd2 || (od == bd2 && oi < best)))) { best = oi; bxx = ox; byy = oy; bd2 = od; }
                }
            }
        }
        __syncthreads();
        if (lane == 0) { red_i[wv] = best; red_v[wv] = bxx; red_d[wv] = byy; }
        __syncthreads();
        if (tid == 0) {
            int b = red_i[0];
            double x = red_v[0], y = red_d[0];
            for (int k = 1; k < 4; ++k) {
                if (red_i[k] < 0) continue;
                if (b < 0) { b = red_i[k]; x = red_v[k]; y = red_d[k]; continue; }
                double orr = vg_orient(cx0, cy0, x, y, red_v[k], red_d[k]);
                double d2a = (x - cx0) * (x - cx0) + (y - cy0) * (y - cy0);
                double d2b = (red_v[k] - cx0) * (red_v[k] - cx0) + (red_d[k] - cy0) * (red_d[k] - cy0);
                if (orr < 0 || (orr == 0 && (d2b > d2a || (d2b == d2a && red_i[k] < b)))) { b = red_i[k]; x = red_v[k]; y = red_d[k]; }
            }
            sh_cur = b;
        }
        __syncthreads();
        const int nxt = sh_cur;
        if (nxt < 0) { degenerate = true; break; }                       // all points coincide
        if (PX(nxt) == PX(sh_start) && PY(nxt) == PY(sh_start)) break;    // closed
        if (hn >= BOX_MAX_HULL) break;                                   // safety (keeps a valid, coarser polygon)
    }
    const int nh = degenerate ? 0 : min(hn, BOX_MAX_HULL);
    // hull area (shoelace) to detect collinear input (qhull raises -> reference falls back to a 0.1 m square)
    __shared__ double sh_area2;
    if (tid == 0) {
        double a2 = 0;
        for (int i = 0; i < nh; ++i) {
            int j = (i + 1) % nh;
            a2 += hx[i] * hy[j] - hx[j] * hy[i];
        }
        sh_area2 = a2;
    }
    __syncthreads();
    double out[7];
    float n_hull = (float)nh, area = 0.f, deg = 0.f;
    const float height = zmax - zmin;                                   // zero_shot_detector.py:459 (float32)
    if (nh < 3 || !(fabs(sh_area2) > 0)) {
        // pointcloud_utils.py:322-326: 0.1 m square at the mean, rz = 0
        deg = 1.f;
        double mx = sh_sum[0] / (double)n, my = sh_sum[1] / (double)n;
        out[0] = mx; out[1] = my; out[3] = 0.1; out[4] = 0.1; out[6] = 0.0;
        // corners (-.05,-.05),(.05,-.05),(.05,.05),(-.05,.05): l = |c0-c1| = 0.1, w = |c0-c3| = 0.1
    } else {
        // edge angles mod pi/2, all edges (closing one included)
        for (int i = tid; i < nh; i += 256) {
            int j = (i + 1) % nh;
            double a = atan2(hy[j] - hy[i], hx[j] - hx[i]);
            double m = fmod(a, M_PI / 2.);
            if (m < 0) m += M_PI / 2.;                                   // np.mod semantics (result has the sign of the divisor)
            ang[i] = fabs(m);
        }
        __syncthreads();
        // np.unique: ascending, first minimum wins -> evaluate every angle, keep (area, angle) lexicographic minimum
        double barea = INFINITY, bang = INFINITY;
        for (int i = tid; i < nh; i += 256) {
            const double a = ang[i];
            const double r00 = cos(a), r01 = cos(a - M_PI / 2.), r10 = cos(a + M_PI / 2.), r11 = cos(a);
            double mnx = INFINITY, mxx = -INFINITY, mny = INFINITY, mxy = -INFINITY;
            for (int k = 0; k < nh; ++k) {
                double x = r00 * hx[k] + r01 * hy[k], y = r10 * hx[k] + r11 * hy[k];
                mnx = fmin(mnx, x); mxx = fmax(mxx, x); mny = fmin(mny, y); mxy = fmax(mxy, y);
            }
            double ar = (mxx - mnx) * (mxy - mny);
            if (ar < barea || (ar == barea && a < bang)) { barea = ar; bang = a; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            double oa = __shfl_xor(barea, o), og = __shfl_xor(bang, o);
            if (oa < barea || (oa == barea && og < bang)) { barea = oa; bang = og; }
        }
        __syncthreads();
        if (lane == 0) { red_v[wv] = barea; red_d[wv] = bang; }
        __syncthreads();
        barea = red_v[0]; bang = red_d[0];
        for (int k = 1; k < 4; ++k)
            if (red_v[k] < barea || (red_v[k] == barea && red_d[k] < bang)) { barea = red_v[k]; bang = red_d[k]; }
        // corners (pointcloud_utils.py:359-370) and the box of zero_shot_detector.py:452-461
        const double a = bang;
        const double r00 = cos(a), r01 = cos(a - M_PI / 2.), r10 = cos(a + M_PI / 2.), r11 = cos(a);
        double mnx = INFINITY, mxx = -INFINITY, mny = INFINITY, mxy = -INFINITY;
        for (int k = 0; k < nh; ++k) {
            double x = r00 * hx[k] + r01 * hy[k], y = r10 * hx[k] + r11 * hy[k];
            mnx = fmin(mnx, x); mxx = fmax(mxx, x); mny = fmin(mny, y); mxy = fmax(mxy, y);
        }
        // rval[k] = [u, v] @ r  ->  (u*r00 + v*r10, u*r01 + v*r11)
        const double c0x = mxx * r00 + mny * r10, c0y = mxx * r01 + mny * r11;
        const double c1x = mnx * r00 + mny * r10, c1y = mnx * r01 + mny * r11;
        const double c2x = mnx * r00 + mxy * r10, c2y = mnx * r01 + mxy * r11;
        const double c3x = mxx * r00 + mxy * r10, c3y = mxx * r01 + mxy * r11;
        double l = sqrt((c0x - c1x) * (c0x - c1x) + (c0y - c1y) * (c0y - c1y));
        double w = sqrt((c0x - c3x) * (c0x - c3x) + (c0y - c3y) * (c0y - c3y));
        double rz = a;
        if (w > l) { double t = l; l = w; w = t; rz += M_PI / 2; }
        out[0] = (c0x + c2x) / 2; out[1] = (c0y + c2y) / 2; out[3] = l; out[4] = w; out[6] = rz;
        area = (float)barea;
    }
    out[2] = (double)zmin + (double)height / 2;
    out[5] = (double)height + 0.3;
    if (tid == 0) {
        for (int k = 0; k < 7; ++k) box[(size_t)c * 7 + k] = out[k];
        aux[c * 3 + 0] = n_hull; aux[c * 3 + 1] = area; aux[c * 3 + 2] = deg;
    }
#undef PX
#undef PY
}

// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// entropy (PP) score of a query frame from its neighbour counts; numpy's pairwise summation order
// (n < 8: left to right; otherwise 8 running partial sums over blocks of 8, combined as a tree, then the tail).
__device__ double vg_np_pairwise(const double* v, int n) {
    if (n < 8) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += v[i];
        return s;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = v[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += v[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += v[i];
    return res;
}

__global__ void k_entropy_scores(const int* __restrict__ counts, int nf, int nq, int seek_row, double* __restrict__ H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    double c[128], t[128];
    long long tot = 0;
    for (int f = 0; f < nf; ++f) {
        int v = counts[(size_t)f * nq + i];
        if (f == seek_row) v -= 1;
        c[f] = (double)v;
        tot += v;
    }
    const double den = (double)tot + 1e-8;
    for (int f = 0; f < nf; ++f) {
        const double P = c[f] / den;
        t[f] = -P * log(P + 1e-8);
    }
    H[i] = vg_np_pairwise(t, nf) / log((double)nf);
}

__global__ void k_subsample_keys(unsigned long long seed, unsigned long long tag, int n, long long* __restrict__ keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = (long long)(vg_mix64(seed * 0x100000001B3ull + (tag << 32) + (unsigned long long)i) >> 1);
}

// ---------------------------------------------------------------------------------------------
// Detection.cluster_mass_center = np.median(cluster_points, axis=0) (objects.py:121-123) for every packed cluster
// and the first n_cols columns of the point rows: exact order statistics by a 4-pass byte radix select per (cluster, column);
// an even count gives the float32 mean of the two middle values, like np.median on a float32 array.
// Round 4 (the kernel took 3 ms per frame in the entry point's trace: one workgroup per cluster walked its five columns one after the
// other, every pass gathered the column again from global memory, and the coordinates of one cluster share their leading bytes, so
// 256 threads added to ONE histogram bin with one LDS atomic each): a workgroup per (cluster, column); the column's keys are staged
// in LDS once (<= MED_CAP points; larger clusters keep gathering); a wave adds a bin's count once per DISTINCT bin among its 64 keys.
#define MED_CAP 12288
__device__ __forceinline__ void vg_hist_add_wave(uint32_t* hist, uint32_t bin, bool active) {
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t lb = (uint32_t)__builtin_amdgcn_readlane((int)bin, leader);
        const unsigned long long same = __ballot(active && bin == lb) & todo;
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[lb], (uint32_t)__popcll(same));
        todo &= ~same;
    }
}

template <bool STAGED>
__device__ float vg_select_keys(const float* __restrict__ pts, int stride, int col, const int* __restrict__ idx, const uint32_t* keys,
                                int n, int k, uint32_t* hist, uint32_t* sh) {
    uint32_t prefix = 0;
    const int n64 = (n + 63) & ~63;                    // whole waves run the aggregation (ballots need every lane of the wave)
    for (int pass = 3; pass >= 0; --pass) {
        for (int b = threadIdx.x; b < 256; b += blockDim.x) hist[b] = 0;
        __syncthreads();
        const int shift = pass * 8;
        for (int i = threadIdx.x; i < n64; i += blockDim.x) {
            uint32_t key = 0;
            if (i < n) key = STAGED ? keys[i] : vg_fkey(pts[(size_t)idx[i] * stride + col]);
            const bool in = i < n && (pass == 3 || (key >> (shift + 8)) == prefix);
            vg_hist_add_wave(hist, (key >> shift) & 255u, in);
        }
        __syncthreads();
        {   // the bin that holds rank k: exclusive prefix of the 256 counts by wave scans; exactly one thread finds it
            const int t = threadIdx.x;
            const uint32_t cnt = hist[t];
            uint32_t inc = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = __shfl_up(inc, o);
                if ((t & 63) >= o) inc += up;
            }
            if ((t & 63) == 63) sh[2 + (t >> 6)] = inc;
            __syncthreads();
            uint32_t base = 0;
            for (int w = 0; w < (t >> 6); ++w) base += sh[2 + w];
            const uint32_t excl = base + inc - cnt;
            if (excl <= (uint32_t)k && (uint32_t)k < excl + cnt) { sh[0] = (uint32_t)t; sh[1] = excl; }
        }
        __syncthreads();
        prefix = (prefix << 8) | sh[0];
        k -= (int)sh[1];
        __syncthreads();
    }
    return vg_fkey_inv(prefix);
}

__global__ __launch_bounds__(256) void k_cluster_medians(const float* __restrict__ pts, int stride, int n_cols,
                                                         const int* __restrict__ index, const int* __restrict__ seg_off,
                                                         float* __restrict__ out) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sh[6];
    __shared__ uint32_t keys[MED_CAP];
    const int c = blockIdx.x, col = blockIdx.y;
    const int p0 = seg_off[c], n = seg_off[c + 1] - p0;
    float m = 0.f;
    if (n > 0) {
        const int* idx = index + p0;
        if (n <= MED_CAP) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) keys[i] = vg_fkey(pts[(size_t)idx[i] * stride + col]);
            __syncthreads();
            const float hi = vg_select_keys<true>(pts, stride, col, idx, keys, n, n / 2, hist, sh);
            m = (n & 1) ? hi : (vg_select_keys<true>(pts, stride, col, idx, keys, n, n / 2 - 1, hist, sh) + hi) / 2.0f;
        } else {
            const float hi = vg_select_keys<false>(pts, stride, col, idx, keys, n, n / 2, hist, sh);
            m = (n & 1) ? hi : (vg_select_keys<false>(pts, stride, col, idx, keys, n, n / 2 - 1, hist, sh) + hi) / 2.0f;
        }
    }
    if (threadIdx.x == 0) out[(size_t)c * n_cols + col] = m;
}

extern "C" {

int vg_cluster_medians(const float* d_points, int stride, int n_cols, const int32_t* d_index, const int32_t* d_seg_off, int n_clusters,
                       float* d_median, void* stream) {
    if (n_clusters <= 0) return VG_OK;
    if (!d_points || !d_index || !d_seg_off || !d_median || n_cols <= 0 || n_cols > stride) return VG_ERR_ARG;
    hipLaunchKernelGGL(k_cluster_medians, dim3(n_clusters, n_cols), dim3(256), 0, (hipStream_t)stream, d_points, stride, n_cols, d_index, d_seg_off,
                       d_median);
    VG_LAUNCH_CHECK();
    return VG_OK;
}

int vg_ref_transform(const float* d_src, int n, int stride, const double* d_T4x4, float* d_dst, void* stream) {
    if (n <= 0) return VG_OK;
    if (!d_src || !d_dst || !d_T4x4 || stride < 3) return VG_ERR_ARG;
    hipLaunchKernelGGL(k_ref_transform, dim3(vg_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, d_src, n, stride, d_T4x4, d_dst);
    VG_LAUNCH_CHECK();
    return VG_OK;
}

/* d_work: >= iters*(4*8+4) + 64 bytes of scratch.  d_plane4: best plane {a,b,c,d} (f64).  d_flags: [n] inlier flags of it.
 * d_count: [1] inlier count. */
int vg_plane_ransac(const float* d_points, int stride, const int32_t* d_index, int n, double thresh, int iters,
                    uint64_t seed, void* d_work, double* d_plane4, uint8_t* d_flags, int32_t* d_count, void* stream) {
    if (!d_points || !d_work || !d_plane4 || !d_flags || !d_count || iters <= 0 || stride < 3) return VG_ERR_ARG;
    if (n < 3) return VG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    double* planes = (double*)d_work;
    int* counts = (int*)((char*)d_work + (size_t)iters * 32);
    VG_CHECK(hipMemsetAsync(counts, 0, sizeof(int) * (size_t)iters, st));
    hipLaunchKernelGGL(k_plane_hyp, dim3(iters, 4), dim3(256), 0, st, d_points, stride, d_index, n, thresh,
                       (unsigned long long)seed, planes, counts);
    hipLaunchKernelGGL(k_plane_best, dim3(1), dim3(64), 0, st, planes, counts, iters, d_plane4, d_count);
    hipLaunchKernelGGL(k_plane_inliers, dim3(vg_div_up(n, 256)), dim3(256), 0, st, d_points, stride, d_index, n, d_plane4, thresh, d_flags);
    VG_LAUNCH_CHECK();
    return VG_OK;
}

int vg_cluster_filter(const float* d_points, int stride, const int32_t* d_index, const int32_t* d_seg_off, int n_clusters,
                      const double* d_plane4, int min_points, int max_points, double max_min_height, double min_max_height,
                      double min_height, double max_height, float* d_stats6, uint8_t* d_valid, void* stream) {
    if (n_clusters <= 0) return VG_OK;
    if (!d_points || !d_index || !d_seg_off || !d_plane4 || !d_stats6 || !d_valid) return VG_ERR_ARG;
    hipLaunchKernelGGL(k_cluster_filter, dim3(n_clusters), dim3(256), 0, (hipStream_t)stream, d_points, stride, d_index, d_seg_off,
                       d_plane4, min_points, max_points, max_min_height, min_max_height, min_height, max_height, d_stats6, d_valid);
    VG_LAUNCH_CHECK();
    return VG_OK;
}

int vg_cluster_boxes(const float* d_points, int stride, const int32_t* d_index, const int32_t* d_seg_off, int n_clusters,
                     double* d_box7, float* d_aux3, void* stream) {
    if (n_clusters <= 0) return VG_OK;
    if (!d_points || !d_index || !d_seg_off || !d_box7 || !d_aux3) return VG_ERR_ARG;
    hipLaunchKernelGGL(k_cluster_box, dim3(n_clusters), dim3(256), 0, (hipStream_t)stream, d_points, stride, d_index, d_seg_off,
                       d_box7, d_aux3);
    VG_LAUNCH_CHECK();
    return VG_OK;
}

int vg_entropy_scores(const int32_t* d_counts, int n_frames, int nq, int seek_row, double* d_H, void* stream) {
    if (nq < 0 || n_frames < 2 || n_frames > 128) return VG_ERR_ARG;
    if (nq == 0) return VG_OK;
    if (!d_counts || !d_H) return VG_ERR_ARG;
    hipLaunchKernelGGL(k_entropy_scores, dim3(vg_div_up(nq, 128)), dim3(128), 0, (hipStream_t)stream, d_counts, n_frames, nq,
                       seek_row, d_H);
    VG_LAUNCH_CHECK();
    return VG_OK;
}

int vg_subsample_keys(uint64_t seed, uint64_t tag, int n, int64_t* d_keys, void* stream) {
    if (n < 0) return VG_ERR_ARG;
    if (n == 0) return VG_OK;
    if (!d_keys) return VG_ERR_ARG;
    hipLaunchKernelGGL(k_subsample_keys, dim3(vg_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, seed, tag, n,
                       (long long*)d_keys);
    VG_LAUNCH_CHECK();
    return VG_OK;
}

}  // extern "C"
