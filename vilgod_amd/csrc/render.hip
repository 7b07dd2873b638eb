// Per-cluster multi-view depth renderer for gfx950.
//
// Replaces, per frame, the reference's per-cluster Python loop
//   src/vilgod/zero_shot_detector.py:389-409  (apply_transform -> transform_cluster_points_to_origin
//   -> RealisticProjection.get_img -> interpolate -> uint8) and third_party/CLIP/clip/clip.py:79-86.
// All clusters x views of a frame are rendered by ONE launch; nothing leaves HBM.
//
// Kernels
//   k_gather_ego      ego[i] = f32( T_ego * [ref[idx[i]],1] )            (pointcloud_utils.py:21-46)
//   k_cluster_median  per-cluster per-axis median (np.median semantics) + view-direction rotation
//                                                                         (pointcloud_utils.py:396-398)
//   k_to_origin       D1 in float64, rounded to float32                  (pointcloud_utils.py:399-412)
//   k_render          D2-D6: one 1024-thread workgroup per (cluster, view); the 112x112 depth
//                     slice, the 5x5 max-pool, the 3x3 Gaussian and the 110x110 image live in
//                     LDS (~100 KB of the CU's 160 KB); only the final 224x224 crop is written.
//
// Arithmetic follows torch-CPU float32 op for op (orders established empirically against the
// reference run on CPU, see DESIGN.md "renderer numerics"): matmul and 3x3 conv are FMA chains,
// bilinear is fma(a,wa,b*wb).  This file is compiled with -ffp-contract=off so that only the
// FMAs written below exist.
#include "common.h"
#include <hip/hip_fp16.h>

#define GR 112           // grid resolution (waymo.yaml: resolution)
#define GO 110           // image side after max-pool (112 + 2*1 - 5 + 1)
#define OUT 224          // CLIP input side
#define RT 1024          // threads per render workgroup
#define ACC_PER_THREAD 12  // ceil(110*110 / 1024)

// ---------------------------------------------------------------------------------------------
__global__ void k_gather_ego(const float* __restrict__ pts, int stride, const int* __restrict__ idx,
                             int n, const double* __restrict__ T, float* __restrict__ ego) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = pts + (size_t)(idx ? idx[i] : i) * stride;
    double x = p[0], y = p[1], z = p[2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double v = ((T[r * 4 + 0] * x + T[r * 4 + 1] * y) + T[r * 4 + 2] * z) + T[r * 4 + 3];
        ego[(size_t)i * 3 + r] = (float)v;
    }
}

// ---------------------------------------------------------------------------------------------
// k-th smallest (0-based) of the `axis` coordinate over one cluster: 4-pass byte radix select.
// Called by a 256-thread GROUP of the workgroup (`t` = thread index inside the group) with the group's own hist / sh arrays; every
// group of the workgroup makes the same number of calls, so the workgroup barriers inside line up.
__device__ float radix_select(const float* __restrict__ v, int n, int axis, int k, uint32_t* hist,
                              uint32_t* sh, int t) {
    uint32_t prefix = 0;
    for (int pass = 3; pass >= 0; --pass) {
        hist[t] = 0;
        __syncthreads();
        int shift = pass * 8;
        for (int i = t; i < n; i += 256) {
            uint32_t key = vg_fkey(v[(size_t)i * 3 + axis]);
            bool match = (pass == 3) || ((key >> (shift + 8)) == prefix);
            if (match) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        // the bin that holds rank k: exclusive prefix of the 256 counts by wave scans (a serial walk by one thread cost ~16 k cycles
        // per pass); exactly one thread finds excl <= k < excl + count
        {
            const uint32_t cnt = hist[t];
            uint32_t inc = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = __shfl_up(inc, o);
                if ((t & 63) >= o) inc += up;
            }
            if ((t & 63) == 63) sh[2 + (t >> 6)] = inc;                // wave totals
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

// out_med[c] = median xyz (float32, np.median semantics); out_rot[c] = {m00,m01,m10,m11,m22,angle} of
// scipy Rotation.from_euler('z', -atan2(med_y, med_x)).as_matrix() in float64.
// 768 threads: the three axes are selected at the same time, one 256-thread group each (a third of the dependent passes).
__global__ __launch_bounds__(768) void k_cluster_median(const float* __restrict__ ego,
                                                        const int* __restrict__ seg_off,
                                                        float* __restrict__ out_med,
                                                        double* __restrict__ out_rot) {
    __shared__ uint32_t hist[3][256];
    __shared__ uint32_t sh[3][6];            // per axis group: bin, rank offset, four wave totals
    __shared__ float med[3];
    int c = blockIdx.x;
    int p0 = seg_off[c], n = seg_off[c + 1] - p0;
    const float* v = ego + (size_t)p0 * 3;
    {
        const int a = threadIdx.x >> 8, t = threadIdx.x & 255;
        float m = 0.f;
        if (n > 0) {                                                  // n is uniform: every group takes the same branches
            float hi = radix_select(v, n, a, n / 2, hist[a], sh[a], t);
            if (n & 1) {
                m = hi;
            } else {
                float lo = radix_select(v, n, a, n / 2 - 1, hist[a], sh[a], t);
                m = (lo + hi) / 2.0f;  // np.mean of the two middle float32 values
            }
        }
        if (t == 0) med[a] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out_med[c * 3 + 0] = med[0];
        out_med[c * 3 + 1] = med[1];
        out_med[c * 3 + 2] = med[2];
        // np.arctan2 on float32 -> float32; scipy converts -angle to float64 and builds the
        // quaternion (0,0,sin(a/2),cos(a/2)), then the matrix from the quaternion.
        float ang = (float)atan2((double)med[1], (double)med[0]);
        double a = -(double)ang;
        double s = sin(a * 0.5), w = cos(a * 0.5);
        double z2 = s * s, w2 = w * w, zw = s * w;
        out_rot[c * 6 + 0] = -z2 + w2;        // m00 = x2 - y2 - z2 + w2
        out_rot[c * 6 + 1] = 2.0 * (-zw);     // m01 = 2 (xy - zw)
        out_rot[c * 6 + 2] = 2.0 * zw;        // m10 = 2 (xy + zw)
        out_rot[c * 6 + 3] = -z2 + w2;        // m11 = -x2 + y2 - z2 + w2
        out_rot[c * 6 + 4] = z2 + w2;         // m22 = -x2 - y2 + z2 + w2
        out_rot[c * 6 + 5] = (double)ang;     // the float32 view angle itself (diagnostics / parity tests)
    }
}

// The rotation entries of k_cluster_median for view angles supplied by the caller (angle_mode='reference': the host's
// own float32 np.arctan2 of the medians, which is what the reference evaluates at pointcloud_utils.py:397).
__global__ void k_cluster_rot(const float* __restrict__ angle, int n, double* __restrict__ out_rot) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    float ang = angle[c];
    double a = -(double)ang;
    double s = sin(a * 0.5), w = cos(a * 0.5);
    double z2 = s * s, w2 = w * w, zw = s * w;
    out_rot[c * 6 + 0] = -z2 + w2;
    out_rot[c * 6 + 1] = 2.0 * (-zw);
    out_rot[c * 6 + 2] = 2.0 * zw;
    out_rot[c * 6 + 3] = -z2 + w2;
    out_rot[c * 6 + 4] = z2 + w2;
    out_rot[c * 6 + 5] = (double)ang;
}

// ---------------------------------------------------------------------------------------------
// D1 (pointcloud_utils.py:399-412).  cluster id of point i comes from a per-point label array.
__global__ void k_to_origin(const float* __restrict__ ego, const int* __restrict__ pt_cluster, int n,
                            const float* __restrict__ med, const double* __restrict__ rot,
                            const double* __restrict__ Timg /*3x3 = Rx(pi) @ Rz(pi/2)*/,
                            float* __restrict__ origin) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c = pt_cluster[i];
    float xs = ego[(size_t)i * 3 + 0] - med[c * 3 + 0];   // float32 subtraction (pts_ is float32)
    float ys = ego[(size_t)i * 3 + 1] - med[c * 3 + 1];
    double x = xs, y = ys, z = ego[(size_t)i * 3 + 2];
    const double* m = rot + (size_t)c * 6;
    double q0 = m[0] * x + m[1] * y;
    double q1 = m[2] * x + m[3] * y;
    double q2 = m[4] * z;
    q0 -= 1.0;
    double w0 = q2, w1 = q1, w2 = q0;  // np.stack([z, y, x])
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double o = (Timg[r * 3 + 0] * w0 + Timg[r * 3 + 1] * w1) + Timg[r * 3 + 2] * w2;
        origin[(size_t)i * 3 + r] = (float)o;
    }
}

// ---------------------------------------------------------------------------------------------
struct RenderArgs {
    const float* origin;   // [Ptot,3] D1 output, clusters packed back to back
    const int* seg_off;    // [C+1]
    const float* view_rot; // [V,9] row-major, points @ rot
    const float* lut;      // [3,256] CLIP-normalised value of each uint8 level
    void* out;
    int n_views;
    int n_clusters;
    int out_kind;          // 0: uint8 [n,224,224,3] (PIL layout)  1: f32 [n,3,224,224]  2: f16 [n,3,224,224]
                           // 3: f32 [n,110,110] = one channel of get_img() before the resize
                           // 4: f16 patch rows [n*196, 768] = the im2col of kind 2 for 16x16 patches (ViT-B/16 input of
                           //    the patch-embedding GEMM: row = crop*196 + py*14 + px, column = ch*256 + i*16 + j)
                           // 5: f16 SINGLE-CHANNEL patch rows [n*196, 256]: column = i*16 + j, value = uint8 level * 2^-8 (exact in fp16).
                           //    The three channels of a crop are the same image (mv_utils.py:36) and differ only by the per-channel
                           //    normalisation (clip.py:79-86), which is affine: vg_vit_encode input_kind 3 folds it and the channel sum
                           //    into a K = 256 conv1 weight + a per-feature constant.  A third of kind 4's bytes.
};

__device__ __forceinline__ void quantise_point(float px, float py, float pz, const float* pc, float prange,
                                               int& gx, int& gy, int& gz, float& val) {
    // mv_utils.py:105-118, float32 op for op
    float nx = (px - pc[0]) / prange * 2.0f;
    float ny = (py - pc[1]) / prange * 2.0f;
    float nz = (pz - pc[2]) / prange * 2.0f;
    nx = nx * 0.8f;
    ny = ny * 0.8f;
    float fx = ceilf((nx + 1.0f) / 2.0f * 112.0f);
    float fy = ceilf((ny + 1.0f) / 2.0f * 112.0f);
    float fz = ((nz + 1.0f) / 2.0f + 0.2f) / 1.2f * 6.0f;
    float zi = ceilf(fz);
    fx = fminf(fmaxf(fx, 1.0f), 110.0f);
    fy = fminf(fmaxf(fy, 1.0f), 110.0f);
    val = fminf(fmaxf(fz, 1.0f), 6.0f);
    gx = (int)fx;
    gy = (int)fy;
    gz = (int)zi;
}

__global__ __launch_bounds__(RT) void k_render(RenderArgs a) {
    extern __shared__ float lds[];
    float* S = lds;                  // [112][112] depth slice, later [110][110] pooled
    float* T = lds + GR * GR;        // [112][110] row-pooled, later [110][110] final image
    __shared__ float red[16 * 6];
    __shared__ float bc[8];          // pcent[3], prange, image max
    __shared__ unsigned int slice_mask;
    __shared__ int t_i0[OUT];
    __shared__ float t_l0[OUT], t_l1[OUT];

    const int tid = threadIdx.x;
    const int V = a.n_views;
    // LARGEST CLUSTERS FIRST (round 5): workgroup b renders the cluster of rank b / V by point count.  One workgroup owns a CU (~100 KB of
    // LDS) and a frame's ~340 workgroups run in two rounds on 256 CUs: in label order a 15 000-point wall that starts in the second round
    // ends the launch long after everything else; started first, the short ones fill in behind it.  The rank is recomputed by every
    // workgroup from the offsets (C <= a few hundred: C^2 / 1024 compares per thread) -- no scratch buffer, no extra launch; the crop a
    // workgroup writes is still cluster * V + view.
    __shared__ int s_cluster;
    {
        const int C = a.n_clusters, want = (int)blockIdx.x / V;
        if (C > 1024) { if (tid == 0) s_cluster = want; }          // thousands of clusters (valid_only off, dense scenes): label order, not C^2 compares per workgroup
        else
        for (int t = tid; t < C; t += RT) {
            const int sz = a.seg_off[t + 1] - a.seg_off[t];
            int rank = 0;
            for (int u = 0; u < C; ++u) {
                const int su = a.seg_off[u + 1] - a.seg_off[u];
                rank += (su > sz || (su == sz && u < t)) ? 1 : 0;
            }
            if (rank == want) s_cluster = t;
        }
        __syncthreads();
    }
    const int c = s_cluster, v = blockIdx.x % V;
    const int crop_id = c * V + v;
    const int p0 = a.seg_off[c], P = a.seg_off[c + 1] - p0;
    const float* pts = a.origin + (size_t)p0 * 3;
    float r[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) r[i] = a.view_rot[v * 9 + i];

    if (tid == 0) slice_mask = 0u;
    if (tid < OUT) {
        // F.interpolate(..., align_corners=True): src = (109/223) * dst in float32
        float scale = 109.0f / 223.0f;
        float src = scale * (float)tid;
        int i0 = (int)src;
        if (i0 > GO - 1) i0 = GO - 1;
        float l1 = src - (float)i0;
        t_i0[tid] = i0;
        t_l1[tid] = l1;
        t_l0[tid] = 1.0f - l1;
    }

    // ---- pass 1: per-view bounding box (mv_utils.py:101-104) --------------------------------
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < P; i += RT) {
        float x = pts[i * 3 + 0], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float pv = fmaf(z, r[6 + j], fmaf(y, r[3 + j], x * r[j]));
            mn[j] = fminf(mn[j], pv);
            mx[j] = fmaxf(mx[j], pv);
        }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        mn[j] = vg_wave_min(mn[j]);
        mx[j] = vg_wave_max(mx[j]);
    }
    int wid = tid >> 6, lane = tid & 63;
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            red[wid * 6 + j] = mn[j];
            red[wid * 6 + 3 + j] = mx[j];
        }
    }
    __syncthreads();
    if (tid == 0) {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int w = 0; w < RT / 64; ++w)
            for (int j = 0; j < 3; ++j) {
                lo[j] = fminf(lo[j], red[w * 6 + j]);
                hi[j] = fmaxf(hi[j], red[w * 6 + 3 + j]);
            }
        float pr = -INFINITY;
        for (int j = 0; j < 3; ++j) {
            bc[j] = (hi[j] + lo[j]) / 2.0f;
            pr = fmaxf(pr, hi[j] - lo[j]);
        }
        bc[3] = pr;
    }
    __syncthreads();
    const float pc[3] = {bc[0], bc[1], bc[2]};
    const float prange = bc[3];

    // which depth slices are occupied at all?
    unsigned int mymask = 0u;
    for (int i = tid; i < P; i += RT) {
        float x = pts[i * 3 + 0], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
        float pv0 = fmaf(z, r[6], fmaf(y, r[3], x * r[0]));
        float pv1 = fmaf(z, r[7], fmaf(y, r[4], x * r[1]));
        float pv2 = fmaf(z, r[8], fmaf(y, r[5], x * r[2]));
        int gx, gy, gz;
        float val;
        quantise_point(pv0, pv1, pv2, pc, prange, gx, gy, gz, val);
        if (gz >= 0 && gz < 8) mymask |= 1u << gz;
    }
    if (mymask) atomicOr(&slice_mask, mymask);
    __syncthreads();
    const unsigned int smask = slice_mask;

    float acc[ACC_PER_THREAD];
#pragma unroll
    for (int k = 0; k < ACC_PER_THREAD; ++k) acc[k] = 0.0f;

    // 3x3 Gaussian (mv_utils.py:204-220, sigma=3): float32 values as torch computes them
    const float g_c = a.lut[768 + 0], g_e = a.lut[768 + 1], g_m = a.lut[768 + 2];  // corner, edge, middle

    // Per slice only the footprint of its points is processed (round 5: the 110 x 110 passes over a slice that holds a car's side at one
    // depth -- a few thousand of the 12 100 pixels -- were most of the kernel's 206 us).  Raw slice S: nonzero only inside the bounding box
    // [r0, r1] x [c0, c1] of the cells its points fall into; row-pooled T: rows [r0 - 4, r1 + 4] (zero rows included, so nothing stale is
    // read), columns [c0 - 3, c1 + 1]; pooled image, written back into S with the RAW row stride: [r0 - 3, r1 + 1] x [c0 - 3, c1 + 1], which
    // covers the raw box for every index the convolution reads (< 110), everything else of S is still the zeroed slice; convolution output:
    // [r0 - 4, r1 + 2] x [c0 - 4, c1 + 2].  Outside those ranges the dense passes produced exact zeros and max(acc, 0) = acc: same bits.
    __shared__ int bb[4];            // r0, r1, c0, c1 of the current slice
    for (int d = 0; d < 8; ++d) {
        if (!((smask >> d) & 1u)) continue;   // empty slice: pool/conv give 0, acc >= 0 already
        for (int i = tid; i < GR * GR; i += RT) S[i] = 0.0f;
        if (tid == 0) { bb[0] = GR; bb[1] = -1; bb[2] = GR; bb[3] = -1; }
        __syncthreads();
        int r0 = GR, r1 = -1, c0 = GR, c1 = -1;
        for (int i = tid; i < P; i += RT) {
            float x = pts[i * 3 + 0], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
            float pv0 = fmaf(z, r[6], fmaf(y, r[3], x * r[0]));
            float pv1 = fmaf(z, r[7], fmaf(y, r[4], x * r[1]));
            float pv2 = fmaf(z, r[8], fmaf(y, r[5], x * r[2]));
            int gx, gy, gz;
            float val;
            quantise_point(pv0, pv1, pv2, pc, prange, gx, gy, gz, val);
            // grid[z][y][x] then permute(0,1,3,2): image row = x, column = y (mv_utils.py:120-125)
            if (gz == d) {
                atomicMax((int*)&S[gx * GR + gy], __float_as_int(val));
                r0 = min(r0, gx); r1 = max(r1, gx); c0 = min(c0, gy); c1 = max(c1, gy);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            r0 = min(r0, __shfl_xor(r0, o)); r1 = max(r1, __shfl_xor(r1, o));
            c0 = min(c0, __shfl_xor(c0, o)); c1 = max(c1, __shfl_xor(c1, o));
        }
        if (lane == 0 && r1 >= 0) { atomicMin(&bb[0], r0); atomicMax(&bb[1], r1); atomicMin(&bb[2], c0); atomicMax(&bb[3], c1); }
        __syncthreads();
        r0 = bb[0]; r1 = bb[1]; c0 = bb[2]; c1 = bb[3];
        const int tj0 = max(c0 - 3, 0), tj1 = min(c1 + 1, GO - 1), tw = tj1 - tj0 + 1;          // columns of T and of the pooled image
        const int ti0 = max(r0 - 4, 0), ti1 = min(r1 + 4, GR - 1);                                  // rows of T
        // MaxPool3d (1,5,5) pad (0,1,1): window [i-1, i+3] x [j-1, j+3]; separable.
        for (int q = tid; q < (ti1 - ti0 + 1) * tw; q += RT) {
            const int di = q / tw;
            const int i = ti0 + di, j = tj0 + (q - di * tw);
            const float* row = S + i * GR;
            float m = row[j];                       // j-1+1 .. (j in [0,110) -> cols j-1..j+3)
            if (j >= 1) m = fmaxf(m, row[j - 1]);
            m = fmaxf(m, row[j + 1]);
            m = fmaxf(m, row[j + 2]);
            if (j + 3 < GR) m = fmaxf(m, row[j + 3]);
            T[i * GO + j] = m;
        }
        __syncthreads();
        const int pi0 = max(r0 - 3, 0), pi1 = min(r1 + 1, GO - 1);                                  // rows of the pooled image
        for (int q = tid; q < (pi1 - pi0 + 1) * tw; q += RT) {
            const int di = q / tw;
            const int i = pi0 + di, j = tj0 + (q - di * tw);
            float m = T[i * GO + j];
            if (i >= 1) m = fmaxf(m, T[(i - 1) * GO + j]);
            m = fmaxf(m, T[(i + 1) * GO + j]);
            m = fmaxf(m, T[(i + 2) * GO + j]);
            if (i + 3 < GR) m = fmaxf(m, T[(i + 3) * GO + j]);
            S[i * GR + j] = m;   // pooled, raw row stride
        }
        __syncthreads();
        // Conv3d (1,3,3) zero pad, FMA chain in row-major tap order; running max over depth.
        // (the thread id is made opaque per slice: left visible, the compiler hoists the 12 pixels' row / column / border terms out of
        // the slice loop and keeps them live across the whole kernel -- 46 spilled registers, their traffic doubled the bytes written)
        int tq = tid;
        asm volatile("" : "+v"(tq));
        const int ci0 = max(r0 - 4, 0), ci1 = min(r1 + 2, GO - 1), cj0 = max(c0 - 4, 0), cj1 = min(c1 + 2, GO - 1);
#pragma unroll
        for (int k = 0; k < ACC_PER_THREAD; ++k) {
            int q = tq + k * RT;
            if (q < GO * GO) {
                int i = q / GO, j = q - i * GO;
                if (i < ci0 || i > ci1 || j < cj0 || j > cj1) continue;      // the dense pass gives exactly 0 here
                float s[9];
#pragma unroll
                for (int di = 0; di < 3; ++di)
#pragma unroll
                    for (int dj = 0; dj < 3; ++dj) {
                        int ii = i + di - 1, jj = j + dj - 1;
                        s[di * 3 + dj] = (ii >= 0 && ii < GO && jj >= 0 && jj < GO) ? S[ii * GR + jj] : 0.0f;
                    }
                float o = s[0] * g_c;
                o = fmaf(s[1], g_e, o);
                o = fmaf(s[2], g_c, o);
                o = fmaf(s[3], g_e, o);
                o = fmaf(s[4], g_m, o);
                o = fmaf(s[5], g_e, o);
                o = fmaf(s[6], g_c, o);
                o = fmaf(s[7], g_e, o);
                o = fmaf(s[8], g_c, o);
                acc[k] = fmaxf(acc[k], o);
            }
        }
        __syncthreads();
    }

    // ---- normalise: img = 1 - img / max(img)  (mv_utils.py:34-35) ---------------------------
    float m = 0.0f;
#pragma unroll
    for (int k = 0; k < ACC_PER_THREAD; ++k) m = fmaxf(m, acc[k]);
    m = vg_wave_max(m);
    if (lane == 0) red[wid] = m;
    __syncthreads();
    if (tid == 0) {
        float mm = 0.0f;
        for (int w = 0; w < RT / 64; ++w) mm = fmaxf(mm, red[w]);
        bc[4] = mm;
    }
    __syncthreads();
    const float imax = bc[4];
#pragma unroll
    for (int k = 0; k < ACC_PER_THREAD; ++k) {
        int q = tid + k * RT;
        if (q < GO * GO) T[q] = 1.0f - acc[k] / imax;
    }
    __syncthreads();
    if (a.out_kind == 3) {   // raw get_img() image, one channel: f32 [n,110,110]
        float* of = (float*)a.out + (size_t)crop_id * GO * GO;
        for (int q = tid; q < GO * GO; q += RT) of[q] = T[q];
        return;
    }

    // ---- D5 + D6: bilinear 110 -> 224, H<->W swap, uint8 truncation, CLIP normalise ---------
    // T_out[ch][i][j] = norm_ch( uint8( 255 * interp[h = j][w = i] ) )
    const size_t crop = (size_t)crop_id;
    for (int q4 = tid; q4 < OUT * OUT / 4; q4 += RT) {
        int i = q4 / (OUT / 4);
        int j0 = (q4 - i * (OUT / 4)) * 4;
        int w0 = t_i0[i];
        int w1 = w0 + (w0 < GO - 1 ? 1 : 0);
        float lw0 = t_l0[i], lw1 = t_l1[i];
        unsigned char u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int j = j0 + e;
            int h0 = t_i0[j];
            int h1 = h0 + (h0 < GO - 1 ? 1 : 0);
            float lh0 = t_l0[j], lh1 = t_l1[j];
            float p00 = T[h0 * GO + w0], p01 = T[h0 * GO + w1];
            float p10 = T[h1 * GO + w0], p11 = T[h1 * GO + w1];
            float t0 = fmaf(p00, lw0, p01 * lw1);
            float t1 = fmaf(p10, lw0, p11 * lw1);
            float o = fmaf(t0, lh0, t1 * lh1);
            float s255 = o * 255.0f;
            int iv = (int)s255;              // np.uint8(): truncation
            iv = iv < 0 ? 0 : (iv > 255 ? 255 : iv);
            u[e] = (unsigned char)iv;
        }
        if (a.out_kind >= 4) {
            // patch rows: stage the quantised image in LDS (the 112 x 112 float slice buffer is free now and holds exactly
            // 224 x 224 bytes); the stores happen below in OUTPUT order so that they leave as whole 128-byte lines
            *(unsigned int*)((unsigned char*)S + i * OUT + j0) = (unsigned int)u[0] | ((unsigned int)u[1] << 8) | ((unsigned int)u[2] << 16) |
                                                                 ((unsigned int)u[3] << 24);
        } else if (a.out_kind == 0) {
            unsigned char* o8 = (unsigned char*)a.out + (crop * OUT * OUT + (size_t)i * OUT + j0) * 3;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o8[e * 3 + 0] = u[e];
                o8[e * 3 + 1] = u[e];
                o8[e * 3 + 2] = u[e];
            }
        } else if (a.out_kind == 1) {
            float* of = (float*)a.out + crop * 3 * OUT * OUT + (size_t)i * OUT + j0;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float4 f4 = make_float4(a.lut[ch * 256 + u[0]], a.lut[ch * 256 + u[1]], a.lut[ch * 256 + u[2]],
                                        a.lut[ch * 256 + u[3]]);
                *(float4*)(of + (size_t)ch * OUT * OUT) = f4;
            }
        } else {
            __half* oh = (__half*)a.out + crop * 3 * OUT * OUT + (size_t)i * OUT + j0;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                __half2 h01 = __floats2half2_rn(a.lut[ch * 256 + u[0]], a.lut[ch * 256 + u[1]]);
                __half2 h23 = __floats2half2_rn(a.lut[ch * 256 + u[2]], a.lut[ch * 256 + u[3]]);
                uint2 pk;
                pk.x = *(unsigned int*)&h01;
                pk.y = *(unsigned int*)&h23;
                *(uint2*)(oh + (size_t)ch * OUT * OUT) = pk;
            }
        }
    }
    if (a.out_kind == 4) {
        // fp16 patch rows [crop*196 + py*14 + px][ch*256 + pi*16 + pj] (the conv1 GEMM's A operand).  Work item = 4 pixels
        // = 8 output bytes, enumerated in output order: 192 items per patch row (3 channels x 16 pixel rows x 4), so the 64
        // lanes of a wave write 512 contiguous bytes -- four whole 128-byte lines -- instead of 32-byte pieces of 16 lines
        // (written per pixel row, the pieces reached HBM as partial lines: 200 MB moved for 98 MB of rows, r01 PMC).
        __syncthreads();
        const unsigned char* U = (const unsigned char*)S;
        __half* ob = (__half*)a.out + crop * 196 * 768;
        for (int q = tid; q < 196 * 192; q += RT) {
            const int p = q / 192, rem = q - p * 192;
            const int ch = rem >> 6, pi = (rem >> 2) & 15, pj4 = rem & 3;
            const int py = p / 14, px = p - py * 14;
            const unsigned int u4 = *(const unsigned int*)(U + (py * 16 + pi) * OUT + px * 16 + pj4 * 4);
            const float* lut = a.lut + ch * 256;
            __half2 h01 = __floats2half2_rn(lut[u4 & 255u], lut[(u4 >> 8) & 255u]);
            __half2 h23 = __floats2half2_rn(lut[(u4 >> 16) & 255u], lut[u4 >> 24]);
            uint2 pk;
            pk.x = *(unsigned int*)&h01;
            pk.y = *(unsigned int*)&h23;
            *(uint2*)(ob + (size_t)p * 768 + rem * 4) = pk;
        }
    }
    if (a.out_kind == 5) {
        // single-channel patch rows: 64 items of 4 pixels per patch row, so one wave instruction writes one whole 512-byte row
        __syncthreads();
        const unsigned char* U = (const unsigned char*)S;
        __half* ob = (__half*)a.out + crop * 196 * 256;
        for (int q = tid; q < 196 * 64; q += RT) {
            const int p = q >> 6, rem = q & 63;
            const int pi = rem >> 2, pj4 = rem & 3;
            const int py = p / 14, px = p - py * 14;
            const unsigned int u4 = *(const unsigned int*)(U + (py * 16 + pi) * OUT + px * 16 + pj4 * 4);
            const float sc = 0.00390625f;                               // 2^-8: level / 256 is exact in fp16
            __half2 h01 = __floats2half2_rn((float)(u4 & 255u) * sc, (float)((u4 >> 8) & 255u) * sc);
            __half2 h23 = __floats2half2_rn((float)((u4 >> 16) & 255u) * sc, (float)(u4 >> 24) * sc);
            uint2 pk;
            pk.x = *(unsigned int*)&h01;
            pk.y = *(unsigned int*)&h23;
            *(uint2*)(ob + (size_t)p * 256 + rem * 4) = pk;
        }
    }
}

// ---------------------------------------------------------------------------------------------
extern "C" {

int vg_gather_ego(const float* d_points, int stride, const int32_t* d_index, int n, const double* d_T4x4,
                  float* d_ego, void* stream) {
    if (n <= 0) return VG_OK;
    if (!d_points || !d_T4x4 || !d_ego || stride < 3) return VG_ERR_ARG;
    hipLaunchKernelGGL(k_gather_ego, dim3(vg_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, d_points, stride,
                       d_index, n, d_T4x4, d_ego);
    VG_LAUNCH_CHECK();
    return VG_OK;
}

int vg_cluster_median(const float* d_ego, const int32_t* d_seg_off, int n_clusters, float* d_median,
                      double* d_rot, void* stream) {
    if (n_clusters <= 0) return VG_OK;
    if (!d_ego || !d_seg_off || !d_median || !d_rot) return VG_ERR_ARG;
    hipLaunchKernelGGL(k_cluster_median, dim3(n_clusters), dim3(768), 0, (hipStream_t)stream, d_ego, d_seg_off,
                       d_median, d_rot);
    VG_LAUNCH_CHECK();
    return VG_OK;
}

int vg_cluster_rot(const float* d_angle, int n_clusters, double* d_rot, void* stream) {
    if (n_clusters <= 0) return VG_OK;
    if (!d_angle || !d_rot) return VG_ERR_ARG;
    hipLaunchKernelGGL(k_cluster_rot, dim3(vg_div_up(n_clusters, 64)), dim3(64), 0, (hipStream_t)stream, d_angle, n_clusters, d_rot);
    VG_LAUNCH_CHECK();
    return VG_OK;
}

int vg_to_origin(const float* d_ego, const int32_t* d_point_cluster, int n, const float* d_median,
                 const double* d_rot, const double* d_Timg3x3, float* d_origin, void* stream) {
    if (n <= 0) return VG_OK;
    if (!d_ego || !d_point_cluster || !d_median || !d_rot || !d_Timg3x3 || !d_origin) return VG_ERR_ARG;
    hipLaunchKernelGGL(k_to_origin, dim3(vg_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, d_ego,
                       d_point_cluster, n, d_median, d_rot, d_Timg3x3, d_origin);
    VG_LAUNCH_CHECK();
    return VG_OK;
}

int vg_render_crops(const float* d_origin, const int32_t* d_seg_off, int n_clusters, const float* d_view_rot,
                    int n_views, const float* d_lut, void* d_out, int out_kind, void* stream) {
    if (n_clusters <= 0 || n_views <= 0) return VG_OK;
    if (!d_origin || !d_seg_off || !d_view_rot || !d_lut || !d_out || out_kind < 0 || out_kind > 5)
        return VG_ERR_ARG;
    const size_t lds_bytes = (size_t)(GR * GR + GR * GO) * sizeof(float);
    VG_MAX_DYNAMIC_LDS(k_render, lds_bytes);
    RenderArgs a;
    a.origin = d_origin;
    a.seg_off = d_seg_off;
    a.view_rot = d_view_rot;
    a.lut = d_lut;
    a.out = d_out;
    a.n_views = n_views;
    a.n_clusters = n_clusters;
    a.out_kind = out_kind;
    hipLaunchKernelGGL(k_render, dim3(n_clusters * n_views), dim3(RT), lds_bytes, (hipStream_t)stream, a);
    VG_LAUNCH_CHECK();
    return VG_OK;
}

}  // extern "C"
