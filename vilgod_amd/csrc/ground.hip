// Ground segmentation for gfx950: a from-scratch GPU formulation of the
// algorithm in third_party/patchwork-plusplus/patchworkpp/src/patchworkpp.cpp (Patchwork++), called by
// the reference at src/vilgod/zero_shot_detector.py:129-151 through src/utils/pointcloud_utils.py:49-56.
//
// The reference walks 504 patches sequentially on one CPU thread.  Here:
//   k_pw_classify   one thread per point: z-offset, reflected-noise removal (:378-401), concentric-zone
//                   binning (:579-623) -> patch id; LDS-privatised histogram of patch sizes
//   k_pw_offsets    504-entry exclusive scan
//   k_pw_scatter    (z-key, index) 64-bit keys grouped by patch
//   k_pw_patch      ONE WORKGROUP PER PATCH: bitonic sort of the keys in LDS (= std::sort by z, :200,
//                   ties by index), then R-VPF / R-GPF (:468-550) as predicate-masked block reductions
//                   over the sorted points held in LDS, 3x3 eigen-solve per thread
//   k_pw_decide     one lane per ring: GLE decision tree (:218-283), TGR (:403-465); then the adaptive
//                   threshold update (:339-376) with the per-sequence state kept in HBM
//   k_pw_finalize   per point: ground = in its patch's plane inliers AND patch accepted
//
// Numeric model (identical to oracle/patchworkpp_oracle.cpp, see DESIGN.md "ground numerics"): float64 sums
// of exact float32 products in the fixed order SUM256, one-pass covariance, cyclic Jacobi (8 sweeps) in float64,
// plane/mean/singular values rounded to float32; compiled with -ffp-contract=off.
#include "common.h"
#include "vilgod_hip.h"
#include <float.h>
#include <math.h>
#include <string.h>

#define PW_MAX_PATCHES 1024
#define PW_STORE_CAP 2048            // ring buffers for update_elevation_/update_flatness_ (1000 + <=54 per frame)
#define PW_LDS_POINTS 4096           // patches up to this size are processed entirely in LDS
#define PW_BIG_LDS_POINTS 16384      // larger patches up to this size: keys sorted in LDS by k_pw_sort_big, the rest of the patch from global memory
#define PW_T 256

struct PwGeom {                      // derived in the constructor, patchworkpp.h:116-131
    double min_ranges[4], ring_sizes[4], sector_sizes[4];
    int patch_base[5];
    int n_patches;
};

struct PwState {                     // everything Patchwork++ carries from frame to frame
    double sensor_height;
    double elevation_thr[4], flatness_thr[4];
    int elev_head[4], elev_cnt[4], flat_head[4], flat_cnt[4];
    double elev[4][PW_STORE_CAP], flat[4][PW_STORE_CAP];
};

struct PwPatchRec {
    int n, n_ground;
    float normal[3], mean[3], sv[3];
    int decision;                    // 0 non-ground, 1 ground, 2 TGR candidate (resolved in k_pw_decide)
};

struct vg_ground {
    vg_ground_params p;
    PwGeom g;
    int max_points;
    vg_ground_params* d_p;
    PwGeom* d_g;
    PwState* d_state;
    int* d_patch_id;        // [max_points]
    int* d_count;           // [PW_MAX_PATCHES] sizes, then cursors
    int* d_offset;          // [PW_MAX_PATCHES+1]
    int* d_cursor;          // [PW_MAX_PATCHES]
    unsigned long long* d_keys;  // [max_points]
    unsigned char* d_inlier;     // [max_points]
    PwPatchRec* d_rec;      // [PW_MAX_PATCHES]
};

// ---------------------------------------------------------------------------------------------
__global__ void k_pw_classify(const float* __restrict__ pts, int n, int stride, double z_offset,
                              const vg_ground_params* __restrict__ P, const PwGeom* __restrict__ G,
                              const PwState* __restrict__ S, int* __restrict__ patch_id, int* __restrict__ count,
                              unsigned char* __restrict__ inlier) {
    __shared__ int hist[PW_MAX_PATCHES];
    for (int b = threadIdx.x; b < PW_MAX_PATCHES; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        inlier[i] = 0;                    // (was a memset in front of the pass: one launch less on the frame chain)
        const float x = pts[(size_t)i * stride], y = pts[(size_t)i * stride + 1];
        // pointcloud_utils.py:50-51: float64 subtraction, then Eigen::MatrixXf (float32)
        const float z = (float)((double)pts[(size_t)i * stride + 2] - z_offset);
        const float inten = pts[(size_t)i * stride + 3];
        int pid = -1;
        bool noise = false;
        if (P->enable_RNR) {
            float rr = x * x + y * y;
            // patchworkpp.cpp:388: `sqrt` of a FLOAT expression under `using namespace std` (patchworkpp.h:10) is the float
            // overload; (float)sqrt((double)rr) is that correctly rounded float root (53 >= 2*24+2 bits: no double rounding)
            double r = (double)(float)sqrt((double)rr);
            double zd = z;
            double ang = atan2(zd, r) * 180 / M_PI;
            noise = ang < P->RNR_ver_angle_thr && zd < -S->sensor_height - 0.8 && (double)inten < P->RNR_intensity_thr;
        }
        if (!noise && z != FLT_MIN) {
            double xd = x, yd = y;
            double r = sqrt(xd * xd + yd * yd);
            if (r <= P->max_range && r > P->min_range) {
                double theta = atan2(yd, xd);
                if (!(theta > 0)) theta = 2 * M_PI + theta;
                int zone = r < G->min_ranges[1] ? 0 : (r < G->min_ranges[2] ? 1 : (r < G->min_ranges[3] ? 2 : 3));
                int ring = min((int)((r - G->min_ranges[zone]) / G->ring_sizes[zone]), P->num_rings_each_zone[zone] - 1);
                int sector = min((int)(theta / G->sector_sizes[zone]), P->num_sectors_each_zone[zone] - 1);
                pid = G->patch_base[zone] + ring * P->num_sectors_each_zone[zone] + sector;
            }
        }
        patch_id[i] = pid;
        if (pid >= 0) atomicAdd(&hist[pid], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < PW_MAX_PATCHES; b += blockDim.x)
        if (hist[b]) atomicAdd(&count[b], hist[b]);
}

__global__ __launch_bounds__(64) void k_pw_offsets(int* __restrict__ count, int* __restrict__ offset, int* __restrict__ cursor,
                                                   int n_patches) {
    // exclusive scan of the patch sizes by one wave: lane l takes a run of patches, the runs' totals are scanned across the lanes
    const int lane = threadIdx.x, per = (n_patches + 63) / 64;
    const int lo = min(lane * per, n_patches), hi = min(lo + per, n_patches);
    int run = 0;
    for (int p = lo; p < hi; ++p) run += count[p];
    int incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    int acc = incl - run;
    for (int p = lo; p < hi; ++p) {
        offset[p] = acc;
        cursor[p] = 0;
        acc += count[p];
        count[p] = 0;                     // ready for the next pass (the histogram is zero at creation; was a memset per pass)
    }
    if (lane == 63) offset[n_patches] = incl;
}

__global__ void k_pw_scatter(const float* __restrict__ pts, int n, int stride, double z_offset,
                             const int* __restrict__ patch_id, const int* __restrict__ offset, int* __restrict__ cursor,
                             unsigned long long* __restrict__ keys) {
    // slots are reserved per workgroup: rank inside the workgroup from an LDS counter, one global atomic per (workgroup, patch).
    // (One global atomic per point serialised on the cursors of the dense near-range patches: 197 us per scan.)  The order inside a
    // patch is arbitrary either way; k_pw_patch / k_pw_sort_big sort the (unique) keys.
    __shared__ int hist[PW_MAX_PATCHES];
    for (int b = threadIdx.x; b < PW_MAX_PATCHES; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int pid = i < n ? patch_id[i] : -1;
    int local = 0;
    if (pid >= 0) local = atomicAdd(&hist[pid], 1);
    __syncthreads();
    for (int b = threadIdx.x; b < PW_MAX_PATCHES; b += blockDim.x)
        if (hist[b]) hist[b] = atomicAdd(&cursor[b], hist[b]);
    __syncthreads();
    if (pid < 0) return;
    const float z = (float)((double)pts[(size_t)i * stride + 2] - z_offset);
    keys[offset[pid] + hist[pid] + local] = ((unsigned long long)vg_fkey(z) << 32) | (unsigned int)i;
}

// ---------------------------------------------------------------------------------------------
__device__ void pw_eig3(const double Ain[3][3], double w[3], double V[3][3]) {
    double A[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            A[i][j] = Ain[i][j];
            V[i][j] = (i == j) ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 8; ++sweep) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int p = (r == 2) ? 1 : 0, q = (r == 0) ? 1 : 2;
            const double apq = A[p][q];
            if (!(fabs(apq) > 1e-300)) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            const double app = A[p][p], aqq = A[q][q];
            A[p][p] = app - t * apq;
            A[q][q] = aqq + t * apq;
            A[p][q] = A[q][p] = 0.0;
            const int k = 3 - p - q;
            const double akp = A[k][p], akq = A[k][q];
            A[k][p] = A[p][k] = c * akp - s * akq;
            A[k][q] = A[q][k] = s * akp + c * akq;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double vip = V[i][p], viq = V[i][q];
                V[i][p] = c * vip - s * viq;
                V[i][q] = s * vip + c * viq;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) w[i] = A[i][i];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int i = (r == 2) ? 1 : 0, j = (r == 0) ? 1 : 2;
        if (w[i] < w[j]) {
            double tw = w[i]; w[i] = w[j]; w[j] = tw;
#pragma unroll
            for (int m = 0; m < 3; ++m) { double tv = V[m][i]; V[m][i] = V[m][j]; V[m][j] = tv; }
        }
    }
}

struct PwPlane {
    float normal[3], mean[3], sv[3];
    double d;
};

__device__ void pw_estimate_plane(const double S[9], int n, PwPlane& pl) {
    if (n == 0) return;   // patchworkpp.cpp:50 -- stale plane survives
    const double dn = (double)n, dn1 = (double)(n - 1);
    const double mx = S[0] / dn, my = S[1] / dn, mz = S[2] / dn;
    double C[3][3];
    C[0][0] = (double)(float)((S[3] - S[0] * mx) / dn1);
    C[0][1] = (double)(float)((S[4] - S[0] * my) / dn1);
    C[0][2] = (double)(float)((S[5] - S[0] * mz) / dn1);
    C[1][1] = (double)(float)((S[6] - S[1] * my) / dn1);
    C[1][2] = (double)(float)((S[7] - S[1] * mz) / dn1);
    C[2][2] = (double)(float)((S[8] - S[2] * mz) / dn1);
    C[1][0] = C[0][1]; C[2][0] = C[0][2]; C[2][1] = C[1][2];
    double w[3], V[3][3];
    pw_eig3(C, w, V);
    pl.mean[0] = (float)mx; pl.mean[1] = (float)my; pl.mean[2] = (float)mz;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        pl.sv[i] = (float)fabs(w[i]);
        pl.normal[i] = (float)V[i][2];
    }
    if (pl.normal[2] < 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) pl.normal[i] = -pl.normal[i];
    }
    float dot = (pl.normal[0] * pl.mean[0] + pl.normal[1] * pl.mean[1]) + pl.normal[2] * pl.mean[2];
    pl.d = -(double)dot;
}

__device__ __forceinline__ double pw_plane_dist(const PwPlane& pl, float x, float y, float z) {
    float f = (pl.normal[0] * x + pl.normal[1] * y) + pl.normal[2] * z;
    return (double)f + pl.d;
}

// block-wide SUM256 reduction of 9 doubles + a count; result broadcast to every thread
__device__ void pw_block_sums(double v[9], int cnt, double out[9], int& out_cnt, double* red /*[4][10]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        double x = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x = x + __shfl_xor(x, o);
        v[k] = x;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) red[wave * 10 + k] = v[k];
        red[wave * 10 + 9] = (double)cnt;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 9; ++k) out[k] = ((red[k] + red[10 + k]) + red[20 + k]) + red[30 + k];
    out_cnt = (int)(red[9] + red[19] + red[29] + red[39]);
}

// normalised bitonic network (all comparators ascending) -> works for any n with virtual +inf padding.
// A thread walks COMPARATORS (np2 / 2 per stage), not elements: comparator c of the mirror step of block size k pairs
// i = (c / (k/2)) k + c % (k/2) with the block's mirrored position, comparator c of a step j pairs i = (c / j) 2j + c % j with i + j;
// i grows with c, so a thread stops at the first i >= n.
template <typename KP>
__device__ void pw_bitonic_sort(KP keys, int n) {
    int np2 = 1, lg = 0;
    while (np2 < n) { np2 <<= 1; ++lg; }
    const int half = np2 >> 1;
    for (int lk = 1; lk <= lg; ++lk) {                     // k = 1 << lk
        const int hk = 1 << (lk - 1);                      // k / 2
        for (int c = threadIdx.x; c < half; c += blockDim.x) {
            const int o = c & (hk - 1), base = (c >> (lk - 1)) << lk;
            const int i = base + o, p = base + ((1 << lk) - 1 - o);
            if (i >= n) break;
            if (p < n) {
                unsigned long long a = keys[i], b = keys[p];
                if (a > b) { keys[i] = b; keys[p] = a; }
            }
        }
        __syncthreads();
        for (int lj = lk - 2; lj >= 0; --lj) {             // j = 1 << lj
            const int j = 1 << lj;
            for (int c = threadIdx.x; c < half; c += blockDim.x) {
                const int i = ((c >> lj) << (lj + 1)) + (c & (j - 1)), p = i + j;
                if (i >= n) break;
                if (p < n) {
                    unsigned long long a = keys[i], b = keys[p];
                    if (a > b) { keys[i] = b; keys[p] = a; }
                }
            }
            __syncthreads();
        }
    }
}

template <bool BIG>
__device__ void pw_patch_body(const float* __restrict__ pts, int stride, const vg_ground_params* __restrict__ P,
                              const PwState* __restrict__ S, int zone, int n, unsigned long long* gkeys,
                              unsigned char* __restrict__ inlier, PwPatchRec& rec, unsigned long long* lkeys,
                              float* lx, float* ly, unsigned char* lalive, double* red, double* bc) {
    const int tid = threadIdx.x;
    // ---- sort by (z, index) ----
    if (!BIG) {
        for (int i = tid; i < n; i += PW_T) lkeys[i] = gkeys[i];
        __syncthreads();
        pw_bitonic_sort(lkeys, n);
        for (int i = tid; i < n; i += PW_T) {
            unsigned int idx = (unsigned int)(lkeys[i] & 0xFFFFFFFFull);
            lx[i] = pts[(size_t)idx * stride];
            ly[i] = pts[(size_t)idx * stride + 1];
            lalive[i] = 1;
        }
    } else {
        if (n > PW_BIG_LDS_POINTS) pw_bitonic_sort(gkeys, n);      // up to PW_BIG_LDS_POINTS k_pw_sort_big has sorted the keys already
        __threadfence_block();
        // alive flags of big patches live in the inlier array (indexed by ORIGINAL point index): 2 = alive
        for (int i = tid; i < n; i += PW_T) inlier[(unsigned int)(gkeys[i] & 0xFFFFFFFFull)] = 2;
    }
    __syncthreads();
#define PW_KEY(i) (BIG ? gkeys[i] : lkeys[i])
#define PW_IDX(i) ((unsigned int)(PW_KEY(i) & 0xFFFFFFFFull))
#define PW_Z(i) vg_fkey_inv((uint32_t)(PW_KEY(i) >> 32))
#define PW_X(i) (BIG ? pts[(size_t)PW_IDX(i) * stride] : lx[i])
#define PW_Y(i) (BIG ? pts[(size_t)PW_IDX(i) * stride + 1] : ly[i])
#define PW_ALIVE(i) (BIG ? (inlier[PW_IDX(i)] == 2) : (lalive[i] != 0))
#define PW_KILL(i)                      \
    do {                                \
        if (BIG) inlier[PW_IDX(i)] = 0; \
        else lalive[i] = 0;             \
    } while (0)

    PwPlane pl;
#pragma unroll
    for (int i = 0; i < 3; ++i) pl.normal[i] = pl.mean[i] = pl.sv[i] = 0.f;
    pl.d = 0.0;
    const double sensor_height = S->sensor_height;

    auto seed_threshold = [&](double th) -> double {   // extract_initial_seeds, :78-150
        __syncthreads();
        if (tid == 0) {
            double sum = 0;
            int cnt = 0, i = 0;
            if (zone == 0) {
                for (; i < n; ++i) {
                    if (!PW_ALIVE(i)) continue;
                    if ((double)PW_Z(i) < P->adaptive_seed_selection_margin * sensor_height) continue;
                    break;
                }
            }
            for (; i < n && cnt < P->num_lpr; ++i) {
                if (!PW_ALIVE(i)) continue;
                sum += (double)PW_Z(i);
                cnt++;
            }
            double lpr = cnt != 0 ? sum / cnt : 0;
            bc[0] = lpr + th;
        }
        __syncthreads();
        return bc[0];
    };
    // estimate_plane over {alive, pred}: pred 0: z < thr;  pred 1: signed plane distance < th_dist
    auto fit = [&](int pred, double thr, bool mark_inliers) {
        double v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = 0.0;
        int cnt = 0;
        const PwPlane cur = pl;
        for (int i = tid; i < n; i += PW_T) {
            if (!PW_ALIVE(i)) continue;
            const float x = PW_X(i), y = PW_Y(i), z = PW_Z(i);
            bool take = pred == 0 ? ((double)z < thr) : (pw_plane_dist(cur, x, y, z) < thr);
            if (take) {
                double X = x, Y = y, Z = z;
                v[0] += X; v[1] += Y; v[2] += Z;
                v[3] += X * X; v[4] += X * Y; v[5] += X * Z; v[6] += Y * Y; v[7] += Y * Z; v[8] += Z * Z;
                cnt++;
                if (mark_inliers) inlier[PW_IDX(i)] = 1;
            }
        }
        double Sm[9];
        int tot;
        pw_block_sums(v, cnt, Sm, tot, red);
        pw_estimate_plane(Sm, tot, pl);
        return tot;
    };

    // ---- R-VPF, :478-509 ----
    if (P->enable_RVPF) {
        for (int it = 0; it < P->num_iter; ++it) {
            double thr = seed_threshold(P->th_seeds_v);
            fit(0, thr, false);
            if (zone == 0 && (double)pl.normal[2] < P->uprightness_thr) {
                for (int i = tid; i < n; i += PW_T)
                    if (PW_ALIVE(i) && fabs(pw_plane_dist(pl, PW_X(i), PW_Y(i), PW_Z(i))) < P->th_dist_v) PW_KILL(i);
            } else
                break;
        }
    }
    // ---- R-GPF, :511-544 ----
    {
        double thr = seed_threshold(P->th_seeds);
        fit(0, thr, false);
    }
    int n_ground = 0;
    for (int it = 0; it < P->num_iter; ++it) {
        __syncthreads();
        n_ground = fit(1, P->th_dist, it == P->num_iter - 1);
    }
    if (BIG) {   // surviving "alive" marks (2) of non-inliers must not leak into the inlier flags
        __syncthreads();
        for (int i = tid; i < n; i += PW_T)
            if (inlier[PW_IDX(i)] == 2) inlier[PW_IDX(i)] = 0;
    }
    rec.n_ground = n_ground;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        rec.normal[i] = pl.normal[i];
        rec.mean[i] = pl.mean[i];
        rec.sv[i] = pl.sv[i];
    }
#undef PW_KEY
#undef PW_IDX
#undef PW_Z
#undef PW_X
#undef PW_Y
#undef PW_ALIVE
#undef PW_KILL
}

// Patches too large for k_pw_patch's LDS image (a handful of near-range patches per scan: 4 097 ... 9 646 points on the benchmark
// frames) used to be sorted in global memory by their 256-thread workgroup -- 105 compare-exchange stages over L2 for 9 646 keys,
// 0.43 ms of a 1.15 ms ground pass.  Their keys alone fit the LDS of one CU: 1024 threads sort them there first (keys are unique, so
// the order is the same one).
__global__ __launch_bounds__(1024) void k_pw_sort_big(const vg_ground_params* __restrict__ P, const int* __restrict__ offset,
                                                      unsigned long long* __restrict__ keys) {
    extern __shared__ unsigned long long pw_sk[];
    const int pid = blockIdx.x;
    const int o = offset[pid], n = offset[pid + 1] - o;
    if (n <= PW_LDS_POINTS || n > PW_BIG_LDS_POINTS || n < P->num_min_pts) return;      // workgroup-uniform
    for (int i = threadIdx.x; i < n; i += blockDim.x) pw_sk[i] = keys[o + i];
    __syncthreads();
    pw_bitonic_sort(pw_sk, n);
    for (int i = threadIdx.x; i < n; i += blockDim.x) keys[o + i] = pw_sk[i];
}

__global__ __launch_bounds__(PW_T) void k_pw_patch(const float* __restrict__ pts, int stride,
                                                   const vg_ground_params* __restrict__ P, const PwGeom* __restrict__ G,
                                                   const PwState* __restrict__ S, const int* __restrict__ offset,
                                                   unsigned long long* __restrict__ keys, unsigned char* __restrict__ inlier,
                                                   PwPatchRec* __restrict__ recs) {
    __shared__ unsigned long long lkeys[PW_LDS_POINTS];
    __shared__ float lx[PW_LDS_POINTS], ly[PW_LDS_POINTS];
    __shared__ unsigned char lalive[PW_LDS_POINTS];
    __shared__ double red[40];
    __shared__ double bc[2];
    const int pid = blockIdx.x;
    const int o = offset[pid], n = offset[pid + 1] - o;
    PwPatchRec rec;
    rec.n = n;
    rec.n_ground = 0;
    rec.decision = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) rec.normal[i] = rec.mean[i] = rec.sv[i] = 0.f;
    if (n >= P->num_min_pts) {   // :192-196
        int zone = 0;
        while (zone < 3 && pid >= G->patch_base[zone + 1]) zone++;
        rec.decision = -1;       // to be decided by k_pw_decide
        if (n <= PW_LDS_POINTS)
            pw_patch_body<false>(pts, stride, P, S, zone, n, keys + o, inlier, rec, lkeys, lx, ly, lalive, red, bc);
        else
            pw_patch_body<true>(pts, stride, P, S, zone, n, keys + o, inlier, rec, lkeys, lx, ly, lalive, red, bc);
    }
    if (threadIdx.x == 0) recs[pid] = rec;
}

// ---------------------------------------------------------------------------------------------
// lane r < n_rings: sequential GLE + TGR for ring r (concentric index r).  Then lanes 0..3 / 8..11 update thresholds.  (256 threads:
// the other three waves only help with the copies into LDS.)
__global__ __launch_bounds__(256) void k_pw_decide(vg_ground_params* __restrict__ P, const PwGeom* __restrict__ G,
                                                  PwState* __restrict__ S, PwPatchRec* __restrict__ recs) {
    const int lane = threadIdx.x;
    __shared__ double sh_rf[4][64];          // per near ring: the flatness values it appends to `ringwise_flatness`
    __shared__ int sh_nrf[4], sh_ncand[4];
    // one 48 KB buffer, two uses: the patch records while the rings are decided (each ring's lane walks its <= 54 sectors one
    // after the other: from LDS, not through ~100 dependent reads of global memory), then the threshold stores (below)
    __shared__ double raw[PW_MAX_PATCHES * sizeof(PwPatchRec) / 8];
    PwPatchRec* const lrec = reinterpret_cast<PwPatchRec*>(raw);
    if (lane < 4) { sh_nrf[lane] = 0; sh_ncand[lane] = 0; }
    int n_rings = 0, n_patches = 0;
    for (int k = 0; k < P->num_zones; ++k) { n_rings += P->num_rings_each_zone[k]; n_patches += P->num_rings_each_zone[k] * P->num_sectors_each_zone[k]; }
    for (int w = lane; w < n_patches * (int)(sizeof(PwPatchRec) / 4); w += blockDim.x) reinterpret_cast<int*>(raw)[w] = reinterpret_cast<const int*>(recs)[w];
    __syncthreads();
    if (lane < n_rings) {
        int zone = 0, ring = lane;
        while (ring >= P->num_rings_each_zone[zone]) { ring -= P->num_rings_each_zone[zone]; zone++; }
        const int cidx = lane;
        const int ns = P->num_sectors_each_zone[zone];
        const int base = G->patch_base[zone] + ring * ns;
        const bool is_near = cidx < P->num_rings_of_interest;
        double* rf = sh_rf[cidx < 4 ? cidx : 0];    // this ring's part of `ringwise_flatness` (<= sectors per ring <= 64; only near rings add to it)
        int nrf = 0, ncand = 0;
        for (int s = 0; s < ns; ++s) {
            PwPatchRec& r = lrec[base + s];
            if (r.decision != -1) continue;              // < num_min_pts points: all non-ground
            const double upright = r.normal[2], elevation = r.mean[2];
            const double flatness = fmin(fmin((double)r.sv[0], (double)r.sv[1]), (double)r.sv[2]);
            double heading = 0.0;
            for (int i = 0; i < 3; ++i) heading += (double)(r.mean[i] * r.normal[i]);
            const bool is_upright = upright > P->uprightness_thr;
            const bool heading_outside = heading < 0.0;
            bool not_elevated = false, is_flat = false;
            if (is_near) {
                not_elevated = elevation < S->elevation_thr[cidx];
                is_flat = flatness < S->flatness_thr[cidx];
            }
            if (is_upright && not_elevated && is_near) {     // :254-260
                int e = (S->elev_head[cidx] + S->elev_cnt[cidx]) & (PW_STORE_CAP - 1);
                S->elev[cidx][e] = elevation;
                S->elev_cnt[cidx]++;
                int f = (S->flat_head[cidx] + S->flat_cnt[cidx]) & (PW_STORE_CAP - 1);
                S->flat[cidx][f] = flatness;
                S->flat_cnt[cidx]++;
                rf[nrf++] = flatness;
            }
            int decision;
            if (!is_upright) decision = 0;
            else if (!is_near) decision = 1;
            else if (!heading_outside) decision = 0;
            else if (not_elevated || is_flat) decision = 1;
            else { decision = 2; ncand++; }
            r.decision = decision;
        }
        if (cidx < 4) { sh_nrf[cidx] = nrf; sh_ncand[cidx] = ncand; }
    }
    __syncthreads();
    if (lane < n_rings) {
        int zone = 0, ring = lane;
        while (ring >= P->num_rings_each_zone[zone]) { ring -= P->num_rings_each_zone[zone]; zone++; }
        const int cidx = lane;
        const int ns = P->num_sectors_each_zone[zone];
        const int base = G->patch_base[zone] + ring * ns;
        const int ncand = cidx < 4 ? sh_ncand[cidx] : 0;
        if (ncand > 0) {                                     // :293-305
            // `ringwise_flatness` is cleared only at the end of a ring that HAD candidates (:303-304 sit inside
            // `if (!candidates.empty())`): the values of the candidate-free rings before this one are still in it
            int first = cidx;
            while (first > 0 && sh_ncand[first - 1] == 0) --first;
            int nrf = 0;
            for (int r2 = first; r2 <= cidx; ++r2) nrf += sh_nrf[r2];
            double mean_f = 0.0, std_f = 0.0;
            if (nrf > 1) {
                double sm = 0.0;
                for (int r2 = first; r2 <= cidx; ++r2)
                    for (int i = 0; i < sh_nrf[r2]; ++i) sm += sh_rf[r2][i];
                mean_f = sm / (double)nrf;
                for (int r2 = first; r2 <= cidx; ++r2)
                    for (int i = 0; i < sh_nrf[r2]; ++i) std_f += (sh_rf[r2][i] - mean_f) * (sh_rf[r2][i] - mean_f);
                std_f /= (double)(nrf - 1);
                std_f = sqrt(std_f);
            }
            for (int s = 0; s < ns; ++s) {
                PwPatchRec& r = lrec[base + s];
                if (r.decision != 2) continue;
                bool revert = false;
                if (P->enable_TGR) {
                    const double flatness = fmin(fmin((double)r.sv[0], (double)r.sv[1]), (double)r.sv[2]);
                    const double line_variable = r.sv[1] != 0 ? (double)(r.sv[0] / r.sv[1]) : DBL_MAX;
                    double mu = mean_f + 1.5 * std_f;
                    double prob = 1 / (1 + exp((flatness - mu) / (mu / 10)));
                    if (r.n_ground > 1500 && flatness < P->th_dist * P->th_dist) prob = 1.0;
                    double prob_line = 1.0;
                    if (line_variable > 8.0) prob_line = 0.0;
                    revert = (prob_line * prob > 0.5) && (cidx < P->num_rings_of_interest);
                }
                r.decision = revert ? 1 : 0;
            }
        }
    }
    __syncthreads();
    for (int w = lane; w < n_patches; w += blockDim.x) recs[w].decision = lrec[w].decision;
    __syncthreads();                                        // `raw` changes hands
    // ---- update_elevation_thr (:339-358) and update_flatness_thr (:360-376) ----
    // Mean and standard deviation of the eight stores (<= ~1050 values each), summed SEQUENTIALLY in stored order like the
    // reference's loops (the order is part of the parity model).  One lane per store walking its ring buffer in global memory
    // was a chain of ~2 100 dependent-latency reads: 373 us per pass once the stores are full, two thirds of the whole ground
    // pass.  The wave now copies the stores into LDS with coalesced loads, 512 values at a time, and the owner lanes add from
    // there: the same additions in the same order, ~10 us.
    double (*const sbuf)[512] = reinterpret_cast<double (*)[512]>(raw);      // [8][512]
    __shared__ int sh_head[8], sh_cnt[8];
    const int sid = lane < 4 ? lane : ((lane >= 8 && lane < 12) ? lane - 4 : -1);      // stores 0..3 elevation, 4..7 flatness
    bool own = false;
    if (sid >= 0 && sid < 4) own = sid < P->num_rings_of_interest && S->elev_cnt[sid] > 0;
    if (sid >= 4) {
        const int i = sid - 4;
        own = i < P->num_rings_of_interest;
        for (int j = 0; j <= i; ++j)
            if (S->flat_cnt[j] <= 1) own = false;           // the reference BREAKS at the first ring with <= 1 entries
    }
    if (lane < 8) {
        sh_head[lane] = lane < 4 ? S->elev_head[lane] : S->flat_head[lane - 4];
        sh_cnt[lane] = lane < 4 ? S->elev_cnt[lane] : S->flat_cnt[lane - 4];
    }
    __syncthreads();
    int maxcnt = 0;
    for (int t = 0; t < 8; ++t) maxcnt = max(maxcnt, sh_cnt[t]);
    const int cnt = sid >= 0 ? sh_cnt[sid] : 0;
    double mean = 0.0, stdev = 0.0;
    for (int pass = 0; pass < 2; ++pass) {
        double acc = 0.0;
        for (int c0 = 0; c0 < maxcnt; c0 += 512) {
            for (int t = 0; t < 8; ++t) {
                const double* v = t < 4 ? S->elev[t] : S->flat[t - 4];
                const int m = min(512, sh_cnt[t] - c0);
                for (int i = lane; i < m; i += blockDim.x) sbuf[t][i] = v[(sh_head[t] + c0 + i) & (PW_STORE_CAP - 1)];
            }
            __syncthreads();
            if (own && cnt > 1) {
                const int m = min(512, cnt - c0);
                if (pass == 0) {
#pragma unroll 8
                    for (int i = 0; i < m; ++i) acc += sbuf[sid][i];
                } else {
#pragma unroll 8
                    for (int i = 0; i < m; ++i) { const double x = sbuf[sid][i]; acc += (x - mean) * (x - mean); }
                }
            }
            __syncthreads();
        }
        if (own && cnt > 1) {                               // pw_mean_stdev (:558-567)
            if (pass == 0) mean = acc / (double)cnt;
            else { stdev = acc / (double)(cnt - 1); stdev = sqrt(stdev); }
        }
    }
    if (own && sid < 4) {
        const int i = sid;
        if (i == 0) {
            S->elevation_thr[i] = mean + 3 * stdev;
            S->sensor_height = -mean;
        } else
            S->elevation_thr[i] = mean + 2 * stdev;
        int exceed = S->elev_cnt[i] - P->max_elevation_storage;
        if (exceed > 0) {
            S->elev_head[i] = (S->elev_head[i] + exceed) & (PW_STORE_CAP - 1);
            S->elev_cnt[i] -= exceed;
        }
    }
    if (own && sid >= 4) {
        const int i = sid - 4;
        S->flatness_thr[i] = mean + stdev;
        int exceed = S->flat_cnt[i] - P->max_flatness_storage;
        if (exceed > 0) {
            S->flat_head[i] = (S->flat_head[i] + exceed) & (PW_STORE_CAP - 1);
            S->flat_cnt[i] -= exceed;
        }
    }
}

__global__ void k_pw_finalize(const int* __restrict__ patch_id, const unsigned char* __restrict__ inlier,
                              const PwPatchRec* __restrict__ recs, int n, unsigned char* __restrict__ mask) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int pid = patch_id[i];
    mask[i] = (pid >= 0 && inlier[i] == 1 && recs[pid].decision == 1) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
static void pw_geometry(const vg_ground_params& p, PwGeom& g) {
    double z2 = (7 * p.min_range + p.max_range) / 8.0;
    double z3 = (3 * p.min_range + p.max_range) / 4.0;
    double z4 = (p.min_range + p.max_range) / 2.0;
    g.min_ranges[0] = p.min_range; g.min_ranges[1] = z2; g.min_ranges[2] = z3; g.min_ranges[3] = z4;
    g.ring_sizes[0] = (z2 - p.min_range) / p.num_rings_each_zone[0];
    g.ring_sizes[1] = (z3 - z2) / p.num_rings_each_zone[1];
    g.ring_sizes[2] = (z4 - z3) / p.num_rings_each_zone[2];
    g.ring_sizes[3] = (p.max_range - z4) / p.num_rings_each_zone[3];
    int n = 0;
    for (int k = 0; k < 4; ++k) {
        g.sector_sizes[k] = 2 * M_PI / p.num_sectors_each_zone[k];
        g.patch_base[k] = n;
        n += p.num_rings_each_zone[k] * p.num_sectors_each_zone[k];
    }
    g.patch_base[4] = n;
    g.n_patches = n;
}

extern "C" {

void vg_ground_default_params(vg_ground_params* p) {   // patchworkpp.h:75-107
    memset(p, 0, sizeof(*p));
    p->enable_RNR = p->enable_RVPF = p->enable_TGR = 1;
    p->num_iter = 3; p->num_lpr = 20; p->num_min_pts = 10; p->num_zones = 4; p->num_rings_of_interest = 4;
    p->RNR_ver_angle_thr = -15.0; p->RNR_intensity_thr = 0.2;
    p->sensor_height = 1.723; p->th_seeds = 0.125; p->th_dist = 0.125; p->th_seeds_v = 0.25; p->th_dist_v = 0.1;
    p->max_range = 80.0; p->min_range = 2.7; p->uprightness_thr = 0.707; p->adaptive_seed_selection_margin = -1.2;
    const int s[4] = {16, 32, 54, 32}, r[4] = {2, 4, 4, 4};
    for (int i = 0; i < 4; ++i) { p->num_sectors_each_zone[i] = s[i]; p->num_rings_each_zone[i] = r[i]; }
    p->max_flatness_storage = 1000; p->max_elevation_storage = 1000;
}

static int pw_upload(vg_ground* h) {
    pw_geometry(h->p, h->g);
    VG_CHECK(hipMemcpy(h->d_p, &h->p, sizeof(h->p), hipMemcpyHostToDevice));
    VG_CHECK(hipMemcpy(h->d_g, &h->g, sizeof(h->g), hipMemcpyHostToDevice));
    PwState* st = new PwState();
    memset(st, 0, sizeof(*st));
    st->sensor_height = h->p.sensor_height;
    for (int i = 0; i < 4; ++i) { st->elevation_thr[i] = h->p.elevation_thr[i]; st->flatness_thr[i] = h->p.flatness_thr[i]; }
    hipError_t e = hipMemcpy(h->d_state, st, sizeof(*st), hipMemcpyHostToDevice);
    delete st;
    VG_CHECK(e);
    return VG_OK;
}

int vg_ground_create(vg_ground** out, const vg_ground_params* p, int max_points) {
    if (!out || !p || max_points <= 0) return VG_ERR_ARG;
    int np = 0, nr = 0;
    for (int k = 0; k < 4; ++k) {
        np += p->num_rings_each_zone[k] * p->num_sectors_each_zone[k];
        nr += p->num_rings_each_zone[k];
        if (p->num_sectors_each_zone[k] > 64 || p->num_sectors_each_zone[k] <= 0) return VG_ERR_ARG;
    }
    if (np > PW_MAX_PATCHES || nr > 64 || p->num_zones != 4 || p->num_rings_of_interest > 4 ||
        p->max_elevation_storage + 64 > PW_STORE_CAP || p->max_flatness_storage + 64 > PW_STORE_CAP)
        return VG_ERR_ARG;
    vg_ground* h = new vg_ground();
    memset(h, 0, sizeof(*h));
    h->p = *p;
    h->max_points = max_points;
    VG_CHECK(hipMalloc(&h->d_p, sizeof(vg_ground_params)));
    VG_CHECK(hipMalloc(&h->d_g, sizeof(PwGeom)));
    VG_CHECK(hipMalloc(&h->d_state, sizeof(PwState)));
    VG_CHECK(hipMalloc(&h->d_patch_id, sizeof(int) * (size_t)max_points));
    VG_CHECK(hipMalloc(&h->d_count, sizeof(int) * PW_MAX_PATCHES));
    VG_CHECK(hipMemset(h->d_count, 0, sizeof(int) * PW_MAX_PATCHES));
    VG_CHECK(hipMalloc(&h->d_offset, sizeof(int) * (PW_MAX_PATCHES + 1)));
    VG_CHECK(hipMalloc(&h->d_cursor, sizeof(int) * PW_MAX_PATCHES));
    VG_CHECK(hipMalloc(&h->d_keys, sizeof(unsigned long long) * (size_t)max_points));
    VG_CHECK(hipMalloc(&h->d_inlier, (size_t)max_points));
    VG_CHECK(hipMalloc(&h->d_rec, sizeof(PwPatchRec) * PW_MAX_PATCHES));
    int rc = pw_upload(h);
    if (rc) return rc;
    *out = h;
    return VG_OK;
}

void vg_ground_destroy(vg_ground* h) {
    if (!h) return;
    (void)hipFree(h->d_p); (void)hipFree(h->d_g); (void)hipFree(h->d_state); (void)hipFree(h->d_patch_id);
    (void)hipFree(h->d_count); (void)hipFree(h->d_offset); (void)hipFree(h->d_cursor); (void)hipFree(h->d_keys);
    (void)hipFree(h->d_inlier); (void)hipFree(h->d_rec);
    delete h;
}

int vg_ground_reset(vg_ground* h, const vg_ground_params* p) {
    if (!h) return VG_ERR_ARG;
    if (p) h->p = *p;
    return pw_upload(h);
}

int vg_ground_estimate(vg_ground* h, const float* d_points, int n, int stride, double z_offset,
                       uint8_t* d_ground_mask, void* stream) {
    if (!h || !d_points || !d_ground_mask || n < 0 || stride < 4) return VG_ERR_ARG;
    if (n > h->max_points) return VG_ERR_CAPACITY;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return VG_OK;
    const int nb = vg_div_up(n, 256);
    hipLaunchKernelGGL(k_pw_classify, dim3(nb), dim3(256), 0, st, d_points, n, stride, z_offset, h->d_p, h->d_g,
                       h->d_state, h->d_patch_id, h->d_count, h->d_inlier);
    hipLaunchKernelGGL(k_pw_offsets, dim3(1), dim3(64), 0, st, h->d_count, h->d_offset, h->d_cursor, h->g.n_patches);
    hipLaunchKernelGGL(k_pw_scatter, dim3(nb), dim3(256), 0, st, d_points, n, stride, z_offset, h->d_patch_id,
                       h->d_offset, h->d_cursor, h->d_keys);
    {
        VG_MAX_DYNAMIC_LDS(k_pw_sort_big, PW_BIG_LDS_POINTS * 8);
        hipLaunchKernelGGL(k_pw_sort_big, dim3(h->g.n_patches), dim3(1024), PW_BIG_LDS_POINTS * 8, st, h->d_p, h->d_offset, h->d_keys);
    }
    // k_pw_patch reads x, y straight from d_points and z from the key (already offset)
    hipLaunchKernelGGL(k_pw_patch, dim3(h->g.n_patches), dim3(PW_T), 0, st, d_points, stride, h->d_p, h->d_g, h->d_state,
                       h->d_offset, h->d_keys, h->d_inlier, h->d_rec);
    hipLaunchKernelGGL(k_pw_decide, dim3(1), dim3(256), 0, st, h->d_p, h->d_g, h->d_state, h->d_rec);
    hipLaunchKernelGGL(k_pw_finalize, dim3(nb), dim3(256), 0, st, h->d_patch_id, h->d_inlier, h->d_rec, n, d_ground_mask);
    VG_LAUNCH_CHECK();
    return VG_OK;
}

/* synchronous: sensor_height, elevation_thr[4], flatness_thr[4], stored counts elev[4], flat[4] */
int vg_ground_get_state(vg_ground* h, double* h_out17, void* stream) {
    if (!h || !h_out17) return VG_ERR_ARG;
    VG_CHECK(hipStreamSynchronize((hipStream_t)stream));
    PwState* st = new PwState();
    hipError_t e = hipMemcpy(st, h->d_state, sizeof(PwState), hipMemcpyDeviceToHost);
    if (e == hipSuccess) {
        h_out17[0] = st->sensor_height;
        for (int i = 0; i < 4; ++i) {
            h_out17[1 + i] = st->elevation_thr[i];
            h_out17[5 + i] = st->flatness_thr[i];
            h_out17[9 + i] = st->elev_cnt[i];
            h_out17[13 + i] = st->flat_cnt[i];
        }
    }
    delete st;
    VG_CHECK(e);
    return VG_OK;
}

/* The complete frame-to-frame state (PwState: sensor height, thresholds, the elevation / flatness stores with their ring
 * cursors) as an opaque host blob of vg_ground_state_bytes() bytes: export after frame f on one handle, set on another
 * (another stream, another GPU / rank) and frame f+1 continues exactly as on the first.  Both synchronous. */
int64_t vg_ground_state_bytes(void) { return (int64_t)sizeof(PwState); }

int vg_ground_export_state(vg_ground* h, void* h_blob, void* stream) {
    if (!h || !h_blob) return VG_ERR_ARG;
    VG_CHECK(hipStreamSynchronize((hipStream_t)stream));
    VG_CHECK(hipMemcpy(h_blob, h->d_state, sizeof(PwState), hipMemcpyDeviceToHost));
    return VG_OK;
}

int vg_ground_set_state(vg_ground* h, const void* h_blob, void* stream) {
    if (!h || !h_blob) return VG_ERR_ARG;
    const PwState* st = (const PwState*)h_blob;
    for (int i = 0; i < 4; ++i)                 // a blob from a handle with other storage limits would index out of range
        if (st->elev_cnt[i] < 0 || st->elev_cnt[i] > PW_STORE_CAP || st->flat_cnt[i] < 0 || st->flat_cnt[i] > PW_STORE_CAP ||
            st->elev_head[i] < 0 || st->elev_head[i] >= PW_STORE_CAP || st->flat_head[i] < 0 || st->flat_head[i] >= PW_STORE_CAP)
            return VG_ERR_ARG;
    VG_CHECK(hipStreamSynchronize((hipStream_t)stream));
    VG_CHECK(hipMemcpy(h->d_state, h_blob, sizeof(PwState), hipMemcpyHostToDevice));
    return VG_OK;
}

int vg_ground_num_patches(const vg_ground* h) { return h ? h->g.n_patches : 0; }

/* synchronous: [n_patches,12] = n, n_ground, normal[3], mean[3], sv[3], decision */
int vg_ground_get_patch_info(vg_ground* h, float* h_out, void* stream) {
    if (!h || !h_out) return VG_ERR_ARG;
    VG_CHECK(hipStreamSynchronize((hipStream_t)stream));
    PwPatchRec* r = new PwPatchRec[PW_MAX_PATCHES];
    hipError_t e = hipMemcpy(r, h->d_rec, sizeof(PwPatchRec) * h->g.n_patches, hipMemcpyDeviceToHost);
    if (e == hipSuccess) {
        for (int i = 0; i < h->g.n_patches; ++i) {
            float* o = h_out + (size_t)i * 12;
            o[0] = (float)r[i].n;
            o[1] = (float)r[i].n_ground;
            for (int k = 0; k < 3; ++k) { o[2 + k] = r[i].normal[k]; o[5 + k] = r[i].mean[k]; o[8 + k] = r[i].sv[k]; }
            o[11] = (float)r[i].decision;
        }
    }
    delete[] r;
    VG_CHECK(e);
    return VG_OK;
}

}  // extern "C"
