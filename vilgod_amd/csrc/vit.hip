// CLIP ViT image tower + zero-shot scoring for gfx950.
//
// Replaces third_party/CLIP/clip/model.py:206-240 (VisionTransformer.forward), :171-192
// (ResidualAttentionBlock), :157-168 (LayerNorm in fp32, QuickGELU) as called from
// src/utils/clip_utils.py:39-44 (encode_image -> normalise -> 100*cos -> softmax).
//
// Two precisions, one code path:
//   dtype 1 (f16): GEMM operands and activations fp16, fp32 accumulate, fp32 LayerNorm -- what the
//                  reference runs on a GPU (model.py:375-396 convert_weights) -- except that the
//                  residual stream is kept in fp32 (strictly more accurate than the reference's fp16
//                  stream).  Projection GEMMs: k_gemm_f16_pp64 (v_mfma_f32_16x16x32_f16, 256 x 256 x 64 tiles, 8 waves as two
//                  groups one barrier apart, LDS-DMA rings, LayerNorm folded into the epilogues); shapes with N % 256 != 0:
//                  k_gemm_f16 (256 x 128 x 32).  Attention: persistent workgroups, item = (crop, head), K and V row-major in
//                  LDS, S^T = K Q^T and O^T = V^T P^T both on MFMA with the softmax row living on one lane.
//   dtype 0 (f32): parity mode against the fp32 CPU reference (tolerance 1e-3 on probabilities): the same tower in fp32 --
//                  k_gemm_f32_mfma (v_mfma_f32_32x32x2_f32, 128 x 128 tiles; bit-identical to the vector-ALU k_gemm_f32 that
//                  serves the other shapes), k_attention_f32, k_layernorm -- same epilogues.
//
// Layout (all row-major, K contiguous):
//   tokens M = n_crops * T (T = 1 + (res/patch)^2), rows padded to a multiple of 256 (the padding rows of the
//   residual stream are re-zeroed by k_embed_lnpre on every call; the other buffers' padding rows are written by
//   the GEMMs from them and stay finite).
//   x    [Mp, W]   f32   residual stream
//   h    [Mp, W]   f16|f32   LayerNorm output / attention output
//   qkv  [Mp, 3W]  f16|f32
//   mlp  [Mp, 4W]  f16|f32
#include "common.h"
#include <hip/hip_fp16.h>
#include <string.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <map>
#include <atomic>
#include <mutex>
#include <utility>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// QuickGELU x * sigmoid(1.702 x) (model.py:166-168) as x / (1 + 2^(x * QGELU_C)): one multiply in front of v_exp_f32 instead of two
#define QGELU_C (-1.702f * 1.4426950408889634f)
enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RESID = 2, EPI_NONE_F32 = 3,
       EPI_BIAS_RESID_H = 4,     // 4: fp16 residual stream (k_gemm_f16_pp64 only): resid_h = f16(resid_h + f16(acc + bias))
       EPI_BIAS_RESID_HL = 5 };  // 5: the residual stream as an fp16 PAIR (k_gemm_f16_w4 only, LN = 2): x = hi + lo, hi = the fp16 copy the next
                                 //    GEMM reads (ln_x16), lo = f16(x - hi) (`resid`, as f16*): 22 bits of x in 4 bytes, read 4 + written 4 per element
                                 //    instead of read 4 + written 6

// ---------------------------------------------------------------------------------------------
// XCD-aware tile order: consecutive hardware block ids are dealt round-robin to the 8 XCDs; give each
// XCD a contiguous run of tiles so the column tiles that share one activation row-tile hit one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, s = bid >> 3;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + s;
}

// ---------------------------------------------------------------------------------------------
// fp16 MFMA GEMM:  C[m][n] = sum_k X[m][k] * Wt[n][k]  (+ epilogue).  M%128==0, N%128==0, K%64==0.
// The MFMA computes the TRANSPOSED tile (A operand = weight rows, B operand = activation rows) so a
// lane owns one output row m and 4-wide runs of consecutive n: vector loads/stores in the epilogue.
#define GBM 256                             // block tile: 256 (tokens) x 128 (features), K-step 32
#define GBN 128
#define GK 32
#define G_STAGES 3
#define G_STAGE_BYTES 24576                 // one stage: X tile 16 KB + W tile 8 KB, rows of 64 B, linear
#define G_W_OFF 16384
#define G_EPI_LD 132                        // epilogue tile row stride in floats (128 + 4: conflict-free b128 writes)
#define G_LDS_BYTES (G_STAGES * G_STAGE_BYTES)   // 73,728 B (>= 128*132*4 = 67,584 B half-tile) -> 2 workgroups / CU

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// fp16 MFMA GEMM:  C[m][n] = sum_k X[m][k] * Wt[n][k]  (+ epilogue).  M%256==0, N%128==0, K%32==0.
//  * 8 waves (4 x 2), each a 64x64 sub-tile = 2x2 v_mfma_f32_32x32x16_f16; the MFMA computes the TRANSPOSED tile
//    (A operand = weight rows, B operand = activation rows) so a lane owns one output row.
//  * staging by LDS-DMA (global_load_lds, 16 B per lane): no staging registers, no ds_write pass.  The DMA writes
//    LDS linearly (wave-uniform base + lane*16), so the bank-conflict swizzle is applied to the per-lane SOURCE
//    address and again to the ds_read address (guide rule 21): 16-B chunk c of the 64-B row r sits at slot
//    c ^ ((r>>2)&3).
//  * 3 stages, TWO tiles in flight: per K-step one counted s_waitcnt vmcnt(3) (this wave's 3 newest DMAs = the next
//    tile may stay in flight) + one raw s_barrier; the DMA for tile k+2 is issued right after the barrier.
//  * 72 KB of LDS and <= 128 VGPRs -> TWO workgroups per CU: one block's barrier waits and epilogue overlap the other's
//    MFMAs (ablation: compute-only and DMA-only each take ~60 % of the fused time when run alone in one block).
//  * epilogue through LDS in two 128-row halves: accumulators -> [128][128] f32 tile -> whole rows, 16 B per lane,
//    fused bias / QuickGELU / residual update.  ldc = row stride of C (qkv uses a padded stride: 4608-B rows alias
//    in the memory channels and cost +20 %).
template <int EPI, int VAR = 0>
__global__ __launch_bounds__(512, 2) void k_gemm_f16(const f16* __restrict__ X, const f16* __restrict__ Wt,
                                                     const float* __restrict__ bias, void* __restrict__ Cout,
                                                     float* __restrict__ resid, int M, int N, int K, int ldc, int cw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // tile order: column CHUNKS of cw tiles (<= ~2.4 MB of weights, so the chunk stays in one XCD's 4 MB L2 next to the
    // streaming activations), inside a chunk row-tile major; each XCD walks a contiguous run of this order.
    const int ntm = M / GBM;
    const int t = xcd_remap(blockIdx.x, gridDim.x);
    const int per_chunk = ntm * cw;
    const int chunk = t / per_chunk, tc = t - chunk * per_chunk;
    const int tm = tc / cw, tn = chunk * cw + (tc - tm * cw);
    const int m0 = tm * GBM, n0 = tn * GBN;
    const int wm = wave >> 1, wn = wave & 1;          // 4 x 2 waves, 64x64 each

    // ---- DMA addressing: a 1-KB instruction covers 16 rows x 64 B.  This wave fills X rows [wave*32, +32) (2
    //      instructions) and W rows [wave*16, +16) (1) ----
    const int l2 = lane >> 2, pslot = lane & 3;
    const f16* xsrc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int R = wave * 32 + q * 16 + l2;
        xsrc[q] = X + (size_t)(m0 + R) * K + (pslot ^ ((R >> 2) & 3)) * 8;
    }
    const int RW = wave * 16 + l2;
    const f16* wsrc = Wt + (size_t)(n0 + RW) * K + (pslot ^ ((RW >> 2) & 3)) * 8;
    auto issue = [&](int kt, int stage) {
        char* sb = smem + stage * G_STAGE_BYTES;
        __builtin_amdgcn_global_load_lds((glb_void*)(xsrc[0] + (size_t)kt * GK), (lds_void*)(sb + (wave * 32) * 64), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void*)(xsrc[1] + (size_t)kt * GK), (lds_void*)(sb + (wave * 32 + 16) * 64), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void*)(wsrc + (size_t)kt * GK), (lds_void*)(sb + G_W_OFF + (wave * 16) * 64), 16, 0, 0);
    };

    f32x16 acc00, acc01, acc10, acc11;   // acc[ni][mi]
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }

    const int r31 = lane & 31, hh = lane >> 5;
    const int sw = (r31 >> 2) & 3;
    const int xrow = (wm * 64 + r31) * 64, wrow = G_W_OFF + (wn * 64 + r31) * 64;
    const int nk = K / GK;
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt must have landed; the 3 newest DMAs of this wave (tile kt+1) may stay in flight
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();    // every wave's part of tile kt is in LDS; stage (kt+2)%3 is no longer read
        if (VAR != 1 && kt + 2 < nk) issue(kt + 2, (kt + 2) % G_STAGES);
        const char* sb = smem + (kt % G_STAGES) * G_STAGE_BYTES;
        if (VAR == 2) continue;
#pragma unroll
        for (int s = 0; s < GK / 16; ++s) {
            const int po = ((2 * s + hh) ^ sw) * 16;
            f16x8 fa0 = *(const f16x8*)(sb + wrow + po);
            f16x8 fa1 = *(const f16x8*)(sb + wrow + 32 * 64 + po);
            f16x8 fb0 = *(const f16x8*)(sb + xrow + po);
            f16x8 fb1 = *(const f16x8*)(sb + xrow + 32 * 64 + po);
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0, fb0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0, fb1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1, fb0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1, fb1, acc11, 0, 0, 0);
        }
    }
    if (VAR == 3) { if (acc00[0] + acc01[1] + acc10[2] + acc11[3] == 12345.f) ((float*)Cout)[0] = 1.f; return; }
    // ---- epilogue through LDS, two halves of 128 rows (waves wm = 0,1 hold rows 0..127; wm = 2,3 rows 128..255) ----
    float* ct = (float*)smem;
    f32x16 acc[2][2] = {{acc00, acc01}, {acc10, acc11}};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        if ((wm >> 1) == half) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int m = (wm & 1) * 64 + mi * 32 + r31;     // lane owns row m; register r -> n = (r&3) + 8*(r>>2) + 4*hh
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = wn * 64 + ni * 32 + 8 * g + 4 * hh;
                        *(float4*)(ct + m * G_EPI_LD + n) = make_float4(acc[ni][mi][4 * g], acc[ni][mi][4 * g + 1],
                                                                        acc[ni][mi][4 * g + 2], acc[ni][mi][4 * g + 3]);
                    }
            }
        }
        __syncthreads();
        const int mh = m0 + half * 128;
        if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
            // f16 output: a row of 128 values = 256 B = 16 lanes x 16 B; 32 rows per pass
            const int cn = (tid & 15) * 8, rr = tid >> 4;
            const float4 b0 = *(const float4*)(bias + n0 + cn), b1 = *(const float4*)(bias + n0 + cn + 4);
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int m = pass * 32 + rr;
                const float4 v0 = *(const float4*)(ct + m * G_EPI_LD + cn), v1 = *(const float4*)(ct + m * G_EPI_LD + cn + 4);
                float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                f16x8 h8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float x = v[e] + bb[e];
                    if (EPI == EPI_BIAS_GELU) x = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * QGELU_C));   // QuickGELU (model.py:166-168)
                    h8[e] = (f16)x;
                }
                *(f16x8*)((f16*)Cout + (size_t)(mh + m) * ldc + n0 + cn) = h8;
            }
        } else {
            // f32 output / residual update: a row of 128 floats = 512 B = 32 lanes x 16 B; 16 rows per pass
            const int cn = (tid & 31) * 4, rr = tid >> 5;
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (EPI == EPI_BIAS_RESID) b4 = *(const float4*)(bias + n0 + cn);
#pragma unroll
            for (int pass = 0; pass < 8; ++pass) {
                const int m = pass * 16 + rr;
                float4 v = *(const float4*)(ct + m * G_EPI_LD + cn);
                if (EPI == EPI_BIAS_RESID) {
                    float* p = resid + (size_t)(mh + m) * ldc + n0 + cn;
                    const float4 x4 = *(const float4*)p;
                    v.x += b4.x + x4.x; v.y += b4.y + x4.y; v.z += b4.z + x4.z; v.w += b4.w + x4.w;
                    *(float4*)p = v;
                } else {
                    *(float4*)((float*)Cout + (size_t)(mh + m) * ldc + n0 + cn) = v;
                }
            }
        }
    }
}

#ifdef VG_DEV      // superseded K-step-32 ping-pong kernel: kept for the cycle-stamp tools (tools/dev), not in the product library
#include "dev/vit_gemm_pp32.inc"
#endif  // VG_DEV

// ---------------------------------------------------------------------------------------------
// fp32 parity-mode GEMM (VALU): 64x64 tile, 256 threads, 4x4 micro-tile, same epilogues.
template <int EPI>
__global__ __launch_bounds__(256) void k_gemm_f32(const float* __restrict__ X, const float* __restrict__ Wt,
                                                  const float* __restrict__ bias, float* __restrict__ Cout,
                                                  float* __restrict__ resid, int M, int N, int K) {
    __shared__ float Xs[16][64 + 4];
    __shared__ float Ws[16][64 + 4];
    const int tid = threadIdx.x;
    const int ntn = N / 64;
    const int tm = blockIdx.x / ntn, tn = blockIdx.x - tm * ntn;
    const int m0 = tm * 64, n0 = tn * 64;
    const int ty = tid >> 4, tx = tid & 15;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        // 64 rows x 16 k per operand = 1024 values, 4 per thread
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int c = tid + 256 * i, row = c >> 4, kk = c & 15;
            Xs[kk][row] = X[(size_t)(m0 + row) * K + k0 + kk];
            Ws[kk][row] = Wt[(size_t)(n0 + row) * K + k0 + kk];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = Xs[kk][ty * 4 + i];
                b[i] = Ws[kk][tx * 4 + i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            float v = acc[i][j];
            if (EPI != EPI_NONE_F32) v += bias[n];
            if (EPI == EPI_BIAS_GELU) v = v / (1.0f + expf(-1.702f * v));
            if (EPI == EPI_BIAS_RESID)
                resid[(size_t)m * N + n] += v;
            else
                Cout[(size_t)m * N + n] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fp32 parity-mode GEMM on the matrix cores (round 4): v_mfma_f32_32x32x2_f32 -- f32 in, f32 accumulate, each instruction two exact
// fused multiply-adds per output element in k order, i.e. the same fmaf chain over ascending k as k_gemm_f32 above (the tower's
// numbers do not change), at the vector-ALU peak rate but with 1/32 of the LDS reads per FLOP of the 4 x 4 micro-tile kernel.
// 128 x 128 tile, 256 threads = 2 x 2 waves of 64 x 64 (2 x 2 blocks of 32 x 32: 64 accumulator registers), K-step 16.
// LDS tiles are k-major ([16][128 + 4] floats per operand): lane l reads A[k0 + l / 32][row l % 32] -- 32 consecutive floats per
// half-wave, conflict-free.  The next K-step's global loads (a float4 along k per thread and operand half) are in registers while the
// current one is multiplied.  D layout of the 32 x 32 instruction: register j of lane l = token 8 (j / 4) + 4 (l / 32) + j % 4,
// feature l % 32 -> a store instruction writes two 128-byte row segments.
typedef float f32x16m __attribute__((ext_vector_type(16)));
#define GF_LD 132
template <int EPI>
__global__ __launch_bounds__(256) void k_gemm_f32_mfma(const float* __restrict__ X, const float* __restrict__ Wt,
                                                       const float* __restrict__ bias, float* __restrict__ Cout,
                                                       float* __restrict__ resid, int M, int N, int K) {
    __shared__ float As[16 * GF_LD];
    __shared__ float Bs[16 * GF_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntn = N / 128;
    const int tm = blockIdx.x / ntn, tn = blockIdx.x - tm * ntn;
    const int m0 = tm * 128, n0 = tn * 128;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int l32 = lane & 31, lk = lane >> 5;
    // staging: thread t moves rows (t >> 2) and (t >> 2) + 64 of each operand, k quad t & 3
    const int srow = tid >> 2, skq = (tid & 3) * 4;
    float4 xa, xb, wa, wb;
    auto fetch = [&](int k0) {
        xa = *(const float4*)(X + (size_t)(m0 + srow) * K + k0 + skq);
        xb = *(const float4*)(X + (size_t)(m0 + srow + 64) * K + k0 + skq);
        wa = *(const float4*)(Wt + (size_t)(n0 + srow) * K + k0 + skq);
        wb = *(const float4*)(Wt + (size_t)(n0 + srow + 64) * K + k0 + skq);
    };
    auto stage = [&]() {
        const float xs[2][4] = {{xa.x, xa.y, xa.z, xa.w}, {xb.x, xb.y, xb.z, xb.w}};
        const float ws[2][4] = {{wa.x, wa.y, wa.z, wa.w}, {wb.x, wb.y, wb.z, wb.w}};
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                As[(skq + e) * GF_LD + srow + 64 * h] = xs[h][e];
                Bs[(skq + e) * GF_LD + srow + 64 * h] = ws[h][e];
            }
    };
    f32x16m acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += 16) {
        stage();
        __syncthreads();
        if (k0 + 16 < K) fetch(k0 + 16);
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = As[(kk + lk) * GF_LD + wm + 32 * i + l32];
                b[i] = Bs[(kk + lk) * GF_LD + wn + 32 * i + l32];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + 32 * j + l32;
            const float bn = EPI != EPI_NONE_F32 ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * i + 8 * (r >> 2) + 4 * lk + (r & 3);
                float v = acc[i][j][r];
                if (EPI != EPI_NONE_F32) v += bn;
                if (EPI == EPI_BIAS_GELU) v = v / (1.0f + expf(-1.702f * v));
                if (EPI == EPI_BIAS_RESID)
                    resid[(size_t)m * N + n] += v;
                else
                    Cout[(size_t)m * N + n] = v;
            }
        }
}

// ---------------------------------------------------------------------------------------------
// im2col of CHW crops into patch rows: P[(crop*G*G + py*G + px)][c*ps*ps + i*ps + j]
template <typename TI, typename TO>
__global__ void k_im2col(const TI* __restrict__ crops, TO* __restrict__ P, int n, int res, int ps) {
    const int G = res / ps, Kp = 3 * ps * ps;
    size_t total = (size_t)n * G * G * Kp;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int col = idx % Kp;
        size_t row = idx / Kp;
        int px = row % G, py = (row / G) % G;
        size_t crop = row / ((size_t)G * G);
        int j = col % ps, i = (col / ps) % ps, c = col / (ps * ps);
        P[idx] = (TO)(float)crops[((crop * 3 + c) * res + (py * ps + i)) * res + px * ps + j];
    }
}

// x[crop][t] = (t == 0 ? class_embedding : patch_out[crop][t-1]) + positional_embedding[t]; x = ln_pre(x)
// one wave per token row.  (model.py:227-229)
template <typename TO>
__global__ __launch_bounds__(256) void k_embed_lnpre(const float* __restrict__ patch_out, const float* __restrict__ cls,
                                                     const float* __restrict__ pos, const float* __restrict__ lw,
                                                     const float* __restrict__ lb, TO* __restrict__ x, int n_rows,
                                                     int T, int W, int n_rows_padded, const float* __restrict__ lw1 = nullptr,
                                                     const float* __restrict__ lb1 = nullptr, f16* __restrict__ h1 = nullptr,
                                                     f16* __restrict__ pair_hi = nullptr, f16* __restrict__ pair_lo = nullptr) {
    // pair_hi / pair_lo (TO = float, vectorised widths): the stream is written as the fp16 pair of EPI_BIAS_RESID_HL instead of to x
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n_rows_padded) return;
    if (row >= n_rows) {
        // padding rows of the residual stream (GEMM row tile): re-zeroed on every call -- every residual epilogue adds its
        // bias into them, and the workspace carve-up moves when n_crops changes
        if (pair_hi) for (int c = lane; c < W; c += 64) { pair_hi[(size_t)row * W + c] = (f16)0.f; pair_lo[(size_t)row * W + c] = (f16)0.f; }
        else for (int c = lane; c < W; c += 64) x[(size_t)row * W + c] = (TO)0.f;
        return;
    }
    const int crop = row / T, t = row - crop * T;
    const float* src = (t == 0) ? cls : patch_out + ((size_t)crop * (T - 1) + (t - 1)) * W;
    if ((W & 255) == 0 && W <= 1024) {
        // vectorised path (every shipped width): float4 per lane like k_layernorm -- 1 KB per wave instruction instead of 256 B
        const int nv = W >> 8;
        float4 v4[4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nv) {
                const int c = (i * 64 + lane) * 4;
                const float4 a = *(const float4*)(src + c), b = *(const float4*)(pos + (size_t)t * W + c);
                v4[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
                s += (v4[i].x + v4[i].y) + (v4[i].z + v4[i].w);
            }
        const float mean = vg_wave_sum(s) / (float)W;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nv) {
                const float a = v4[i].x - mean, b = v4[i].y - mean, c = v4[i].z - mean, d = v4[i].w - mean;
                q += (a * a + b * b) + (c * c + d * d);
            }
        const float rstd = rsqrtf(vg_wave_sum(q) / (float)W + 1e-5f);
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nv) {
                const int c = (i * 64 + lane) * 4;
                const float4 w4 = *(const float4*)(lw + c), b4 = *(const float4*)(lb + c);
                float o[4] = {(v4[i].x - mean) * rstd * w4.x + b4.x, (v4[i].y - mean) * rstd * w4.y + b4.y,
                              (v4[i].z - mean) * rstd * w4.z + b4.z, (v4[i].w - mean) * rstd * w4.w + b4.w};
                TO* dst = x + (size_t)row * W + c;
                if (sizeof(TO) == 2) {
                    const f16x4 h4 = {(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]};
                    *(f16x4*)dst = h4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (float)h4[e];           // what a separate LayerNorm kernel would read back
                } else if (pair_hi) {
                    const f16x4 h4 = {(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]};
                    const f16x4 l4 = {(f16)(o[0] - (float)h4[0]), (f16)(o[1] - (float)h4[1]), (f16)(o[2] - (float)h4[2]), (f16)(o[3] - (float)h4[3])};
                    *(f16x4*)(pair_hi + (size_t)row * W + c) = h4;
                    *(f16x4*)(pair_lo + (size_t)row * W + c) = l4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (float)h4[e] + (float)l4[e];          // what the stream holds from here on
                } else {
                    *(float4*)dst = make_float4(o[0], o[1], o[2], o[3]);
                }
                v4[i] = make_float4(o[0], o[1], o[2], o[3]);
                s2 += (o[0] + o[1]) + (o[2] + o[3]);
            }
        if (h1) {
            // ln_1 of the first block on the row just produced, k_layernorm's arithmetic (bit-identical to the separate launch)
            const float mean2 = vg_wave_sum(s2) / (float)W;
            float q2 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nv) {
                    const float a = v4[i].x - mean2, b = v4[i].y - mean2, c = v4[i].z - mean2, d = v4[i].w - mean2;
                    q2 += (a * a + b * b) + (c * c + d * d);
                }
            const float rstd2 = rsqrtf(vg_wave_sum(q2) / (float)W + 1e-5f);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nv) {
                    const int c = (i * 64 + lane) * 4;
                    const float4 w4 = *(const float4*)(lw1 + c), b4 = *(const float4*)(lb1 + c);
                    const f16x4 h4 = {(f16)((v4[i].x - mean2) * rstd2 * w4.x + b4.x), (f16)((v4[i].y - mean2) * rstd2 * w4.y + b4.y),
                                      (f16)((v4[i].z - mean2) * rstd2 * w4.z + b4.z), (f16)((v4[i].w - mean2) * rstd2 * w4.w + b4.w)};
                    *(f16x4*)(h1 + (size_t)row * W + c) = h4;
                }
        }
        return;
    }
    float v[16];
    const int per = W / 64;   // W multiple of 64, <= 1024
    float s = 0.f;
    for (int i = 0; i < per; ++i) {
        int c = lane + 64 * i;
        v[i] = src[c] + pos[(size_t)t * W + c];
        s += v[i];
    }
    float mean = vg_wave_sum(s) / (float)W;
    float q = 0.f;
    for (int i = 0; i < per; ++i) {
        float d = v[i] - mean;
        q += d * d;
    }
    float rstd = rsqrtf(vg_wave_sum(q) / (float)W + 1e-5f);
    float s2 = 0.f;
    for (int i = 0; i < per; ++i) {
        int c = lane + 64 * i;
        v[i] = (v[i] - mean) * rstd * lw[c] + lb[c];
        x[(size_t)row * W + c] = (TO)v[i];
        v[i] = (float)(TO)v[i];                     // what a separate LayerNorm kernel would read back
        s2 += v[i];
    }
    if (h1) {
        // ln_1 of the first block on the row just produced: one LayerNorm launch (and its read of the stream) less per encode
        const float mean2 = vg_wave_sum(s2) / (float)W;
        float q2 = 0.f;
        for (int i = 0; i < per; ++i) {
            float d = v[i] - mean2;
            q2 += d * d;
        }
        const float rstd2 = rsqrtf(vg_wave_sum(q2) / (float)W + 1e-5f);
        for (int i = 0; i < per; ++i) {
            int c = lane + 64 * i;
            h1[(size_t)row * W + c] = (f16)((v[i] - mean2) * rstd2 * lw1[c] + lb1[c]);
        }
    }
}

// LayerNorm (fp32 statistics, model.py:157-163): x f32 [rows,W] -> h (f16 or f32). one wave per row.
// W % 256 == 0 takes the vectorised path: float4 loads (1 KB per wave instruction), packed stores.
template <typename TO, typename TI = float>
__global__ __launch_bounds__(256) void k_layernorm(const TI* __restrict__ x, const float* __restrict__ lw,
                                                   const float* __restrict__ lb, TO* __restrict__ h, int n_rows, int W,
                                                   int row_stride_in /*in rows of W*/) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n_rows) return;
    const TI* src = x + (size_t)row * row_stride_in * W;
    if ((W & 255) == 0 && W <= 1024) {
        const int nv = W >> 8;                       // float4 per lane
        float4 v[4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nv) {
                if (sizeof(TI) == 2) {
                    const f16x4 t4 = *(const f16x4*)(src + (i * 64 + lane) * 4);
                    v[i] = make_float4((float)t4[0], (float)t4[1], (float)t4[2], (float)t4[3]);
                } else {
                    v[i] = *(const float4*)(src + (i * 64 + lane) * 4);
                }
                s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            }
        const float mean = vg_wave_sum(s) / (float)W;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nv) {
                const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
                q += (a * a + b * b) + (c * c + d * d);
            }
        const float rstd = rsqrtf(vg_wave_sum(q) / (float)W + 1e-5f);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < nv) {
                const int c = (i * 64 + lane) * 4;
                const float4 w4 = *(const float4*)(lw + c), b4 = *(const float4*)(lb + c);
                const float o0 = (v[i].x - mean) * rstd * w4.x + b4.x, o1 = (v[i].y - mean) * rstd * w4.y + b4.y;
                const float o2 = (v[i].z - mean) * rstd * w4.z + b4.z, o3 = (v[i].w - mean) * rstd * w4.w + b4.w;
                TO* dst = h + (size_t)row * W + c;
                if (sizeof(TO) == 2) {
                    f16x4 h4 = {(f16)o0, (f16)o1, (f16)o2, (f16)o3};
                    *(f16x4*)dst = h4;
                } else {
                    *(float4*)dst = make_float4(o0, o1, o2, o3);
                }
            }
        return;
    }
    float v[16];
    const int per = W / 64;
    float s = 0.f;
    for (int i = 0; i < per; ++i) {
        v[i] = (float)src[lane + 64 * i];
        s += v[i];
    }
    float mean = vg_wave_sum(s) / (float)W;
    float q = 0.f;
    for (int i = 0; i < per; ++i) {
        float d = v[i] - mean;
        q += d * d;
    }
    float rstd = rsqrtf(vg_wave_sum(q) / (float)W + 1e-5f);
    for (int i = 0; i < per; ++i) {
        int c = lane + 64 * i;
        h[(size_t)row * W + c] = (TO)((v[i] - mean) * rstd * lw[c] + lb[c]);
    }
}

// ---------------------------------------------------------------------------------------------
// fp16 attention, head dim 64, T <= 224.  One workgroup (7 waves) per (crop, head); wave w owns query rows
// [32w, 32w+32).  S^T[key][q] = K Q^T (A = K rows from LDS, B = Q rows from HBM), softmax along keys is a
// per-lane reduction (+ one cross-half shuffle), O^T[d][q] = V^T P^T with the S^T accumulators re-used
// as the B operand (guide §3 "An accumulator tile as the next MFMA's operand").
#define AT_MAXT 224
#define AT_KLD 72      // K rows: 64 + 8 halves
#define AT_LDS_BYTES ((AT_MAXT * AT_KLD + 64 * 228 + 7 * 32 * AT_KLD) * 2)   // 93,696 B
#define AT_LDS_BYTES_TR ((AT_MAXT * AT_KLD + AT_MAXT * AT_KLD + 7 * 32 * AT_KLD) * 2)   // 96,768 B (row-major V)
#define AT_VLD 228     // V^T rows: 224 + 4 halves (456 B: 32 rows of a fragment read hit 32 different banks; the 8 x 8 transposed
                       // writes of a wave land 2-way conflicted)
// NKB: number of 32-key blocks, ceil(T / 32), as a compile-time constant (7 for ViT-B/16's 197 tokens): with a run-time count every
// key block sits behind its own branch (14 scheduling regions per item); with the constant the item is straight-line code.
// TT: the token count as a constant too (197 for ViT-B/16: row clamps and key masks become per-thread constants), 0 = run-time T.
typedef short s16x4v __attribute__((__vector_size__(4 * sizeof(short))));
// TR: V stays ROW-major in LDS (one 16-byte write per chunk, like K) and the O^T MFMAs' A operand -- V^T[d][8 consecutive keys] -- comes
// from gfx950's transposing read: per group of 16 lanes ds_read_b64_tr_b16 takes a 4-key x 16-feature block (lane 4q + p of the group
// supplies the address of key q, features 4p .. 4p + 3) and hands lane i feature i of the four keys.
// STAG (round 5; MI355X_MICROARCH.md "two waves per SIMD", item 9): the two waves that share a SIMD (w and w + 4) run the same program and
// leave the staging barrier together, so they want the matrix pipe at the same time (S^T MFMAs), then the vector ALU at the same time
// (softmax), then the matrix pipe again.  With STAG waves 0-3 issue the NEXT item's twelve loads after their S^T MFMAs (which run on
// while the loads are issued) and waves 4-6 before theirs, as all waves did: the partners reach the softmax ~1-2 k cycles apart, one's
// vector work beside the other's matrix work.  Same instructions per wave on the same data: bit-identical output.  Alone, 337 crops, variants
// interleaved in one process: 96.0 us against 99.4 (331 crops: 94.7 / 98.2).  Measured and dropped: waves 4-6 ALSO deferring their output
// phase to the start of the next item (the guide's full recipe; accumulators kept across the barrier): 98.3 us, no better than no stagger.
template <bool TRACE = false, int NKB = 7, int TT = 0, bool TR = false, bool STAG = false>
__global__ __launch_bounds__(448) void k_attention_f16(const f16* __restrict__ qkv, f16* __restrict__ out, int T_arg,
                                                       int W, int heads, int ld, int n_items, long long* __restrict__ trace = nullptr,
                                                       int q_tiles = 7) {
    const int T = TT ? TT : T_arg;
    long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tc = 0;      // TRACE: cycles per phase, summed over this wave's items
#define AT_STAMP(i) if (TRACE) { const long long c_ = clock64(); tr[i] += c_ - tc; tc = c_; }
    extern __shared__ __attribute__((aligned(16))) char at_smem[];       // AT_LDS_BYTES: K rows | V^T rows | one 32-row tile per wave
    f16* const Ks = (f16*)at_smem;
    f16* const Vt = Ks + AT_MAXT * AT_KLD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // per-wave 32 x 64 tile (row stride AT_KLD): the wave's Q rows arrive coalesced (eight lanes per 128-byte row) and are re-read as
    // MFMA fragments (one row per lane); the output rows go the other way.  Only this wave touches it: LDS operations of one wave
    // execute in program order, no barrier needed.
    f16* const Vs = Vt;                       // TR: [AT_MAXT][AT_KLD] rows instead of the transposed [64][AT_VLD] image
    f16* const Qs = Vt + (TR ? AT_MAXT * AT_KLD : 64 * AT_VLD) + wave * 32 * AT_KLD;
    const int tr_i = lane & 15, tr_g = lane >> 4;
    const int tr_off = ((4 * (tr_g >> 1) + (tr_i >> 2)) * AT_KLD + 16 * (tr_g & 1) + 4 * (tr_i & 3)) * 2;      // bytes: + (kb*32 + 16s [+ 8]) rows, + dt*32 features
    const int crow = lane >> 3, cpart = lane & 7;                           // coalesced layout: row it * 8 + crow, 16-byte part
    const int r31 = lane & 31, hh = lane >> 5;
    const int q0 = wave * 32;
    const int q = q0 + r31;
    constexpr int nkb = NKB;                                          // == (T + 31) / 32, checked by the launcher
    static_assert(AT_MAXT * 8 == 4 * 448, "staging split");
    // PERSISTENT workgroups (182 VGPRs and 62 KB of LDS allow one 7-wave workgroup per CU, so nothing else would hide
    // an item's load latency): item = (crop, head); the NEXT item's Q / K / V rows are fetched into registers while
    // the current item is computed.  All eight 16-byte K/V loads of a thread go out back to back, branch-free
    // (clamped row, zeroed by select when written to LDS).
    f16x8 qf[4], qn[4];
    uint4 kreg[4];
    f16x8 vreg[4];
    // Per-lane element offsets inside an item, 32-bit and item-invariant (an item's rows span < T * ld elements): the item's base is a
    // wave-uniform 64-bit pointer, so a load is SGPR base + VGPR offset and nothing 64-bit per lane lives across the item loop (the
    // size_t row products did: 256 VGPRs + a spilled pair in the TR instantiation, round 3).
    unsigned qoff[4], kvoff[4], ooff[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int qr = q0 + it * 8 + crow;
        qoff[it] = (unsigned)((qr < T ? qr : T - 1) * ld + cpart * 8);
        ooff[it] = (unsigned)(qr * W + cpart * 8);
        const int c = tid + it * 448, key = c >> 3, part = c & 7;
        kvoff[it] = (unsigned)((key < T ? key : T - 1) * ld + part * 8);
    }
    auto fetch = [&](int item) {
        const int crop = item / heads, head = item - crop * heads;
        const f16* qbase = qkv + (size_t)crop * T * ld + head * 64;
        const f16* kbase = qbase + W;
        const f16* vbase = qbase + 2 * W;
#pragma unroll
        for (int it = 0; it < 4; ++it) qn[it] = *(const f16x8*)(qbase + qoff[it]);
        // K and V rows: eight lanes cover one 128-byte row (clamped row, zeroed by select when written to LDS).  For V the transpose
        // (TR = false) happens on the LDS-write side, 2-way bank-conflicted.  One key per lane (conflict-free writes, but 64 rows x 16 B
        // per load instruction) kept the waves that issue last waiting ~3.5 k cycles per item on the address path: 151 -> 146 us per
        // launch; a 16-key x 64-byte pattern (conflict-free writes, 16 rows per instruction) measured 149 us.
#pragma unroll
        for (int it = 0; it < 4; ++it) kreg[it] = *(const uint4*)(kbase + kvoff[it]);
#pragma unroll
        for (int it = 0; it < 4; ++it) vreg[it] = *(const f16x8*)(vbase + kvoff[it]);
    };
    int item = blockIdx.x;
    if (item < n_items) fetch(item);
    for (; item < n_items; item += gridDim.x) {
    if (TRACE) tc = clock64();
    const int crop = item / heads, head = item - crop * heads;
    f16* const obase = out + (size_t)crop * T * W + head * 64;
    // K -> LDS rows (zero beyond T)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = tid + it * 448, key = c >> 3, part = c & 7;
        *(uint4*)(Ks + key * AT_KLD + part * 8) = key < T ? kreg[it] : make_uint4(0, 0, 0, 0);
    }
    // V -> LDS transposed (zero beyond T)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = tid + it * 448, key = c >> 3, part = c & 7;
        const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        const f16x8 v = key < T ? vreg[it] : z;
        if (TR) *(f16x8*)(Vs + key * AT_KLD + part * 8) = v;
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) Vt[(part * 8 + e) * AT_VLD + key] = v[e];
        }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) *(f16x8*)(Qs + (it * 8 + crow) * AT_KLD + cpart * 8) = qn[it];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const f16x8*)(Qs + r31 * AT_KLD + s * 16 + hh * 8);
    __syncthreads();
    AT_STAMP(0)                                                        // K / V^T into LDS + barrier
    const bool has_next = item + (int)gridDim.x < n_items;
    const bool computes = q0 < T && wave < q_tiles;   // (q_tiles: query tiles wanted -- 1 in the last block, whose class-token row alone is used)
    const bool late = STAG && wave < 4 && computes;   // this wave issues the next item's loads behind its S^T MFMAs
    if (has_next && !late) fetch(item + gridDim.x);                    // in flight during the compute below
    AT_STAMP(1)                                                        // issue of the next item's loads
    if (computes) {


    f32x16 sacc[7];
#pragma unroll
    for (int kb = 0; kb < 7; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
        if (kb < nkb) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f16x8 kf = *(const f16x8*)(Ks + (kb * 32 + r31) * AT_KLD + s * 16 + hh * 8);
                sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sacc[kb], 0, 0, 0);
            }
        }
    }
    if (STAG && has_next && late) fetch(item + gridDim.x);             // (the matrix pipe works on the 28 MFMAs meanwhile)
    AT_STAMP(2)                                                        // S^T MFMAs issued
    // softmax over keys (scores scaled by 1/8 = dh^-0.5, model.py via nn.MultiheadAttention).  Only the last key block
    // can hold keys >= T; blocks beyond it were never computed (zeros) and are skipped below.
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 7; ++kb) {
        if (kb == nkb - 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (key >= T) sacc[kb][r] = -INFINITY;
            }
        }
        if (kb < nkb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    AT_STAMP(3)                                                        // max pass (waits for the MFMA results)
    const float c2 = 0.125f * 1.4426950408889634f;
    const float mc = -mx * c2;
    float sum = 0.f;
    // keys of the LAST block that exist: register group g (4 registers) holds keys 8g .. 8g+7 of the block (both lane halves), so
    // a group with 8g >= tail is -inf throughout -> p = 0 without an exp (197 tokens: 12 of the block's 16 registers)
    const int tail = T - 32 * (nkb - 1);
#pragma unroll
    for (int kb = 0; kb < 7; ++kb) {
        if (kb < nkb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (kb < nkb - 1 || 8 * g < tail) {
#pragma unroll
                    for (int r = 4 * g; r < 4 * g + 4; ++r) {
                        const float p = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r], c2, mc));      // raw v_exp_f32: exp2(-inf) = 0
                        sacc[kb][r] = p;
                        sum += p;
                    }
                } else {
#pragma unroll
                    for (int r = 4 * g; r < 4 * g + 4; ++r) sacc[kb][r] = 0.f;
                }
            }
        }
    }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    AT_STAMP(4)                                                        // exp pass

    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 7; ++kb) {
        if (kb < nkb) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (kb == nkb - 1 && 16 * s >= tail) continue;         // k-step of keys that do not exist: P = 0 there, adds exact zeros
                f16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (f16)sacc[kb][8 * s + j];
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    // A[d][k]: element j <-> key kb*32 + 16s + 8(j>>2) + 4hh + (j&3)
                    f16x4 lo, hi;
                    if (TR) {
                        const char* vb = (const char*)Vs + tr_off + ((kb * 32 + 16 * s) * AT_KLD + dt * 32) * 2;
                        lo = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)vb));
                        hi = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(vb + 8 * AT_KLD * 2)));
                    } else {
                        const f16* vp = Vt + (dt * 32 + r31) * AT_VLD + kb * 32 + 16 * s + 4 * hh;
                        lo = *(const f16x4*)vp; hi = *(const f16x4*)(vp + 8);
                    }
                    f16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[dt], 0, 0, 0);
                }
            }
        }
    }
    AT_STAMP(5)                                                        // P -> fp16, V^T fragments, O^T MFMAs issued
    // output rows through the wave's tile: a lane's 4-feature pieces in, 16-byte parts of whole 128-byte rows out
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f16x4 h4;
#pragma unroll
            for (int e = 0; e < 4; ++e) h4[e] = (f16)(oacc[dt][4 * g + e] * inv);
            *(f16x4*)(Qs + r31 * AT_KLD + dt * 32 + 8 * g + 4 * hh) = h4;
        }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int qr = q0 + it * 8 + crow;
        const f16x8 v = *(const f16x8*)(Qs + (it * 8 + crow) * AT_KLD + cpart * 8);
        if (qr < T) *(f16x8*)(obase + ooff[it]) = v;
    }
    }
    AT_STAMP(6)                                                        // scaled output (waits for the O^T MFMAs) + stores issued
    __syncthreads();      // every wave is done with this item's K / V^T before the next item overwrites them
    AT_STAMP(7)                                                        // barrier at the end of the item
    }
    if (TRACE && trace && lane == 0) {
        long long* o = trace + ((size_t)blockIdx.x * 7 + wave) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = tr[i];
    }
#undef AT_STAMP
}

// Measured and dropped (round 4): k_attention_f16_dma -- K and V double-buffered in LDS and filled by LDS-DMA (8-row x 128-byte pieces,
// chunk swizzle by the reversed bits of row >> 1: conflict-free for the K fragment reads and the transposing V reads), the next item
// requested under this item's MFMAs and softmax, no staging phase, one barrier per item, 207 instead of 249 VGPRs.  Two things the
// compiler needed: the transposing reads as inline asm (through the builtin it waits vmcnt(0) in front of the first one: it cannot see
// that the DMA in flight fills the other buffers) and a raw s_barrier (__syncthreads adds vmcnt(0) for the output stores).
// Bit-identical to this kernel; 111.5 vs 114 us per launch alone (337 crops, two interleaved pairs), 65.2 vs 65.2 frames/s in the
// pipeline.  With the whole staging phase gone the item is as long as before: it is bound by the two waves that share a SIMD (seven
// waves on four SIMDs: 2 : 2 : 2 : 1) issuing their softmax and MFMA streams one after the other, not by getting K and V into LDS.
// fp32 parity-mode attention: one workgroup per (crop, head); K,V in LDS; one wave per query row at a time.
__global__ __launch_bounds__(256) void k_attention_f32(const float* __restrict__ qkv, float* __restrict__ out, int T,
                                                       int W, int heads) {
    extern __shared__ float sm[];
    float* Ks = sm;                 // [T][65]
    float* Vs = sm + (size_t)T * 65;  // [T][64]
    float* Ps = Vs + (size_t)T * 64;  // [4 waves][4 rows][T]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int crop = blockIdx.x / heads, head = blockIdx.x - crop * heads;
    const size_t row0 = (size_t)crop * T;
    const int ld = 3 * W;
    const float* qbase = qkv + row0 * ld + head * 64;
    for (int c = tid; c < T * 64; c += 256) {
        int key = c >> 6, d = c & 63;
        Ks[key * 65 + d] = qbase[(size_t)key * ld + W + d];
        Vs[key * 64 + d] = qbase[(size_t)key * ld + 2 * W + d];
    }
    __syncthreads();
    // Four query rows per wave at a time (round 4; one row at a time spent an LDS read on every multiply-add: 2.2 ms per layer for 64
    // crops, half of the fp32 tower): a K / V element read from LDS serves four rows, the query and probability values come out of
    // registers as wave-uniform scalars (v_readlane).  Per (row, key) the same fmaf chain over d, per (row, feature) the same chain over
    // the keys as before: the numbers do not change.
    constexpr int QB = 4;
    float* P = Ps + wave * QB * T;
    for (int q0 = wave * QB; q0 < T; q0 += 4 * QB) {
        float qd[QB];
#pragma unroll
        for (int j = 0; j < QB; ++j) qd[j] = q0 + j < T ? qbase[(size_t)(q0 + j) * ld + lane] * 0.125f : 0.f;   // q scaled before QK^T like nn.MultiheadAttention
        float mx[QB];
#pragma unroll
        for (int j = 0; j < QB; ++j) mx[j] = -INFINITY;
        for (int k0 = 0; k0 < T; k0 += 64) {
            const int key = k0 + lane;
            const float* kr = Ks + (key < T ? key : T - 1) * 65;
            float sacc[QB] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int d = 0; d < 64; ++d) {
                const float kv = kr[d];
#pragma unroll
                for (int j = 0; j < QB; ++j)
                    sacc[j] = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qd[j]), d)), kv, sacc[j]);
            }
            if (key < T) {
#pragma unroll
                for (int j = 0; j < QB; ++j) {
                    P[j * T + key] = sacc[j];
                    mx[j] = fmaxf(mx[j], sacc[j]);
                }
            }
        }
        float sum[QB];
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            mx[j] = vg_wave_max(mx[j]);
            float sj = 0.f;
            for (int key = lane; key < T; key += 64) {
                const float p = expf(P[j * T + key] - mx[j]);
                P[j * T + key] = p;
                sj += p;
            }
            sum[j] = vg_wave_sum(sj);
        }
        __builtin_amdgcn_s_waitcnt(0);
        float o[QB] = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < T; k0 += 64) {
            float pv[QB];
#pragma unroll
            for (int j = 0; j < QB; ++j) pv[j] = k0 + lane < T ? P[j * T + k0 + lane] : 0.f;
            const int nk = T - k0 < 64 ? T - k0 : 64;
            if (nk == 64) {
#pragma unroll
                for (int kk = 0; kk < 64; ++kk) {
                    const float vv = Vs[(k0 + kk) * 64 + lane];
#pragma unroll
                    for (int j = 0; j < QB; ++j)
                        o[j] = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pv[j]), kk)), vv, o[j]);
                }
            } else {
                for (int kk = 0; kk < nk; ++kk) {
                    const float vv = Vs[(k0 + kk) * 64 + lane];
#pragma unroll
                    for (int j = 0; j < QB; ++j) o[j] = fmaf(__shfl(pv[j], kk), vv, o[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < QB; ++j)
            if (q0 + j < T) out[(row0 + q0 + j) * (size_t)W + head * 64 + lane] = o[j] / sum[j];
    }
}

// ---------------------------------------------------------------------------------------------
// head: ln_post(x[crop, 0, :]) @ proj  (model.py:235-238).  one workgroup per crop.
template <typename TI>
__global__ __launch_bounds__(256) void k_head(const TI* __restrict__ x, const float* __restrict__ lw,
                                              const float* __restrict__ lb, const float* __restrict__ proj,
                                              float* __restrict__ feat, int T, int W, int D) {
    extern __shared__ float sm[];   // [W]
    __shared__ float red[8];
    const int crop = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const TI* src = x + (size_t)crop * T * W;
    float s = 0.f;
    for (int c = tid; c < W; c += 256) s += (float)src[c];
    s = vg_wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    float mean = (red[0] + red[1] + red[2] + red[3]) / (float)W;
    __syncthreads();
    float q = 0.f;
    for (int c = tid; c < W; c += 256) {
        float d = (float)src[c] - mean;
        q += d * d;
    }
    q = vg_wave_sum(q);
    if (lane == 0) red[wave] = q;
    __syncthreads();
    float rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)W + 1e-5f);
    for (int c = tid; c < W; c += 256) sm[c] = ((float)src[c] - mean) * rstd * lw[c] + lb[c];
    __syncthreads();
    for (int j = tid; j < D; j += 256) {
        float a = 0.f;
#pragma unroll 16                                  // 16 independent loads of the projection column in flight per trip (the chain
        for (int i = 0; i < W; ++i)               // a = fma(.., a) stays in order: same sum as before)
            a = fmaf(sm[i], proj[(size_t)i * D + j], a);
        feat[(size_t)crop * D + j] = a;
    }
}

// clip_utils.py:42-61: f /= |f|; probs = softmax(100 f T^T); top-1.  one wave per crop, K <= 64 classes.
__global__ __launch_bounds__(64) void k_clip_scores(const float* __restrict__ feat, const float* __restrict__ text,
                                                    float* __restrict__ probs, int* __restrict__ top1,
                                                    float* __restrict__ top1_score, int n, int D, int Kc) {
    const int crop = blockIdx.x, lane = threadIdx.x;
    const float* f = feat + (size_t)crop * D;
    float ss = 0.f;
    for (int i = lane; i < D; i += 64) ss += f[i] * f[i];
    float nrm = sqrtf(vg_wave_sum(ss));
    float logit = -INFINITY;
    if (lane < Kc) {
        float a = 0.f;
        for (int i = 0; i < D; ++i) a = fmaf(100.0f * (f[i] / nrm), text[(size_t)lane * D + i], a);
        logit = a;
    }
    float mx = vg_wave_max(logit);
    float e = (lane < Kc) ? expf(logit - mx) : 0.f;
    float p = e / vg_wave_sum(e);
    if (lane < Kc) probs[(size_t)crop * Kc + lane] = p;
    // argmax, lowest index on ties
    float best = (lane < Kc) ? p : -1.f;
    int bi = lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o);
        int oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) {
            best = ob;
            bi = oi;
        }
    }
    if (lane == 0) {
        top1[crop] = bi;
        top1_score[crop] = best;
    }
}

// ---------------------------------------------------------------------------------------------
#define VG_PROF_MAX 4096
struct vg_vit {
    int width, layers, heads, patch, res, out_dim, dtype, T;
#ifdef VG_DEV
    bool gemm_x2 = getenv("VG_GEMM_X2") ? atoi(getenv("VG_GEMM_X2")) != 0 : false;   // projection GEMMs by k_gemm_f16_x2 (two workgroups per CU)
#endif
    bool cls_last = !(getenv("VG_VIT_CLS_LAST") && atoi(getenv("VG_VIT_CLS_LAST")) == 0);   // last block: class-token rows only (see vg_vit_encode)
    // A/B switches every test exercises (tests/test_vit.py, test_gemm.py), read ONCE when the handle is made -- never per launch: getenv
    // is not safe against a concurrent setenv, the worker threads launch concurrently, and a captured graph must replay what the plain
    // launches of the same handle run.  The handle-less entry points (vg_gemm, vg_gemm_resid_splitk, vg_attention) build a
    // temporary handle per call, i.e. read them per call on the caller's thread.
    int splitk_max = getenv("VG_GEMM_SPLITK") ? atoi(getenv("VG_GEMM_SPLITK")) : 0;          // opt-in split-K tail of the residual GEMMs (splitk_plan)
    bool att_tr = !(getenv("VG_ATT_TR") && atoi(getenv("VG_ATT_TR")) == 0);                    // attention: row-major V + transposing LDS reads
    bool att_stagger = !(getenv("VG_ATT_STAGGER") && atoi(getenv("VG_ATT_STAGGER")) == 0);     // attention: SIMD partners apart in phase (k_attention_f16 STAG)
    bool f32_mfma = !(getenv("VG_GEMM_F32_MFMA") && atoi(getenv("VG_GEMM_F32_MFMA")) == 0);    // fp32 tower on the matrix cores
    int gemm_w4 = getenv("VG_GEMM_W4") ? atoi(getenv("VG_GEMM_W4")) : 1;                      // projection GEMMs by k_gemm_f16_w4 (round 6: persistent, 4 waves, assembly K loop); 0: k_gemm_f16_pp64 (a tower uses one family: their LayerNorm partials differ in granularity)
    int n_cu = 0;                    // compute units of the device the handle works on (set at the first launch)
    bool resid_hl = false;           // the residual stream kept as an fp16 PAIR (EPI_BIAS_RESID_HL; round 6, default for the k_gemm_f16_w4 tower with the folded
                                     // LayerNorm and the last block on the class rows; VG_VIT_RESID_HL=0: the fp32 stream): 22 bits of the stream, a third less
                                     // written by out_proj / c_proj -- features 1.8e-4 from the fp32-stream tower's, the same distance to the fp32 tower
    bool resid_h = false;            // opt-in (VG_VIT_RESID16=1, dtype 1, width % 256 == 0): fp16 residual stream like upstream's fp16 run.
                                     // +2.7 % frames/s, 3x the feature error (1.1e-3 vs 3.4e-4 rel. L2): default keeps the fp32 stream
    // optional per-launch timing of the projection GEMMs (bench.py roofline): event pairs on the launch stream
    int prof_on = 0, prof_n = 0;
    hipEvent_t prof_ev[2 * VG_PROF_MAX];
    double prof_flops[VG_PROF_MAX];
    int prof_kind[VG_PROF_MAX];      // 0 = k_gemm_f16 / k_gemm_f32, 1 = k_gemm_f16_pp64 / pp16
    bool prof_init = false;
    std::map<std::string, void*> w;        // device pointers (f32 or f16 depending on role)
    std::map<std::string, size_t> numel;
    // LayerNorm folded into the GEMMs around it (k_gemm_f16_pp64 LN = 1 / 2): fp16 tower with the fp32 residual stream and
    // width % 256 == 0, unless VG_VIT_LN_FOLD=0.  The fp32 originals of in_proj / c_fc stay on the device (w32) so that the
    // pre-scaled fp16 weights are rounded once, from g[k] * W[n,k]; the derived tensors live in `w` under "<name>#ln" (weights),
    // "#c1", "#c2" and are rebuilt when any of their inputs is set again.
    bool ln_fold = false;
    std::atomic<bool> fold_ready{false}, warmed{false};
    // single-channel patch rows (vg_vit_encode input_kind 3): per-channel input normalisation the fold assumes (clip.py:79-86; set by
    // vg_vit_set_input_norm) and the readiness of "conv1.weight#1ch" / "positional_embedding#1ch" (vit_fold_conv1)
    float in_mean[3] = {0.48145466f, 0.4578275f, 0.40821073f}, in_std[3] = {0.26862954f, 0.26130258f, 0.27577711f};
    std::atomic<bool> conv1_ready{false};
    // bumped whenever a device tensor the kernels read is replaced (vg_vit_set_weight, vit_fold_ln): part of the captured graphs'
    // key, so a graph that holds pointers to freed weights is never replayed
    std::atomic<uint64_t> weights_gen{0};
    std::mutex mtx;
    std::map<std::string, void*> w32;
};

static bool is_gemm_weight(const std::string& n) {
    return n == "conv1.weight" || n.find("in_proj_weight") != std::string::npos ||
           n.find("out_proj.weight") != std::string::npos || n.find("c_fc.weight") != std::string::npos ||
           n.find("c_proj.weight") != std::string::npos;
}

// column tiles per L2 chunk: the largest divisor-friendly count whose weight rows (128*K fp16 each) fit ~2.4 MB
static int gemm_chunk_tiles(int N, int K) {
    const int ntn = N / GBN;
#ifdef VG_DEV
    if (getenv("VG_GEMM_NO_CHUNK")) return ntn;        // tile-order sweep (development build)
#endif
    const long tile_bytes = (long)GBN * K * 2;
    int cw_max = (int)(2400000L / tile_bytes);
    if (cw_max < 1) cw_max = 1;
    if (cw_max >= ntn) return ntn;
    int nchunks = (ntn + cw_max - 1) / cw_max;
    while (ntn % nchunks) ++nchunks;
    return ntn / nchunks;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifdef VG_DEV      // K-step-32 predecessor of k_gemm_f16_pp64 (tools/dev only)
#include "dev/vit_gemm_pp16.inc"
#endif  // VG_DEV

// ---------------------------------------------------------------------------------------------
// K-step 64 (128-byte rows).  Measured with tools/micro/dma_rate.hip: LDS-DMA from L2-resident data streams at
// 39 B/clk/CU when a piece covers 16 rows x 64 B and at 64 B/clk/CU (the L1 peak) when it covers 8 rows x 128 B -- a
// 256 x 256 x 32 K-step needs 32 B/clk/CU at full MFMA rate, so 64-byte row segments leave the LOAD segment the
// longer one of the ping-pong pair.  Here a phase is one 64-wide K-tile (two k32 sub-steps, 64 MFMAs per wave):
//   LDS (all 160 KB): X ring 2 x 32 KB + W ring 3 x 32 KB; a row is 128 B, 16-byte chunk c of row r sits at c ^ ((r>>1)&7).
//   Group g reads X rows [128 g, +128) only -> it refills them itself (K-tile j+1 in LOAD_j, its slot was last read
//   in the group's own MMA_{j-1}).  W rows are read by both groups; group 1, which runs one barrier behind, is the
//   last reader of a W slot: W of K-tile j+2 goes into the slot of K-tile j-1, half by group 1 in its LOAD_j and half
//   by group 0 in its LOAD_{j+1} (3-deep ring -> at least one interval of lead).
//   The second k32 sub-step's fragments are read during the first sub-step's MFMAs into the registers those have
//   just consumed.  Every wave waits for its own pieces (vmcnt(0)) before the barrier that ends its MMA segment.
//
// LayerNorm folded into the GEMMs around it (LN = 1 / 2; ViT blocks, fp32 residual stream).  ln(x) W^T + b with
// ln(x) = (x - mean) * rstd * g + beta is  rstd * (x (g.W)^T - mean * c1) + c2,  c1[n] = sum_k g[k] W[n,k],
// c2[n] = b[n] + sum_k beta[k] W[n,k]:  the CONSUMER (LN = 1: in_proj, c_fc) multiplies the raw residual (as fp16) by the
// pre-scaled weights and applies the row statistics in its epilogue; the PRODUCER (LN = 2: out_proj, c_proj, whose epilogue
// holds the new residual row segments anyway) writes that fp16 copy and, per row and 256-column tile, (mean, sum of squared
// deviations from it) -- the consumer merges the K / 256 partials of a row (Chan et al.), so the variance is the two-pass
// variance, not E[x^2] - mean^2.  Saves the separate LayerNorm pass over the residual stream (295 MB per launch, 23 per frame).
struct LnPartial { float mean, m2; };
__device__ __forceinline__ float row256_sum(float v) {          // sum over the 64 lanes, fixed order: DPP inside rows of 16, then the four rows
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    const int b = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}

// Measured and dropped (round 4): "RI" -- every lane loads the residual elements of its accumulator registers in the prologue, in
// front of the first LDS-DMA pieces, the MFMAs accumulate on top and the epilogue is write-only.  Launches interleaved on one box at
// M = 66560: out_proj 188 vs 177 us and c_proj 382 vs 366 us with the residual stream out of the memory-side cache (as in the
// pipeline), 157 vs 156 / 362 vs 360 with it cached; whole pipeline 65.1 vs 65.0 frames/s.  The residual epilogue is bound by the bytes
// all CUs move at the same time (510 MB per launch), not by the latency of its load -> add -> store rounds.
//
// SPLITK (round 4; EPI_NONE_F32 only): the workgroup computes K-tiles [part np / P, (part + 1) np / P) of tile blockIdx.x / P (row-major
// over the M / 256 x N / 256 tiles of THIS launch) and stores the raw fp32 accumulators as a dense 256 x 256 image at
// Cout + blockIdx.x * 65536 -- the partial sums of a split-K tail (launch_gemm_resid_tail below; k_splitk_resid adds them up).
template <int EPI, bool TRACE = false, bool PERSIST = false, int LN = 0, bool SPLITK = false>
__global__ __launch_bounds__(512, 1) void k_gemm_f16_pp64(const f16* __restrict__ X, const f16* __restrict__ Wt,
                                                          const float* __restrict__ bias, void* __restrict__ Cout,
                                                          float* __restrict__ resid, int M, int N, int K, int ldc, int cw,
                                                          long long* __restrict__ trace = nullptr,
                                                          const float* __restrict__ ln_c1 = nullptr, LnPartial* __restrict__ ln_stats = nullptr,
                                                          f16* __restrict__ ln_x16 = nullptr, int sk_parts = 1) {
    static_assert(!SPLITK || (EPI == EPI_NONE_F32 && !PERSIST && LN == 0), "split-K partials are raw fp32 tiles");
    constexpr int BM = 256, BN = 256, NT = 512, TM = 8, TN = 4;
    constexpr int XBUF = 32768, WBASE = 2 * XBUF, WBUF = 32768;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave id in an SGPR
    const int ntm = M / BM;
    const int grp = wave >> 2, wn = wave & 3;
    long long tr_entry = 0, tr_load = 0, tr_bar = 0, tr_mma = 0, tr_wait = 0, tr_t0 = 0, tr_w0 = 0, tr_main = 0, tr_w_entry = 0;
    if (TRACE) { tr_entry = clock64(); tr_w_entry = wall_clock64(); }
    // PERSIST: one workgroup per CU walks its XCD's contiguous run of tiles (slot s of G/8 takes tiles s, s + G/8, ...), which
    // removes the 1-2 us a CU idles between two workgroups; nothing of consecutive tiles overlaps (the registers are full).
    const int nt_all = ntm * (N / BN);
    const int xq = nt_all >> 3, xr_ = nt_all & 7, xcd = blockIdx.x & 7;
    const int xbase = (xcd < xr_) ? xcd * (xq + 1) : xr_ * (xq + 1) + (xcd - xr_) * xq;
    const int xcnt = xq + (xcd < xr_ ? 1 : 0);
    for (int ti = PERSIST ? (int)(blockIdx.x >> 3) : 0; ti < (PERSIST ? xcnt : 1); ti += PERSIST ? (int)(gridDim.x >> 3) : 1) {
    int lane = tid & 63;
    if (PERSIST) asm volatile("" : "+v"(lane));      // per-lane offsets are re-derived per tile instead of living through the epilogue
    const int t = PERSIST ? xbase + ti : xcd_remap(blockIdx.x, gridDim.x);
    const int per_chunk = ntm * cw;
    const int chunk = t / per_chunk, tc = t - chunk * per_chunk;
    int tm = tc / cw, tn = chunk * cw + (tc - tm * cw);
    int kt0 = 0, np = K / 64;                      // host guarantees K % 64 == 0 and np >= 2 (per part when SPLITK)
    if (SPLITK) {
        const int st_ = (int)blockIdx.x / sk_parts, part = (int)blockIdx.x - st_ * sk_parts, ntn_ = N / BN;
        tm = st_ / ntn_; tn = st_ - tm * ntn_;
        kt0 = part * np / sk_parts;
        np = (part + 1) * np / sk_parts - kt0;
    }
    const int m0 = tm * BM, n0 = tn * BN;

    // DMA pieces: 8 rows x 128 B; lane -> row l >> 3, chunk slot l & 7 (source chunk = slot ^ ((row >> 1) & 7))
    const int prow = lane >> 3, pslot = lane & 7;
    const int xr = grp * 128 + wn * 32 + prow;                         // + 8 i, i = 0..3
    auto issue_x = [&](int kt) {                                       // this group's X half of K-tile kt
        char* d = smem + (kt & 1) * XBUF + (grp * 128 + wn * 32) * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = xr + 8 * i;
            const f16* sp = X + (size_t)(m0 + r) * K + (size_t)(kt0 + kt) * 64 + (pslot ^ ((r >> 1) & 7)) * 8;
            __builtin_amdgcn_global_load_lds((glb_void*)sp, (lds_void*)(d + i * 1024), 16, 0, 0);
        }
    };
    // W rows [128 h, +128) of K-tile kt, four pieces per wave.  Half 0 is issued by group 1 in LOAD_{kt-2}, half 1 by
    // group 0 in LOAD_{kt-1} (both after the slot's last reader, group 1's MMA_{kt-3}, has passed its barrier), which
    // balances the DMA work: 8 pieces per wave and phase in either group.
    auto issue_w = [&](int kt, int h) {
        char* d = smem + WBASE + (kt % 3) * WBUF + (h * 128 + wn * 32) * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = h * 128 + wn * 32 + prow + 8 * i;
            const f16* sp = Wt + (size_t)(n0 + r) * K + (size_t)(kt0 + kt) * 64 + (pslot ^ ((r >> 1) & 7)) * 8;
            __builtin_amdgcn_global_load_lds((glb_void*)sp, (lds_void*)(d + i * 1024), 16, 0, 0);
        }
    };
    f32x4 acc[TN][TM];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r15 = lane & 15, q4 = lane >> 4;
    const int swz = (r15 >> 1) & 7;
    const int xo0 = (grp * 128 + r15) * 128 + ((q4 ^ swz) << 4), xo1 = (grp * 128 + r15) * 128 + (((4 + q4) ^ swz) << 4);
    const int wo0 = (wn * 64 + r15) * 128 + ((q4 ^ swz) << 4), wo1 = (wn * 64 + r15) * 128 + (((4 + q4) ^ swz) << 4);
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;     // LDS byte address of the rings (inline-asm reads)
    // folded LayerNorm, consumer side: thread t < 256 merges the K / 256 partials of tile row t into (mean, rstd) here, where the
    // loads' latency hides behind the prologue's DMA (in the epilogue it cost ~7 us per tile), and carries two registers
    float2 ln_row = make_float2(0.f, 0.f);
    LnPartial pt[4] = {};
    const int nst = K >> 8;                        // <= 4 (width <= 1024)
    if (LN == 1 && tid < BM) {
        const LnPartial* sp = ln_stats + (size_t)(m0 + tid) * nst;
#pragma unroll
        for (int t = 0; t < 4; ++t) pt[t] = t < nst ? sp[t] : LnPartial{0.f, 0.f};
    }

#define PP_BAR()                                  \
    __builtin_amdgcn_sched_barrier(0);            \
    __builtin_amdgcn_s_barrier();                 \
    __builtin_amdgcn_sched_barrier(0);
    issue_x(0);
    if (grp == 1) { issue_w(0, 0); issue_w(1, 0); } else issue_w(0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (LN == 1 && tid < BM) {                     // the partials were requested before the first pieces: no extra wait here
        float ms = 0.f, m2 = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) { ms += pt[t].mean; m2 += pt[t].m2; }
        const float mean = ms / (float)nst;
        float dev = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) if (t < nst) { const float d = pt[t].mean - mean; dev += d * d; }
        ln_row = make_float2(mean, rsqrtf((m2 + 256.f * dev) / (float)K + 1e-5f));
    }
    PP_BAR()
    if (grp == 1) { PP_BAR() }
    f16x8 fa[TN], fa2[TN], fb[TM];
    if (TRACE) { tr_t0 = clock64(); tr_w0 = wall_clock64(); }
    for (int j = 0; j < np; ++j) {
        const char* xb = smem + (j & 1) * XBUF;
        const char* wb = smem + WBASE + (j % 3) * WBUF;
        long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
        if (TRACE) c0 = clock64();
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) fa[ni] = *(const f16x8*)(wb + wo0 + ni * 2048);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) fb[mi] = *(const f16x8*)(xb + xo0 + mi * 2048);
        if (j + 1 < np) issue_x(j + 1);
        if (grp == 1) { if (j + 2 < np) issue_w(j + 2, 0); }
        else if (j + 1 < np) issue_w(j + 1, 1);
        if (TRACE) c1 = clock64();
        PP_BAR()
        if (TRACE) c2 = clock64();
        __builtin_amdgcn_s_setprio(1);
        // The second k32 sub-step's fragments are read DURING the first sub-step's MFMAs, by inline-asm ds_read_b128 with counted waits
        // (round 4).  Written as plain loads hipcc put `s_waitcnt lgkmcnt(0)` in front of the segment's first MFMA (right behind the four
        // W reads it had just issued: their whole LDS latency exposed) and again in front of the second sub-step (behind the x reload it
        // had issued three MFMAs earlier): MMA segment 1 180 cycles for 1 024 cycles of MFMAs.  Now: MFMA group 0 starts at once (its
        // operands were read in the LOAD segment), the four W reads follow it, the x fragment mi is re-read two groups after its last use
        // (a read into a register an in-flight MFMA still reads stalls the issue), and every group of the second sub-step waits only for
        // the reads it consumes -- LDS returns in order, so `lgkmcnt(n)` with n = the reads issued after them.  The wait names its
        // registers as operands: the MFMAs that consume them cannot be scheduled in front of it (guide 5.4 rule 18).  Same MFMAs on
        // the same operands in the same order: the same numbers.  Interleaved with the compiler-placed form on one box (M = 66 560): in_proj 247.8 vs
        // 258.3 us, c_fc + QuickGELU 327.9 vs 337.0, out_proj 136.3 vs 138.1, c_proj 340.7 vs 348.3; 13.01 vs 13.33 ms of GEMMs per frame.
        const unsigned xa1 = lds0 + (unsigned)((j & 1) * XBUF + xo1), wa1 = lds0 + (unsigned)(WBASE + (j % 3) * WBUF + wo1);
#define PP_DS128(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define PP_GROUP(FA, MI)                                                                                              \
        _Pragma("unroll") for (int ni = 0; ni < TN; ++ni)                                                             \
            acc[ni][MI] = __builtin_amdgcn_mfma_f32_16x16x32_f16(FA[ni], fb[MI], acc[ni][MI], 0, 0, 0);
        PP_GROUP(fa, 0)
        PP_DS128(fa2[0], wa1, 0); PP_DS128(fa2[1], wa1, 2048); PP_DS128(fa2[2], wa1, 4096); PP_DS128(fa2[3], wa1, 6144);
        __builtin_amdgcn_sched_barrier(0);
        PP_GROUP(fa, 1)
        __builtin_amdgcn_sched_barrier(0);
        PP_GROUP(fa, 2) PP_DS128(fb[0], xa1, 0);          __builtin_amdgcn_sched_barrier(0);
        PP_GROUP(fa, 3) PP_DS128(fb[1], xa1, 2048);       __builtin_amdgcn_sched_barrier(0);
        PP_GROUP(fa, 4) PP_DS128(fb[2], xa1, 2 * 2048);   __builtin_amdgcn_sched_barrier(0);
        PP_GROUP(fa, 5) PP_DS128(fb[3], xa1, 3 * 2048);   __builtin_amdgcn_sched_barrier(0);
        PP_GROUP(fa, 6) PP_DS128(fb[4], xa1, 4 * 2048);   __builtin_amdgcn_sched_barrier(0);
        PP_GROUP(fa, 7) PP_DS128(fb[5], xa1, 5 * 2048);   __builtin_amdgcn_sched_barrier(0);
        // reads in flight, oldest first: fa2[0..3], fb[0..5]; fb[6], fb[7] follow groups 0 and 1 of the second sub-step
        asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(fa2[0]), "+v"(fa2[1]), "+v"(fa2[2]), "+v"(fa2[3]), "+v"(fb[0]));
        PP_GROUP(fa2, 0) PP_DS128(fb[6], xa1, 6 * 2048);  __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(fb[1]));
        PP_GROUP(fa2, 1) PP_DS128(fb[7], xa1, 7 * 2048);  __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(fb[2]));
        PP_GROUP(fa2, 2) __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fb[3]));
        PP_GROUP(fa2, 3) __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fb[4]));
        PP_GROUP(fa2, 4) __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fb[5]));
        PP_GROUP(fa2, 5) __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fb[6]));
        PP_GROUP(fa2, 6) __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[7]));
        PP_GROUP(fa2, 7) __builtin_amdgcn_sched_barrier(0);
#undef PP_GROUP
#undef PP_DS128
        __builtin_amdgcn_s_setprio(0);
        if (TRACE) c3 = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // my pieces issued in this phase have landed
        if (TRACE) c4 = clock64();
        PP_BAR()
        if (TRACE) { const long long c5 = clock64(); tr_load += c1 - c0; tr_bar += (c2 - c1) + (c5 - c4); tr_mma += c3 - c2; tr_wait += c4 - c3; }
    }
    if (TRACE) tr_main += clock64() - tr_t0;
    if (grp == 0) { PP_BAR() }
#undef PP_BAR
    // ---- epilogue: the same chunk-XOR-swizzled LDS image as k_gemm_f16_pp ----
    __syncthreads();
    if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RESID_H) {
        float ln_mean[TM], ln_rstd[TM];
        if (LN == 1) {                      // the tile's 256 (mean, rstd) pairs go through LDS (above the 128 KB output image)
            float2* lsm = (float2*)(smem + 131072);
            if (tid < BM) lsm[tid] = ln_row;
            __syncthreads();
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                const float2 t2 = lsm[grp * 128 + mi * 16 + r15];
                ln_mean[mi] = t2.x; ln_rstd[mi] = t2.y;
            }
        }
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int nloc = wn * 64 + ni * 16 + 4 * q4;
            const float4 b4 = *(const float4*)(bias + n0 + nloc);
            float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (LN == 1) c4 = *(const float4*)(ln_c1 + n0 + nloc);
            const int ch = nloc >> 3, hf = (nloc >> 2) & 1;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                const int m = grp * 128 + mi * 16 + r15;
                float v[4] = {acc[ni][mi][0] + b4.x, acc[ni][mi][1] + b4.y, acc[ni][mi][2] + b4.z, acc[ni][mi][3] + b4.w};
                if (LN == 1) {
                    v[0] = ln_rstd[mi] * (acc[ni][mi][0] - ln_mean[mi] * c4.x) + b4.x;
                    v[1] = ln_rstd[mi] * (acc[ni][mi][1] - ln_mean[mi] * c4.y) + b4.y;
                    v[2] = ln_rstd[mi] * (acc[ni][mi][2] - ln_mean[mi] * c4.z) + b4.z;
                    v[3] = ln_rstd[mi] * (acc[ni][mi][3] - ln_mean[mi] * c4.w) + b4.w;
                }
                f16x4 h4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = v[e];
                    if (EPI == EPI_BIAS_GELU) x = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * QGELU_C));
                    h4[e] = (f16)x;
                }
                *(f16x4*)(smem + m * 512 + ((ch ^ (m & 31)) << 4) + hf * 8) = h4;
            }
        }
        __syncthreads();
        const int j = tid & 31, rr = tid >> 5;
        if (EPI == EPI_BIAS_RESID_H) {
            // fp16 residual stream, updated in place: x = f16(x + f16(acc + bias)) -- the arithmetic of upstream's fp16 run
            // (model.py:190-191 on half tensors).  Eight residual loads go out before the first store.
            f16* xr = (f16*)resid;
#pragma unroll
            for (int p8 = 0; p8 < 2; ++p8) {
                f16x8 x8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int m = (p8 * 8 + q) * 16 + rr;
                    x8[q] = *(const f16x8*)(xr + (size_t)(m0 + m) * ldc + n0 + ((j ^ (m & 31)) << 3));
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int m = (p8 * 8 + q) * 16 + rr;
                    const f16x8 v = *(const f16x8*)(smem + m * 512 + j * 16);
                    f16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (f16)((float)x8[q][e] + (float)v[e]);
                    *(f16x8*)(xr + (size_t)(m0 + m) * ldc + n0 + ((j ^ (m & 31)) << 3)) = o;
                }
            }
        } else {
#pragma unroll 4
            for (int pass = 0; pass < 16; ++pass) {
                const int m = pass * 16 + rr;
                const f16x8 v = *(const f16x8*)(smem + m * 512 + j * 16);
                *(f16x8*)((f16*)Cout + (size_t)(m0 + m) * ldc + n0 + ((j ^ (m & 31)) << 3)) = v;
            }
        }
    } else {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half) __syncthreads();
            if (grp == half) {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    const int nloc = wn * 64 + ni * 16 + 4 * q4;
                    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (EPI == EPI_BIAS_RESID) b4 = *(const float4*)(bias + n0 + nloc);
                    const int ch = nloc >> 2;
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi) {
                        const int m = mi * 16 + r15;
                        *(float4*)(smem + m * 1024 + ((ch ^ (m & 31)) << 4)) =
                            make_float4(acc[ni][mi][0] + b4.x, acc[ni][mi][1] + b4.y, acc[ni][mi][2] + b4.z, acc[ni][mi][3] + b4.w);
                    }
                }
            }
            __syncthreads();
            const int j = tid & 63, rr = tid >> 6;
#pragma unroll
            for (int p8 = 0; p8 < 2; ++p8) {
                // eight residual loads in flight before the first store (a load / add / store loop serialises on aliasing)
                float4 x4[8];
                if (EPI == EPI_BIAS_RESID) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int m = (p8 * 8 + q) * 8 + rr;
                        x4[q] = *(const float4*)(resid + (size_t)(m0 + half * 128 + m) * ldc + n0 + ((j ^ (m & 31)) << 2));
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int m = (p8 * 8 + q) * 8 + rr;
                    float4 v = *(const float4*)(smem + m * 1024 + j * 16);
                    const size_t off = SPLITK ? (size_t)blockIdx.x * 65536 + (size_t)(half * 128 + m) * 256 + ((j ^ (m & 31)) << 2)
                                              : (size_t)(m0 + half * 128 + m) * ldc + n0 + ((j ^ (m & 31)) << 2);
                    if (EPI == EPI_BIAS_RESID) {
                        v.x += x4[q].x; v.y += x4[q].y; v.z += x4[q].z; v.w += x4[q].w;
                        *(float4*)(resid + off) = v;
                        if (LN == 2) {
                            // this wave holds the row's 256 columns of the tile: fp16 copy for the next GEMM + the row's partial statistics
                            const f16x4 h4 = {(f16)v.x, (f16)v.y, (f16)v.z, (f16)v.w};
                            *(f16x4*)(ln_x16 + off) = h4;
                            const float mean = row256_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 256.0f);
                            const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
                            const float m2 = row256_sum((a * a + b * b) + (c * c + d * d));
                            if (j == 0) ln_stats[(size_t)(m0 + half * 128 + m) * (N >> 8) + tn] = LnPartial{mean, m2};
                        }
                    } else {
                        *(float4*)((float*)Cout + off) = v;
                    }
                }
            }
        }
    }
    if (PERSIST) __syncthreads();          // the epilogue's LDS image is read before the next tile's first pieces land
    }
    if (TRACE && (tid & 63) == 0 && trace) {
        long long* o = trace + ((size_t)blockIdx.x * 8 + wave) * 8;
        const long long t2 = clock64();
        o[0] = tr_main; o[1] = tr_wait; o[2] = tr_bar; o[3] = (t2 - tr_entry) - tr_main; o[4] = tr_load; o[5] = tr_mma; o[6] = wave;
        o[7] = wall_clock64() - tr_w0;
        if (wave == 0) {                       // second record (slots of wave 1..): where the tile's non-MFMA time goes and on which CU
            long long* e = trace + (size_t)gridDim.x * 64 + (size_t)blockIdx.x * 8;
            e[0] = tr_w_entry; e[1] = wall_clock64();
            e[2] = ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
            e[3] = tr_t0 - tr_entry; e[4] = t2 - tr_t0 - tr_main; e[5] = 1;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// k_gemm_f16_w4 (round 6): the same 256 x 256 x 64 macro tile with FOUR waves, one per SIMD, each 128 tokens x 128 features (8 x 8
// MFMA tiles, 256 accumulators in a0..a255), and a K loop that is ONE hand-scheduled assembly block (csrc/gen_gemm_w4.py writes
// gemm_w4_loop.inc: five-slot LDS ring, one barrier per K-tile, every LDS read / DMA piece / wait at a fixed distance between the
// MFMAs).  A third less LDS read traffic per FLOP than 8 waves x 128 x 64, and no compiler between the MFMAs.
//   A operand = token rows, B operand = weight rows; the LDS image of W is ROW-PERMUTED (through the source addresses of its DMA
//   pieces) so that the eight accumulator tiles ni = 0..7 of a lane are consecutive features: lane (r = lane & 15, q = lane >> 4) of
//   tile (ni, mi), register e holds token mi 16 + 4 q + e and
//       fp16 outputs:  feature 8 r + ni                      -> one 16-byte store per (mi, e), sixteen lanes = 256 contiguous bytes
//       fp32 outputs:  feature 64 (ni >> 2) + 4 r + (ni & 3) -> two 16-byte accesses per (mi, e), sixteen lanes = 256 contiguous bytes
//   of the wave's 128 x 128 quadrant (wm, wn).  The epilogue goes straight from the accumulators to global memory: no LDS image, no
//   barrier.  Same MFMA products in the same K order as k_gemm_f16_pp64 => the same accumulator bits (tests/test_gemm.py).
//   Folded LayerNorm: the producer (LN = 2) writes one (mean, M2) partial per row and 128-COLUMN half (this wave's), the consumer
//   (LN = 1) merges K / 128 of them; k_gemm_f16_pp64 keeps one per 256 columns -- a tower uses one kernel family throughout.
#include "gemm_w4_loop.inc"
// register E of the eight accumulator tiles (ni = 0..7, mi = MI), out of the AGPRs the K loop left them in: a[4 (8 ni + MI) + E].  ONE asm
// statement per eight values, and a whole mi block's four statements in a row: inline asm is a scheduling boundary for hipcc, and with one
// statement per value every element's exp -> add -> rcp chain (QuickGELU) or DPP chain (row statistics) sat alone in its region, padded
// with s_nop (469 per tile in the c_fc epilogue).
template <int MI, int E>
__device__ __forceinline__ void w4_acc8(float (&r)[8]) {
    asm volatile("v_accvgpr_read_b32 %0, a[%8]\n\tv_accvgpr_read_b32 %1, a[%9]\n\tv_accvgpr_read_b32 %2, a[%10]\n\tv_accvgpr_read_b32 %3, a[%11]\n\t"
                 "v_accvgpr_read_b32 %4, a[%12]\n\tv_accvgpr_read_b32 %5, a[%13]\n\tv_accvgpr_read_b32 %6, a[%14]\n\tv_accvgpr_read_b32 %7, a[%15]"
                 : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7])
                 : "n"(4 * (0 * 8 + MI) + E), "n"(4 * (1 * 8 + MI) + E), "n"(4 * (2 * 8 + MI) + E), "n"(4 * (3 * 8 + MI) + E),
                   "n"(4 * (4 * 8 + MI) + E), "n"(4 * (5 * 8 + MI) + E), "n"(4 * (6 * 8 + MI) + E), "n"(4 * (7 * 8 + MI) + E));
}
template <int MI>
__device__ __forceinline__ void w4_acc_block(float (&r)[4][8]) {       // the 4 x 8 values of an mi block (token 4 q + e, eight features)
    w4_acc8<MI, 0>(r[0]); w4_acc8<MI, 1>(r[1]); w4_acc8<MI, 2>(r[2]); w4_acc8<MI, 3>(r[3]);
}
template <int I> struct w4_ic { static constexpr int value = I; };
template <int... Is, typename F>
__device__ __forceinline__ void w4_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(w4_ic<Is>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void w4_for(F&& f) { w4_for_impl(std::make_integer_sequence<int, N>{}, f); }
__device__ __forceinline__ float w4_row16_sum(float v) {      // sum over the 16 lanes of a DPP row (the lanes that share lane >> 4), in every lane
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

#define VG_W4_ASM_OF_(V) VG_W4_ASM_##V
#define VG_W4_ASM_OF(V) VG_W4_ASM_OF_(V)
template <int EPI, int LN = 0, int VAR = 0>
__global__ __launch_bounds__(256, 1) void k_gemm_f16_w4(const f16* __restrict__ X, const f16* __restrict__ Wt,
                                                        const float* __restrict__ bias, void* __restrict__ Cout,
                                                        float* __restrict__ resid, int M, int N, int K, int ldc, int cw,
                                                        const float* __restrict__ ln_c1 = nullptr, LnPartial* __restrict__ ln_stats = nullptr,
                                                        f16* __restrict__ ln_x16 = nullptr, long long* __restrict__ trace = nullptr) {
    constexpr int BM = 256, BN = 256;
    constexpr bool F16OUT = EPI == EPI_BIAS || EPI == EPI_BIAS_GELU;
    constexpr bool HL = EPI == EPI_BIAS_RESID_HL;            // the residual stream as an fp16 pair (hi = ln_x16, lo = (f16*)resid)
    constexpr bool L16 = F16OUT || HL;                       // the accumulator-to-feature layout of 16-bit outputs (eight consecutive features per lane)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    [[maybe_unused]] long long tr_entry = 0, tr_asm0 = 0, tr_asm1 = 0;
    [[maybe_unused]] unsigned t_pro = 0, t_loop = 0, t_wait = 0, t_bar = 0, t_end = 0, t_cal = 0;
    if (VAR == VG_W4_TRACE_VAR) tr_entry = clock64();
    const int ntm = M / BM;
    const int wm = wave >> 1, wn = wave & 1;
    // PERSISTENT: one workgroup per CU walks its XCD's contiguous run of tiles (slot s of G / 8 takes tiles s, s + G / 8, ...: at any time
    // the CUs of an XCD work on neighbouring tiles, the chunked order of the one-tile-per-workgroup kernels)
    const int nt_all = ntm * (N / BN);
    const int xq = nt_all >> 3, xr_ = nt_all & 7, xcd = blockIdx.x & 7;
    const int xbase = (xcd < xr_) ? xcd * (xq + 1) : xr_ * (xq + 1) + (xcd - xr_) * xq;
    const int xcnt = xq + (xcd < xr_ ? 1 : 0);
    const int tstride = (int)(gridDim.x >> 3);
    const int per_chunk = ntm * cw;
    // tile t of the launch -> (row tile, column tile): column tiles in chunks of cw that share their row tile's activations in L2.  One
    // division per tile by the launch constant cw, as a multiplication (exact for t < 2^20, cw <= 2^10); the chunk advances incrementally.
    const unsigned inv_cw = (unsigned)((0x100000000ull + (unsigned)cw - 1u) / (unsigned)cw);
    auto tile_of = [&](int chunk, int tc, int& m0_, int& n0_, int& tn_) {
        const int tm = cw == 1 ? tc : (int)(((unsigned long long)(unsigned)tc * inv_cw) >> 32);      // (cw = 1: the reciprocal 2^32 does not fit)
        tn_ = chunk * cw + (tc - tm * cw);
        m0_ = tm * BM; n0_ = tn_ * BN;
    };
    const int np = K / 64;                          // host guarantees K % 64 == 0 and np >= 4
    constexpr int NSTMAX = 8;                       // width <= 1024
    const int nst = K >> 7;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem);
    const unsigned rowbs = __builtin_amdgcn_readfirstlane((unsigned)K * 2u), wdst = (unsigned)wave * 8192u, npu = (unsigned)__builtin_amdgcn_readfirstlane(np);
    unsigned ring = 0, first = 1;
    int ti = (int)(blockIdx.x >> 3);
    if (ti >= xcnt) return;
    int m0, n0, tn;
    int t_chunk = (xbase + ti) / per_chunk, t_tc = (xbase + ti) - t_chunk * per_chunk;      // (the one real division: before the tile loop)
    tile_of(t_chunk, t_tc, m0, n0, tn);
    // Per-tile operands of the epilogue (bias, and for the folded LayerNorm's consumer c1 and the K / 128 partial statistics of the
    // rows wm 128 + lane, wm 128 + 64 + lane).  Vector memory operations retire IN ORDER and the compiler cannot see the K loop's DMA
    // pieces: a load it waits for behind the assembly block would wait for every prefetched piece, and one issued behind the epilogue's
    // stores for the stores to drain.  So tile t + 1's operands are REQUESTED in front of tile t's K loop (older than everything tile t
    // issues) and TAKEN behind tile t's stores, where the compiler's own count of younger operations already allows them to be in flight.
    float4 raw_b[2] = {}, raw_c[2] = {};
    LnPartial raw_pt[2][NSTMAX] = {};
    float4 cur_b[2] = {}, cur_c[2] = {};
    float ln_mean2[2] = {0.f, 0.f}, ln_rstd2[2] = {0.f, 0.f};
    auto issue_raw = [&](int m0_, int n0_, int lane_) {
        const int r15k = lane_ & 15;
        if constexpr (L16) {
            const int ncol = n0_ + wn * 128 + r15k * 8;
            raw_b[0] = *(const float4*)(bias + ncol); raw_b[1] = *(const float4*)(bias + ncol + 4);
            if (LN == 1) { raw_c[0] = *(const float4*)(ln_c1 + ncol); raw_c[1] = *(const float4*)(ln_c1 + ncol + 4); }
        } else if (EPI == EPI_BIAS_RESID) {
            const int ncol = n0_ + wn * 128 + r15k * 4;
            raw_b[0] = *(const float4*)(bias + ncol); raw_b[1] = *(const float4*)(bias + ncol + 64);
        }
        if (LN == 1) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const LnPartial* sp = ln_stats + (size_t)(m0_ + wm * 128 + h * 64 + lane_) * nst;
#pragma unroll
                for (int i = 0; i < NSTMAX; ++i) raw_pt[h][i] = sp[i < nst ? i : nst - 1];     // unconditional loads (a branch around a load costs a vmcnt(0)); the extras are zeroed when taken
            }
        }
    };
    auto take_raw = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {             // (pinned here: the wait for the loads sits at this point of the program)
            asm volatile("" : "+v"(raw_b[i].x), "+v"(raw_b[i].y), "+v"(raw_b[i].z), "+v"(raw_b[i].w));
            if (LN == 1) asm volatile("" : "+v"(raw_c[i].x), "+v"(raw_c[i].y), "+v"(raw_c[i].z), "+v"(raw_c[i].w));
            cur_b[i] = raw_b[i]; cur_c[i] = raw_c[i];
        }
        if (LN == 1) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float ms = 0.f, m2 = 0.f;
#pragma unroll
                for (int i = 0; i < NSTMAX; ++i) {
                    asm volatile("" : "+v"(raw_pt[h][i].mean), "+v"(raw_pt[h][i].m2));
                    if (i >= nst) raw_pt[h][i] = LnPartial{0.f, 0.f};
                    ms += raw_pt[h][i].mean; m2 += raw_pt[h][i].m2;
                }
                const float mean = ms / (float)nst;
                float dev = 0.f;
#pragma unroll
                for (int i = 0; i < NSTMAX; ++i) if (i < nst) { const float d = raw_pt[h][i].mean - mean; dev += d * d; }
                ln_mean2[h] = mean;
                ln_rstd2[h] = rsqrtf((m2 + 128.f * dev) / (float)K + 1e-5f);
            }
        }
    };
    issue_raw(m0, n0, lane);
    take_raw();
    for (; ti < xcnt; ti += tstride) {
    int m0n = m0, n0n = n0, tnn = tn;               // the next tile, whose first pieces this tile's last K iterations request (none left: this tile again -- the
    if (ti + tstride < xcnt) {                                                   // pieces land in slots nobody reads and are drained before the workgroup ends)
        t_tc += tstride;
        while (t_tc >= per_chunk) { t_tc -= per_chunk; ++t_chunk; }
        tile_of(t_chunk, t_tc, m0n, n0n, tnn);
    }
    int lane_t;                                    // the lane id, produced INSIDE the loop (v_mbcnt on an opaque zero): a loop-invariant lane id is hoisted, kept live
    { unsigned z_; asm volatile("v_mov_b32 %0, 0" : "=v"(z_)); lane_t = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z_)); }      // across the whole loop and spilled
    issue_raw(m0n, n0n, lane_t);                    // the NEXT tile's bias / LayerNorm operands: older than this tile's stores (see issue_raw)
    {
    // per-lane operands of the K loop, re-derived per tile from the lane id (a dozen integer instructions): kept live across the tile
    // loop they are what the register allocator spills first -- and it reloads spills behind a vmcnt(0), i.e. behind the previous tile's stores
    // DMA piece p of wave w fills LDS rows 64 w + 8 p + (lane >> 3) of a slot, 16-byte chunk slot lane & 7; the source chunk is
    // (lane & 7) ^ ((LDS row >> 1) & 7) = (lane & 7) ^ (lane >> 4) ^ 4 (p & 1).  X: LDS row = tile row.  W: LDS row h 128 + ni 16 + r
    // (h = w >> 1, ni = 4 (w & 1) + (p >> 1), r = 8 (p & 1) + (lane >> 3)) holds the feature the output layout asks for (above).
    const unsigned rowb = (unsigned)K * 2u;
    const int lane = lane_t;
    const unsigned c0 = (unsigned)((lane & 7) ^ (lane >> 4));
    const unsigned dv0 = (unsigned)(lane >> 3) * rowb + (c0 << 4), dv1 = (unsigned)(lane >> 3) * rowb + ((c0 ^ 4u) << 4);
    const unsigned wlm = L16 ? 8u : 4u;                                      // feature step per (lane >> 3)
    const unsigned dw0 = (unsigned)(lane >> 3) * wlm * rowb + (c0 << 4), dw1 = (unsigned)(lane >> 3) * wlm * rowb + ((c0 ^ 4u) << 4);
    const unsigned wpo = __builtin_amdgcn_readfirstlane((L16 ? 64u : 32u) * rowb);        // odd pieces: r += 8
    const int wrow0 = (wave >> 1) * 128 + (L16 ? (wave & 1) * 4 : (wave & 1) * 64);         // feature of (ni = 4 (w & 1), r = 0)
    const int r15t = lane & 15, q4t = lane >> 4;
    const unsigned swz = (unsigned)((r15t >> 1) & 7);
    const unsigned xo0 = (unsigned)(wm * 128 + r15t) * 128u + (((unsigned)q4t ^ swz) << 4), xo1 = (unsigned)(wm * 128 + r15t) * 128u + (((unsigned)(4 + q4t) ^ swz) << 4);
    const unsigned wo0 = (unsigned)(wn * 128 + r15t) * 128u + (((unsigned)q4t ^ swz) << 4), wo1 = (unsigned)(wn * 128 + r15t) * 128u + (((unsigned)(4 + q4t) ^ swz) << 4);
        const unsigned long long xp = (unsigned long long)(X + (size_t)(m0 + wave * 64) * K), xpn = (unsigned long long)(X + (size_t)(m0n + wave * 64) * K);
        const unsigned long long wp = (unsigned long long)(Wt + (size_t)(n0 + wrow0) * K), wpn = (unsigned long long)(Wt + (size_t)(n0n + wrow0) * K);
        const unsigned xlo = __builtin_amdgcn_readfirstlane((unsigned)xp), xhi = __builtin_amdgcn_readfirstlane((unsigned)(xp >> 32));
        const unsigned wlo = __builtin_amdgcn_readfirstlane((unsigned)wp), whi = __builtin_amdgcn_readfirstlane((unsigned)(wp >> 32));
        const unsigned nxlo = __builtin_amdgcn_readfirstlane((unsigned)xpn), nxhi = __builtin_amdgcn_readfirstlane((unsigned)(xpn >> 32));
        const unsigned nwlo = __builtin_amdgcn_readfirstlane((unsigned)wpn), nwhi = __builtin_amdgcn_readfirstlane((unsigned)(wpn >> 32));
        const unsigned firsts = __builtin_amdgcn_readfirstlane(first);
        ring = __builtin_amdgcn_readfirstlane(ring);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#define VG_W4_IN [dv0] "v"(dv0), [dv1] "v"(dv1), [dw0] "v"(dw0), [dw1] "v"(dw1), [xo0] "v"(xo0), [xo1] "v"(xo1), [wo0] "v"(wo0),            \
                 [wo1] "v"(wo1), [xlo] "s"(xlo), [xhi] "s"(xhi), [wlo] "s"(wlo), [whi] "s"(whi), [nxlo] "s"(nxlo), [nxhi] "s"(nxhi),        \
                 [nwlo] "s"(nwlo), [nwhi] "s"(nwhi), [rowb] "s"(rowbs), [wpo] "s"(wpo), [lds0] "s"(lds0), [wdst] "s"(wdst), [np] "s"(npu),  \
                 [first] "s"(firsts)
#define VG_W4_RUN(V) asm volatile(VG_W4_ASM_OF(V) : [ring] "+s"(ring) : VG_W4_IN : VG_W4_CLOBBERS)
#define VG_W4_RUN_TRACE(V)                                                                                                            \
        tr_asm0 = clock64();                                                                                                          \
        asm volatile(VG_W4_ASM_OF(V)                                                                                                  \
                     : [ring] "+s"(ring), [t_pro] "=&s"(t_pro), [t_loop] "=&s"(t_loop), [t_wait] "=&s"(t_wait), [t_bar] "=&s"(t_bar),  \
                       [t_end] "=&s"(t_end), [t_cal] "=&s"(t_cal)                                                                      \
                     : VG_W4_IN : VG_W4_CLOBBERS);                                                                                    \
        tr_asm1 = clock64()
        if constexpr (VAR == 0) { if constexpr (F16OUT) { VG_W4_RUN(0H); } else { VG_W4_RUN(0F); } }
#ifdef VG_DEV      // ablations and alternative schedules (gen_gemm_w4.py VARIANTS): development build, VG_GEMM_W4 = 1 + VAR; fp16 outputs only
        VG_W4_DEV_RUNS
#endif
#undef VG_W4_RUN
#undef VG_W4_RUN_TRACE
#undef VG_W4_IN
#pragma clang diagnostic pop
    }
    // ---- epilogue: accumulators -> global memory ----
    // Addresses are a wave-uniform row pointer (SGPR pair) plus ONE per-lane byte offset: 32 row pointers per lane would not leave room
    // for the residual rows in flight.  The lane id is re-derived per tile (mbcnt) behind an opaque asm: nothing of the epilogue's
    // address arithmetic is hoisted out of the tile loop, across the K loop's assembly block.
    int lane_e;
    { unsigned z_; asm volatile("v_mov_b32 %0, 0" : "=v"(z_)); lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z_)); }
    const int r15 = lane_e & 15, q4 = lane_e >> 4;
    const int urow = m0 + wm * 128;                         // wave-uniform first row of the quadrant
    if constexpr (F16OUT) {
        const float bb[8] = {cur_b[0].x, cur_b[0].y, cur_b[0].z, cur_b[0].w, cur_b[1].x, cur_b[1].y, cur_b[1].z, cur_b[1].w};
        const float cc[8] = {cur_c[0].x, cur_c[0].y, cur_c[0].z, cur_c[0].w, cur_c[1].x, cur_c[1].y, cur_c[1].z, cur_c[1].w};
        const unsigned loff = ((unsigned)(4 * q4) * (unsigned)ldc + (unsigned)(r15 * 8)) * 2u;
        const char* cbase = (const char*)((f16*)Cout + (size_t)urow * ldc + n0 + wn * 128);
        // folded LayerNorm: the rows' (mean, rstd) are fetched for FOUR mi blocks at a time, 32 ds_bpermute back to back and pinned behind
        // them (fetched row by row inside the loop every pair sat behind its own lgkmcnt wait -- 32 exposed LDS round trips per tile -- and
        // left alone the compiler sinks every pair back in front of its use).  Row mi 16 + 4 q + e of the wave's half sits in lane
        // (row & 63), register set mi >> 2.
        float rmean[4][4], rrstd[4][4];
        auto fetch_rows = [&](int mg) {
#pragma unroll
            for (int mm = 0; mm < 4; ++mm)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int src = (((mg * 4 + mm) * 16 + 4 * q4 + e) & 63) << 2;
                    rmean[mm][e] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, ln_mean2[mg])));
                    rrstd[mm][e] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, ln_rstd2[mg])));
                }
#pragma unroll
            for (int mm = 0; mm < 4; ++mm)
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(rmean[mm][e]), "+v"(rrstd[mm][e]));
        };
        w4_for<8>([&](auto mic) {
            constexpr int mi = decltype(mic)::value;
            if (LN == 1 && (mi & 3) == 0) fetch_rows(mi >> 2);
            float acc[4][8];
            w4_acc_block<mi>(acc);
            w4_for<4>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                float mean = 0.f, rstd = 1.f;
                if (LN == 1) { mean = rmean[mi & 3][e]; rstd = rrstd[mi & 3][e]; }
                f16x8 h8;
#pragma unroll
                for (int np2 = 0; np2 < 4; ++np2) {      // feature pairs: the non-transcendental steps as packed fp32 operations (same IEEE operations per element)
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const f32x2 a = {acc[e][2 * np2], acc[e][2 * np2 + 1]}, b2 = {bb[2 * np2], bb[2 * np2 + 1]};
                    f32x2 x = a + b2;
                    if (LN == 1) {
                        const f32x2 c2 = {cc[2 * np2], cc[2 * np2 + 1]};
                        x = rstd * (a - mean * c2) + b2;
                    }
                    if (EPI == EPI_BIAS_GELU) {
                        const f32x2 t = x * QGELU_C;
                        const f32x2 d = f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.0f;
                        x = x * f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
                    }
                    h8[2 * np2] = (f16)x.x; h8[2 * np2 + 1] = (f16)x.y;
                }
                *(f16x8*)(const_cast<char*>(cbase) + (size_t)(mi * 16 + e) * ldc * 2 + loff) = h8;
            });
        });
    } else if constexpr (HL) {
        // The residual stream as an fp16 PAIR, in the 16-bit layout (a lane holds eight consecutive features of a token): x = hi + lo read as
        // two 16-byte rows, v = x + acc + bias, hi' = f16(v), lo' = f16(v - hi') written as two 16-byte rows; hi' IS the next GEMM's operand.
        // The row statistics are those of hi' + lo' (what every later reader reconstructs).
        const float bb[8] = {cur_b[0].x, cur_b[0].y, cur_b[0].z, cur_b[0].w, cur_b[1].x, cur_b[1].y, cur_b[1].z, cur_b[1].w};
        const unsigned loff = ((unsigned)(4 * q4) * (unsigned)ldc + (unsigned)(r15 * 8)) * 2u;
        char* hbase = (char*)(ln_x16 + (size_t)urow * ldc + n0 + wn * 128);
        char* lbase = (char*)((f16*)resid + (size_t)urow * ldc + n0 + wn * 128);
        constexpr int WIN = 3;                               // mi blocks of the stream in flight (a rolling window: 96 registers; four spilled)
        f16x8 xh[WIN][4], xl[WIN][4];
        auto load_block = [&](int mb) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const size_t ro = (size_t)(mb * 16 + e) * ldc * 2 + loff;
                xh[mb % WIN][e] = *(const f16x8*)(hbase + ro);
                xl[mb % WIN][e] = *(const f16x8*)(lbase + ro);
            }
        };
        load_block(0); load_block(1); load_block(2);
        w4_for<8>([&](auto mic) {
            constexpr int mi = decltype(mic)::value;
            float st_mean[4], st_m2[4];
            float acc[4][8];
            w4_acc_block<mi>(acc);
            w4_for<4>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                const size_t ro = (size_t)(mi * 16 + e) * ldc * 2 + loff;
                const f16x8 ph = xh[mi % WIN][e], pl = xl[mi % WIN][e];
                f16x8 nh, nl;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float t = (acc[e][j] + bb[j]) + ((float)ph[j] + (float)pl[j]);
                    nh[j] = (f16)t;
                    nl[j] = (f16)(t - (float)nh[j]);
                    v[j] = (float)nh[j] + (float)nl[j];
                }
                *(f16x8*)(hbase + ro) = nh;
                *(f16x8*)(lbase + ro) = nl;
                const float mean = w4_row16_sum(((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) * (1.0f / 128.0f);
                const float a0 = v[0] - mean, a1 = v[1] - mean, a2 = v[2] - mean, a3 = v[3] - mean;
                const float a4 = v[4] - mean, a5 = v[5] - mean, a6 = v[6] - mean, a7 = v[7] - mean;
                st_mean[e] = mean;
                st_m2[e] = w4_row16_sum(((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3)) + ((a4 * a4 + a5 * a5) + (a6 * a6 + a7 * a7)));
            });
            if (mi + WIN < 8) load_block(mi + WIN);          // (this block's registers are free: the next one of the window is requested)
            const float mean = r15 == 0 ? st_mean[0] : r15 == 1 ? st_mean[1] : r15 == 2 ? st_mean[2] : st_mean[3];
            const float m2 = r15 == 0 ? st_m2[0] : r15 == 1 ? st_m2[1] : r15 == 2 ? st_m2[2] : st_m2[3];
            if (r15 < 4)
                *(LnPartial*)((char*)(ln_stats + (size_t)(urow + mi * 16) * (size_t)(N >> 7) + 2 * tn + wn) + (unsigned)((4 * q4 + r15) * (N >> 7)) * 8u) = LnPartial{mean, m2};
        });
    } else {
        // fp32: lane holds features 64 g + 4 r + j (ni = 4 g + j) of the wave's 128: two float4 per (mi, e)
        const float4 b4[2] = {cur_b[0], cur_b[1]};          // (zero unless EPI_BIAS_RESID)
        const unsigned loff = ((unsigned)(4 * q4) * (unsigned)ldc + (unsigned)(r15 * 4)) * 4u, loffh = loff >> 1;
        char* fbase = (char*)((EPI == EPI_BIAS_RESID ? resid : (float*)Cout) + (size_t)urow * ldc + n0 + wn * 128);
        // residual rows: the 128 fragment registers of the K loop are free now -- the residual of FOUR mi blocks (32 float4) is requested
        // at once, twice; a row's store follows its own loads only
        float4 xr[4][4][2];
        auto load_group = [&](int mg) {
#pragma unroll
            for (int mm = 0; mm < 4; ++mm)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const char* rp = fbase + (size_t)((mg * 4 + mm) * 16 + e) * ldc * 4 + loff;
                    xr[mm][e][0] = *(const float4*)rp;
                    xr[mm][e][1] = *(const float4*)(rp + 256);
                }
        };
        w4_for<8>([&](auto mic) {
            constexpr int mi = decltype(mic)::value;
            if (EPI == EPI_BIAS_RESID && (mi & 3) == 0) load_group(mi >> 2);
            [[maybe_unused]] float st_mean[4], st_m2[4];
            float acc[4][8];
            w4_acc_block<mi>(acc);
            w4_for<4>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                float4 v[2];
                char* rp = fbase + (size_t)(mi * 16 + e) * ldc * 4 + loff;
                w4_for<2>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    v[g] = make_float4(acc[e][4 * g + 0] + b4[g].x, acc[e][4 * g + 1] + b4[g].y, acc[e][4 * g + 2] + b4[g].z, acc[e][4 * g + 3] + b4[g].w);
                    if (EPI == EPI_BIAS_RESID) { const float4 x = xr[mi & 3][e][g]; v[g].x += x.x; v[g].y += x.y; v[g].z += x.z; v[g].w += x.w; }
                    *(float4*)(rp + 256 * g) = v[g];
                });
                if (EPI == EPI_BIAS_RESID && LN == 2) {
                    // fp16 copy for the next GEMM + this row's statistics over the wave's 128 columns (every lane of the row's 16 holds them)
                    char* hp = (char*)(ln_x16 + (size_t)(urow + mi * 16 + e) * ldc + n0 + wn * 128) + loffh;
                    const f16x4 h0 = {(f16)v[0].x, (f16)v[0].y, (f16)v[0].z, (f16)v[0].w}, h1 = {(f16)v[1].x, (f16)v[1].y, (f16)v[1].z, (f16)v[1].w};
                    *(f16x4*)hp = h0; *(f16x4*)(hp + 128) = h1;
                    const float mean = w4_row16_sum(((v[0].x + v[0].y) + (v[0].z + v[0].w)) + ((v[1].x + v[1].y) + (v[1].z + v[1].w))) * (1.0f / 128.0f);
                    const float a0 = v[0].x - mean, a1 = v[0].y - mean, a2 = v[0].z - mean, a3 = v[0].w - mean;
                    const float a4 = v[1].x - mean, a5 = v[1].y - mean, a6 = v[1].z - mean, a7 = v[1].w - mean;
                    st_mean[e] = mean;
                    st_m2[e] = w4_row16_sum(((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3)) + ((a4 * a4 + a5 * a5) + (a6 * a6 + a7 * a7)));
                }
            });
            if (EPI == EPI_BIAS_RESID && LN == 2) {
                // ONE predicated store per mi block, behind the four rows' arithmetic (a branch per row cut the block into four scheduling
                // regions and left every row's DPP chain alone with its s_nops): lane r = e of each 16 writes row e's pair
                const float mean = r15 == 0 ? st_mean[0] : r15 == 1 ? st_mean[1] : r15 == 2 ? st_mean[2] : st_mean[3];
                const float m2 = r15 == 0 ? st_m2[0] : r15 == 1 ? st_m2[1] : r15 == 2 ? st_m2[2] : st_m2[3];
                if (r15 < 4)
                    *(LnPartial*)((char*)(ln_stats + (size_t)(urow + mi * 16) * (size_t)(N >> 7) + 2 * tn + wn) + (unsigned)((4 * q4 + r15) * (N >> 7)) * 8u) = LnPartial{mean, m2};
            }
        });
    }
#ifdef VG_DEV
    if (VAR == VG_W4_TRACE_VAR && lane == 0 && trace) {      // per tile and wave: cycles of entry -> asm, block entry, K loop, its waits, the epilogue
        long long* o = trace + ((size_t)(xbase + ti) * 4 + wave) * 12;
        const long long t2 = clock64();
        o[0] = tr_asm0 - tr_entry; o[1] = t_pro; o[2] = t_loop; o[3] = t_wait; o[4] = t_bar; o[5] = t_end; o[6] = t_cal;
        o[7] = t2 - tr_asm1; o[8] = tr_asm1 - tr_asm0; o[9] = first; o[10] = wave; o[11] = wall_clock64();
        tr_entry = clock64();
    }
#endif
    take_raw();
    m0 = m0n; n0 = n0n; tn = tnn; first = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the last block's pieces for a "next tile" have landed before the LDS is given back
}

// Column tiles (256 wide) per L2 chunk of the tile order for the 256 x 256 kernels.  Measured at M = 64512 (sweep with
// VG_GEMM_CW): sharing one activation row-tile between ALL column tiles that run together wins as long as there are at most
// 9 of them (in_proj 9: 278 vs 292 us; c_proj 3, K = 3072: 302 vs 345 us although its 4.7 MB of weights exceed one XCD's L2 --
// with single-column chunks the 396 MB of hidden activations stream from HBM three times); c_fc (12) is best with 4.
static int gemm_chunk_tiles_256(int ntn) {
    if (ntn <= 9) return ntn;
    int cw = 4;
    while (ntn % cw) --cw;
    return cw;
}

#ifdef VG_DEV      // measured slower than k_gemm_f16_pp64 on every projection shape (LAB_NOTES.md section 3, round 3): development build only (VG_GEMM_X2=1)
#include "dev/vit_gemm_x2.inc"
#endif  // VG_DEV

// ---------------------------------------------------------------------------------------------
// Split-K tail of the residual GEMMs (out_proj, c_proj: N = width = three 256-wide column tiles for ViT-B).  One workgroup owns a CU, so a
// launch runs in ROUNDS of n_cu tiles: 333 crops are 257 row tiles x 3 = 771 tiles = three full rounds and a fourth with 3 tiles on 256
// CUs -- the launch takes 4 / 3 of the time for 0.4 % more work (tools/exp_tile_tail.py: +7.6 % tower time from 332 to 333 crops, one
// encode at a time).  The row tiles that do not fill complete rounds are therefore computed K-split: P workgroups per tile, each a
// slice of the K-tiles (k_gemm_f16_pp64<EPI_NONE_F32, ..., SPLITK>: raw fp32 partial tiles into scratch), and k_splitk_resid adds the P
// partials IN FIXED ORDER and applies the residual epilogue -- bias, fp32 residual read-modify-write, and for LN = 2 the fp16 copy and
// the row's partial statistics, the same expressions as the GEMM's own epilogue.  Deterministic; a row in the tail differs from the
// unsplit result by the rounding of ((p0 + p1) + ...) against one fp32 accumulation chain (<= a few 1e-7 relative, tests/test_gemm.py).
template <int LN>
__global__ __launch_bounds__(256) void k_splitk_resid(const float* __restrict__ part, int P, const float* __restrict__ bias,
                                                      float* __restrict__ resid, int N, int ldc, int row0, int ntn,
                                                      LnPartial* __restrict__ ln_stats, f16* __restrict__ ln_x16) {
    // grid: tail tiles x 16; a wave = one row of the tile's 256 columns at a time (lane j: columns 4 j .. 4 j + 3), four rows per wave
    const int tile = blockIdx.x >> 4, rb = blockIdx.x & 15, w = threadIdx.x >> 6, j = threadIdx.x & 63;
    const int tm = tile / ntn, tn = tile - tm * ntn;
    const float4 b4 = *(const float4*)(bias + tn * 256 + 4 * j);
    const float* pb = part + (size_t)tile * P * 65536 + 4 * j;
    float4 x4[4], v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = rb * 16 + w * 4 + i;
        x4[i] = *(const float4*)(resid + (size_t)(row0 + tm * 256 + r) * ldc + tn * 256 + 4 * j);
        v[i] = *(const float4*)(pb + (size_t)r * 256);
    }
    for (int p = 1; p < P; ++p) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = rb * 16 + w * 4 + i;
            const float4 a = *(const float4*)(pb + (size_t)p * 65536 + (size_t)r * 256);
            v[i].x += a.x; v[i].y += a.y; v[i].z += a.z; v[i].w += a.w;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = rb * 16 + w * 4 + i;
        const size_t off = (size_t)(row0 + tm * 256 + r) * ldc + tn * 256 + 4 * j;
        float4 o = make_float4(v[i].x + b4.x, v[i].y + b4.y, v[i].z + b4.z, v[i].w + b4.w);
        o.x += x4[i].x; o.y += x4[i].y; o.z += x4[i].z; o.w += x4[i].w;
        *(float4*)(resid + off) = o;
        if (LN == 2) {
            const f16x4 h4 = {(f16)o.x, (f16)o.y, (f16)o.z, (f16)o.w};
            *(f16x4*)(ln_x16 + off) = h4;
            const float mean = row256_sum((o.x + o.y) + (o.z + o.w)) * (1.0f / 256.0f);
            const float a = o.x - mean, b = o.y - mean, c = o.z - mean, d = o.w - mean;
            const float m2 = row256_sum((a * a + b * b) + (c * c + d * d));
            if (j == 0) ln_stats[(size_t)(row0 + tm * 256 + r) * (N >> 8) + tn] = LnPartial{mean, m2};
        }
    }
}

// how a residual GEMM of ntm x ntn tiles is divided: row tiles [0, r_main) in complete rounds, the rest K-split `parts` ways (0 = no split)
struct SplitPlan { int r_main, parts; };
static SplitPlan splitk_plan(int ntm, int ntn, int np, size_t scratch_bytes, int max_parts, int n_cu) {
    // OPT-IN (VG_GEMM_SPLITK=n, read when the handle is made: at most n parts per tile, 8 is the measured optimum; unset / 0 = never split).
    // Measured (tools/bench_gemm_splitk.py, N = 768, one launch at a time): K = 3072 (c_proj) 356 -> 305 us for 3-15 tail tiles, -12 % at 48,
    // -8 % at 72, -2 % at 126; K = 768 (out_proj: the fourth round costs 16 us, two more launches cost as much) -3 % at 6 tiles, +2 % at 30,
    // +9 % at 48: only long K loops are split.  Whole tower (tools/exp_tile_tail.py): one encode at a time 14.12 -> 13.73 ms at 333 crops,
    // 14.35 -> 13.94 at 338; TWO encodes in flight (what the pipeline runs) 12.94 -> 13.01 / 13.27 -> 13.29: the other encode's tiles
    // already fill the last round, and the partial tiles are 60 MB of extra traffic per launch; pipeline 65.5 vs 65.7 frames/s.  So the
    // split is for one-frame-at-a-time (latency) use and stays off in the throughput configuration.
    if (max_parts < 2 || n_cu < 8) return {ntm, 0};        // (more parts = more partial-sum traffic than K-loop saved: 256 KB per part and tile)
    if (np < 24) return {ntm, 0};
    const int total = ntm * ntn, full = total / n_cu;
    if (full < 1) return {ntm, 0};
    const int r_main = full * n_cu / ntn, tail = (ntm - r_main) * ntn;
    if (tail <= 0 || 2 * tail > n_cu) return {ntm, 0};     // a last round that is more than half full is left alone
    int parts = n_cu / tail;
    if (parts > np / 2) parts = np / 2;                    // the kernel's ring needs two K-tiles per workgroup
    if (parts > max_parts) parts = max_parts;
    while (parts >= 2 && (size_t)tail * parts * 262144 > scratch_bytes) --parts;
    if (parts < 2) return {ntm, 0};
    return {r_main, parts};
}

template <int EPI, bool TRACE = false, bool PERSIST = false, int LN = 0>
static int launch_gemm_pp64(const void* X, const void* Wt, const float* bias, void* C, float* resid, int M, int N, int K, int ldc,
                            hipStream_t st, long long* trace = nullptr, const float* ln_c1 = nullptr, LnPartial* ln_stats = nullptr,
                            f16* ln_x16 = nullptr) {
    if (M % 256 || N % 256 || K % 64 || K / 64 < 2) return VG_ERR_ARG;
    if (LN == 1 && (K % 256 || !ln_c1 || !ln_stats)) return VG_ERR_ARG;
    if (LN == 2 && (ldc != N || !ln_stats || !ln_x16)) return VG_ERR_ARG;
    auto kern = k_gemm_f16_pp64<EPI, TRACE, PERSIST, LN>;
    const int lds = 5 * 32768;
    VG_MAX_DYNAMIC_LDS(kern, lds);
    const int ntn = N / 256;
    int cwt = gemm_chunk_tiles_256(ntn);
#ifdef VG_DEV
    if (getenv("VG_GEMM_CW")) { cwt = atoi(getenv("VG_GEMM_CW")); if (cwt < 1 || ntn % cwt) cwt = ntn; }      // tile-order sweep (development build)
#endif
    int grid = (M / 256) * ntn;
    if (PERSIST) {
        static int n_cu = 0;
        if (!n_cu) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); if (n_cu < 8) n_cu = 256; }
        if (grid > n_cu) grid = n_cu;
        grid = (grid + 7) / 8 * 8;                 // slot s of XCD x = block 8 s + x
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, (const f16*)X, (const f16*)Wt, bias, C, resid, M, N, K,
                       ldc, cwt, trace, ln_c1, ln_stats, ln_x16, 1);
    VG_LAUNCH_CHECK();
    return VG_OK;
}

template <int EPI, int LN = 0, int VAR = 0>
static int launch_gemm_w4(const void* X, const void* Wt, const float* bias, void* C, float* resid, int M, int N, int K, int ldc,
                          hipStream_t st, const float* ln_c1 = nullptr, LnPartial* ln_stats = nullptr, f16* ln_x16 = nullptr,
                          long long* trace = nullptr) {
    if (M % 256 || N % 256 || K % 64 || K / 64 < 4) return VG_ERR_ARG;
    if (LN == 1 && (K % 128 || K > 1024 || !ln_c1 || !ln_stats)) return VG_ERR_ARG;
    if (LN == 2 && (ldc != N || !ln_stats || !ln_x16)) return VG_ERR_ARG;
    auto kern = k_gemm_f16_w4<EPI, LN, VAR>;
    const int lds = 5 * 32768;
    VG_MAX_DYNAMIC_LDS(kern, lds);
    const int ntn = N / 256;
    static std::atomic<int> n_cu_{0};                // persistent grid: one workgroup per CU, a multiple of 8 (slot s of XCD x = block 8 s + x)
    int n_cu = n_cu_.load();
    if (!n_cu) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); if (n_cu < 8) n_cu = 256; n_cu_.store(n_cu); }
    int grid = (M / 256) * ntn;
    if (grid > n_cu) grid = n_cu;
#ifdef VG_DEV
    { static const int cap = getenv("VG_GEMM_W4_GRID") ? atoi(getenv("VG_GEMM_W4_GRID")) : 0; if (cap >= 8 && grid > cap) grid = cap; }   // experiment: two encodes on disjoint CU halves
#endif
    grid = (grid + 7) / 8 * 8;
    int cwt = gemm_chunk_tiles_256(ntn);
#ifdef VG_DEV
    if (getenv("VG_GEMM_CW")) { const int c = atoi(getenv("VG_GEMM_CW")); if (c >= 1 && ntn % c == 0) cwt = c; }      // tile-order sweep (development build; read per launch)
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, (const f16*)X, (const f16*)Wt, bias, C, resid, M, N, K,
                       ldc, cwt, ln_c1, ln_stats, ln_x16, trace);
    VG_LAUNCH_CHECK();
    return VG_OK;
}

// rows [row0, row0 + Mt) of a residual GEMM, K-split `parts` ways through `scratch` (see k_splitk_resid)
template <int LN>
static int launch_gemm_resid_tail(const void* X, const void* Wt, const float* bias, float* resid, int row0, int Mt, int N, int K, int ldc,
                                  int parts, float* scratch, hipStream_t st, LnPartial* ln_stats, f16* ln_x16) {
    if (Mt % 256 || N % 256 || K % 64 || parts < 2 || K / 64 / parts < 2 || !scratch) return VG_ERR_ARG;
    auto kern = k_gemm_f16_pp64<EPI_NONE_F32, false, false, 0, true>;
    const int lds = 5 * 32768;
    VG_MAX_DYNAMIC_LDS(kern, lds);
    const int ntn = N / 256, tiles = (Mt / 256) * ntn;
    hipLaunchKernelGGL(kern, dim3(tiles * parts), dim3(512), lds, st, (const f16*)X + (size_t)row0 * K, (const f16*)Wt, (const float*)nullptr,
                       (void*)scratch, (float*)nullptr, Mt, N, K, 256, ntn, (long long*)nullptr, (const float*)nullptr, (LnPartial*)nullptr,
                       (f16*)nullptr, parts);
    VG_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_splitk_resid<LN>), dim3(tiles * 16), dim3(256), 0, st, (const float*)scratch, parts, bias, resid, N, ldc, row0, ntn,
                       ln_stats, ln_x16);
    VG_LAUNCH_CHECK();
    return VG_OK;
}


#ifdef VG_DEV
#include "dev/vit_dev_launchers.inc"
#endif  // VG_DEV

template <int EPI, int LN = 0>
static int launch_gemm(const vg_vit* cv, const void* X, const void* Wt, const float* bias, void* C, float* resid, int M,
                       int N, int K, hipStream_t st, int ldc = 0, const float* ln_c1 = nullptr, LnPartial* ln_stats = nullptr,
                       f16* ln_x16 = nullptr, float* sk_scratch = nullptr, size_t sk_bytes = 0) {
    if (ldc == 0) ldc = N;
    vg_vit* v = const_cast<vg_vit*>(cv);
    // f16 ViT shapes (N % 256 == 0, K >= 128) take the ping-pong kernel; k_gemm_f16 serves the remaining legal shapes.
    bool use_pp = v->dtype == 1 && N % 256 == 0 && K % 64 == 0 && K / 64 >= 2;
#ifdef VG_DEV
    if (getenv("VG_GEMM_V4")) use_pp = false;          // every shape through k_gemm_f16 (tools/dev/bench_gemm_v4.sh)
#endif
    if (!v->n_cu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&v->n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    }
    const bool prof = v->prof_on && v->prof_n < VG_PROF_MAX;
    if (prof) (void)hipEventRecord(v->prof_ev[2 * v->prof_n], st);
    struct Closer {
        vg_vit* v; bool prof; hipStream_t st; double fl; int kind;
        ~Closer() {
            if (prof) {
                (void)hipEventRecord(v->prof_ev[2 * v->prof_n + 1], st);
                v->prof_kind[v->prof_n] = kind;
                v->prof_flops[v->prof_n++] = fl;
            }
        }
    } closer{v, prof, st, 2.0 * (double)M * (double)N * (double)K, use_pp ? 1 : 0};
    if constexpr (EPI == EPI_BIAS_RESID_HL) {     // the stream as an fp16 pair: k_gemm_f16_w4's producer epilogue only
        if (!(LN == 2 && v->dtype == 1 && v->gemm_w4 && use_pp && M % 256 == 0 && K % 64 == 0 && K / 64 >= 4 && resid && ln_x16)) return VG_ERR_ARG;
        return launch_gemm_w4<EPI, LN>(X, Wt, bias, C, resid, M, N, K, ldc, st, ln_c1, ln_stats, ln_x16);
    } else
    if constexpr (EPI == EPI_BIAS_RESID_H) {      // fp16 residual stream: only the 256 x 256 kernel implements it
        if (!(use_pp && K % 64 == 0 && K / 64 >= 2)) return VG_ERR_ARG;
        return launch_gemm_pp64<EPI>(X, Wt, bias, C, resid, M, N, K, ldc, st);
    } else
    if (v->dtype == 1) {
        if (M % GBM || N % GBN || K % GK) return VG_ERR_ARG;
        if (use_pp) {
            {
                // experiment (development build only): persistent workgroups (one per CU walking its XCD's run of tiles) per epilogue kind,
                // bit EPI of VG_GEMM_PERSIST.  Alone at M = 64256: in_proj (+bias) 271 -> 244 us, c_fc (GELU) and out_proj +-0,
                // c_proj -3 %; inside the pipeline no measurable change (13.65-13.70 ms of GEMMs per frame either way): the next
                // tile's first pieces still wait for the previous tile's stores (one vmcnt), so only the dispatch gap is saved.
#ifdef VG_DEV
                static const int persist_mask = getenv("VG_GEMM_PERSIST") ? atoi(getenv("VG_GEMM_PERSIST")) : 0;
                if constexpr (LN == 0 && (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RESID)) {
                    if ((persist_mask >> EPI) & 1) return launch_gemm_pp64<EPI, false, true>(X, Wt, bias, C, resid, M, N, K, ldc, st);
                }
#endif
#ifdef VG_DEV
                if (v->gemm_x2 && K % 256 == 0 && (LN != 1 || K / 64 <= X2_LN_MAXP))
                    return launch_gemm_x2<EPI, LN>(X, Wt, bias, C, resid, M, N, K, ldc, st, ln_c1, ln_stats, ln_x16);
                if (v->gemm_x2 && LN != 0) return VG_ERR_ARG;       // (the two kernels keep different partial statistics)
#endif
                if constexpr (EPI == EPI_BIAS_RESID && LN != 1) {
                    // residual GEMMs with scratch at hand: the row tiles beyond the last complete round of tiles run K-split
                    if (sk_scratch && M % 256 == 0 && !(v->gemm_w4 && K / 64 >= 4)) {        // (k_gemm_f16_w4 has no tile rounds to fill: persistent workgroups)
                        const SplitPlan sp = splitk_plan(M / 256, N / 256, K / 64, sk_bytes, v->splitk_max, v->n_cu);
                        if (sp.parts) {
                            const int rc = launch_gemm_pp64<EPI, false, false, LN>(X, Wt, bias, C, resid, sp.r_main * 256, N, K, ldc, st, nullptr, ln_c1, ln_stats, ln_x16);
                            if (rc) return rc;
                            return launch_gemm_resid_tail<LN>(X, Wt, bias, resid, sp.r_main * 256, M - sp.r_main * 256, N, K, ldc, sp.parts, sk_scratch, st,
                                                              ln_stats, ln_x16);
                        }
                    }
                }
                if (v->gemm_w4 && K / 64 >= 4) {
#ifdef VG_DEV
                    if constexpr (LN == 0 && EPI == EPI_BIAS) {
                        switch (v->gemm_w4) { VG_W4_DEV_CASES(EPI, LN) }
                    }
#endif
                    return launch_gemm_w4<EPI, LN>(X, Wt, bias, C, resid, M, N, K, ldc, st, ln_c1, ln_stats, ln_x16);
                }
                return launch_gemm_pp64<EPI, false, false, LN>(X, Wt, bias, C, resid, M, N, K, ldc, st, nullptr, ln_c1, ln_stats, ln_x16);
            }
        }
        if (LN != 0) return VG_ERR_ARG;            // the folded LayerNorm exists in the 256 x 256 kernel only
        int nwg = (M / GBM) * (N / GBN);
        VG_MAX_DYNAMIC_LDS(k_gemm_f16<EPI>, G_LDS_BYTES);
        hipLaunchKernelGGL((k_gemm_f16<EPI>), dim3(nwg), dim3(512), G_LDS_BYTES, st, (const f16*)X, (const f16*)Wt, bias, C,
                           resid, M, N, K, ldc, gemm_chunk_tiles(N, K));
    } else {
        if (M % 64 || N % 64 || K % 16) return VG_ERR_ARG;
        if (M % 128 == 0 && N % 128 == 0 && v->f32_mfma) {      // (VG_GEMM_F32_MFMA=0: the vector-ALU kernel for every shape, bit-identical)
            hipLaunchKernelGGL((k_gemm_f32_mfma<EPI>), dim3((M / 128) * (N / 128)), dim3(256), 0, st, (const float*)X, (const float*)Wt, bias,
                               (float*)C, resid, M, N, K);
        } else {
            int nwg = (M / 64) * (N / 64);
            hipLaunchKernelGGL((k_gemm_f32<EPI>), dim3(nwg), dim3(256), 0, st, (const float*)X, (const float*)Wt, bias,
                               (float*)C, resid, M, N, K);
        }
    }
    VG_LAUNCH_CHECK();
    return VG_OK;
}

template <bool TRACE>
static int launch_attention(const f16* qkv, f16* out, int T, int W, int heads, int ld, int items, long long* trace, hipStream_t st,
                            int q_tiles = 7, bool tr = true, bool stagger = true) {
    const int nkb = (T + 31) / 32;
    if (nkb < 1 || nkb > 7) return VG_ERR_ARG;
    // ViT-B/16: row-major V + transposing LDS reads (VG_ATT_TR=0: the transposed V image of rounds 1-2; same numbers, 1.6 % slower)
    // (tr: vg_vit::att_tr of the calling handle; tests run both paths in one process and compare them bit for bit)
    if (T == 197 && !TRACE && tr && stagger) {
        const dim3 grid7(items < 256 ? items : 256);
        VG_MAX_DYNAMIC_LDS((k_attention_f16<false, 7, 197, true, true>), AT_LDS_BYTES_TR);
        hipLaunchKernelGGL((k_attention_f16<false, 7, 197, true, true>), grid7, dim3(448), AT_LDS_BYTES_TR, st, qkv, out, T, W, heads, ld, items, trace, q_tiles);
        VG_LAUNCH_CHECK();
        return VG_OK;
    }
    if (T == 197 && !TRACE && tr) {
        const dim3 grid7(items < 256 ? items : 256);
        VG_MAX_DYNAMIC_LDS((k_attention_f16<false, 7, 197, true>), AT_LDS_BYTES_TR);
        hipLaunchKernelGGL((k_attention_f16<false, 7, 197, true>), grid7, dim3(448), AT_LDS_BYTES_TR, st, qkv, out, T, W, heads, ld, items, trace, q_tiles);
        VG_LAUNCH_CHECK();
        return VG_OK;
    }
    const dim3 grid(items < 256 ? items : 256), block(448);
#define VG_ATT(N)                                                                                                                  \
    case N: {                                                                                                                      \
        VG_MAX_DYNAMIC_LDS((k_attention_f16<TRACE, N>), AT_LDS_BYTES);                                                              \
        hipLaunchKernelGGL((k_attention_f16<TRACE, N>), grid, block, AT_LDS_BYTES, st, qkv, out, T, W, heads, ld, items, trace, q_tiles);   \
        break; }
    if (T == 197 && !TRACE) {                  // ViT-B/16
        VG_MAX_DYNAMIC_LDS((k_attention_f16<false, 7, 197>), AT_LDS_BYTES);
        hipLaunchKernelGGL((k_attention_f16<false, 7, 197>), grid, block, AT_LDS_BYTES, st, qkv, out, T, W, heads, ld, items, trace, q_tiles);
        VG_LAUNCH_CHECK();
        return VG_OK;
    }
    switch (nkb) { VG_ATT(1) VG_ATT(2) VG_ATT(3) VG_ATT(4) VG_ATT(5) VG_ATT(6) VG_ATT(7) }
#undef VG_ATT
    VG_LAUNCH_CHECK();
    return VG_OK;
}

// W'[n,k] = f16(g[k] * W[n,k]);  c1[n] = sum_k W'[n,k] (of the ROUNDED values, which is what the MFMAs will sum);
// c2[n] = b[n] + sum_k beta[k] * W[n,k].  One workgroup per output feature n, fixed summation order.
__global__ __launch_bounds__(256) void k_ln_fold(const float* __restrict__ W32, const float* __restrict__ g, const float* __restrict__ beta,
                                                 const float* __restrict__ b, f16* __restrict__ Wf, float* __restrict__ c1,
                                                 float* __restrict__ c2, int K) {
    __shared__ double sh1[256], sh2[256];
    const int n = blockIdx.x, tid = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int k = tid; k < K; k += 256) {
        const float w = W32[(size_t)n * K + k];
        const f16 wf = (f16)(g[k] * w);
        Wf[(size_t)n * K + k] = wf;
        s1 += (double)(float)wf;
        s2 += (double)beta[k] * (double)w;
    }
    sh1[tid] = s1; sh2[tid] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { sh1[tid] += sh1[tid + o]; sh2[tid] += sh2[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) { c1[n] = (float)sh1[0]; c2[n] = (float)((double)b[n] + sh2[0]); }
}

static void* vit_w(const vg_vit* v, const std::string& n) {
    auto it = v->w.find(n);
    return it == v->w.end() ? nullptr : it->second;
}

/* Builds the derived tensors of the folded LayerNorms once per handle (and again after a weight was replaced).  Called at the top
 * of vg_vit_encode; allocates and synchronises, so the first encode of a handle must not run inside a stream capture
 * (vg_vit_classify_graph runs it as plain launches). */
static int vit_fold_ln(vg_vit* v, hipStream_t st) {
    if (!v->ln_fold || v->fold_ready.load()) return VG_OK;
    std::lock_guard<std::mutex> lock(v->mtx);
    if (v->fold_ready.load()) return VG_OK;
    const int W = v->width;
    for (int l = 0; l < v->layers; ++l) {
        const std::string p = "transformer.resblocks." + std::to_string(l) + ".";
        const char* sets[2][4] = {{"ln_1.weight", "ln_1.bias", "attn.in_proj_weight", "attn.in_proj_bias"},
                                  {"ln_2.weight", "ln_2.bias", "mlp.c_fc.weight", "mlp.c_fc.bias"}};
        for (int k = 0; k < 2; ++k) {
            const std::string wn = p + sets[k][2];
            auto it32 = v->w32.find(wn);
            const float* g = (const float*)vit_w(v, p + sets[k][0]);
            const float* be = (const float*)vit_w(v, p + sets[k][1]);
            const float* b = (const float*)vit_w(v, p + sets[k][3]);
            if (it32 == v->w32.end() || !g || !be || !b) continue;      // vg_vit_encode reports the missing weight by name
            const int N = (int)(v->numel[wn] / (size_t)W);
            for (const char* suffix : {"#ln", "#c1", "#c2"}) {
                auto old = v->w.find(wn + suffix);
                if (old != v->w.end()) { (void)hipFree(old->second); v->w.erase(old); }
            }
            void *wf = nullptr, *c1 = nullptr, *c2 = nullptr;
            VG_CHECK(hipMalloc(&wf, (size_t)N * W * 2));
            VG_CHECK(hipMalloc(&c1, (size_t)N * 4));
            VG_CHECK(hipMalloc(&c2, (size_t)N * 4));
            hipLaunchKernelGGL(k_ln_fold, dim3(N), dim3(256), 0, st, (const float*)it32->second, g, be, b, (f16*)wf, (float*)c1, (float*)c2, W);
            VG_LAUNCH_CHECK();
            v->w[wn + "#ln"] = wf; v->w[wn + "#c1"] = c1; v->w[wn + "#c2"] = c2;
        }
    }
    VG_CHECK(hipStreamSynchronize(st));          // other threads' streams may use the tensors as soon as the flag is up
    v->weights_gen.fetch_add(1);
    v->fold_ready.store(true);
    return VG_OK;
}

// Single-channel patch embedding (mv_utils.py:36: the three channels of a rendered crop are one image).  With the crop's
// uint8 level u, channel c of the reference's input is (u / 255 - mean_c) / std_c (clip.py:79-86), so
//   conv1(x)[n] = sum_p (u_p / 256) * W1[n,p] + b1[n],   W1[n,p] = (256 / 255) * sum_c conv1[n,c,p] / std_c,
//                                                        b1[n]   = - sum_c (mean_c / std_c) * sum_p conv1[n,c,p]
// -- exact in real arithmetic; the renderer hands over u / 256 (exact in fp16), W1 is rounded to fp16 once from the fp32 sum.
// b1 is added to the positional embedding of the 196 patch tokens (k_embed_lnpre adds that table to the GEMM's rows anyway).
// One workgroup per output feature n, fixed summation order, double accumulation.
__global__ __launch_bounds__(256) void k_conv1_fold(const float* __restrict__ conv1, const float* __restrict__ pos, f16* __restrict__ W1,
                                                    float* __restrict__ pos1, int W, int T, int pp, float m0, float m1, float m2,
                                                    float s0, float s1, float s2) {
    __shared__ double sh[256];
    const int n = blockIdx.x, tid = threadIdx.x;
    const float* w = conv1 + (size_t)n * 3 * pp;
    double acc = 0.0;
    for (int p = tid; p < pp; p += 256) {
        const double a = (double)w[p], b = (double)w[pp + p], c = (double)w[2 * pp + p];
        W1[(size_t)n * pp + p] = (f16)(float)((256.0 / 255.0) * (a / (double)s0 + b / (double)s1 + c / (double)s2));
        acc += a * ((double)m0 / (double)s0) + b * ((double)m1 / (double)s1) + c * ((double)m2 / (double)s2);
    }
    sh[tid] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) sh[tid] += sh[tid + o];
        __syncthreads();
    }
    const float b1 = (float)(-sh[0]);
    for (int t = tid; t < T; t += 256) pos1[(size_t)t * W + n] = pos[(size_t)t * W + n] + (t > 0 ? b1 : 0.f);
}

/* Derived tensors of the single-channel patch embedding, once per handle (again after conv1.weight / positional_embedding /
 * the input normalisation changed).  Allocates and synchronises like vit_fold_ln: not inside a stream capture. */
static int vit_fold_conv1(vg_vit* v, hipStream_t st) {
    if (v->conv1_ready.load()) return VG_OK;
    std::lock_guard<std::mutex> lock(v->mtx);
    if (v->conv1_ready.load()) return VG_OK;
    auto it32 = v->w32.find("conv1.weight");
    const float* pos = (const float*)vit_w(v, "positional_embedding");
    if (v->dtype != 1 || it32 == v->w32.end() || !pos) return VG_ERR_ARG;
    const int W = v->width, pp = v->patch * v->patch;
    for (const char* nm : {"conv1.weight#1ch", "positional_embedding#1ch"}) {
        auto old = v->w.find(nm);
        if (old != v->w.end()) { (void)hipFree(old->second); v->w.erase(old); }
    }
    void *w1 = nullptr, *p1 = nullptr;
    VG_CHECK(hipMalloc(&w1, (size_t)W * pp * 2));
    VG_CHECK(hipMalloc(&p1, (size_t)v->T * W * 4));
    hipLaunchKernelGGL(k_conv1_fold, dim3(W), dim3(256), 0, st, (const float*)it32->second, pos, (f16*)w1, (float*)p1, W, v->T, pp,
                       v->in_mean[0], v->in_mean[1], v->in_mean[2], v->in_std[0], v->in_std[1], v->in_std[2]);
    VG_LAUNCH_CHECK();
    v->w["conv1.weight#1ch"] = w1;
    v->w["positional_embedding#1ch"] = p1;
    VG_CHECK(hipStreamSynchronize(st));
    v->weights_gen.fetch_add(1);
    v->conv1_ready.store(true);
    return VG_OK;
}

extern "C" {

/* per-channel input normalisation (x / 255 - mean_c) / std_c that the single-channel patch rows (input_kind 3) fold into the patch
 * embedding; default: CLIP's constants (clip.py:79-86). */
int vg_vit_set_input_norm(vg_vit* v, const float* h_mean3, const float* h_std3) {
    if (!v || !h_mean3 || !h_std3) return VG_ERR_ARG;
    for (int c = 0; c < 3; ++c) {
        if (!(h_std3[c] > 0.f)) return VG_ERR_ARG;
        v->in_mean[c] = h_mean3[c]; v->in_std[c] = h_std3[c];
    }
    v->conv1_ready.store(false);
    return VG_OK;
}

int vg_vit_create(vg_vit** out, int width, int layers, int heads, int patch, int resolution, int out_dim, int dtype) {
    if (!out || width % 128 || width > 1024 || heads * 64 != width || resolution % patch || dtype < 0 || dtype > 1)
        return VG_ERR_ARG;
    int T = (resolution / patch) * (resolution / patch) + 1;
    if (T > AT_MAXT || (3 * patch * patch) % 64) return VG_ERR_ARG;
    vg_vit* v = new vg_vit();
    v->width = width; v->layers = layers; v->heads = heads; v->patch = patch; v->res = resolution;
    v->out_dim = out_dim; v->dtype = dtype; v->T = T;
    v->resid_h = dtype == 1 && width % 256 == 0 && getenv("VG_VIT_RESID16");
    const char* fold = getenv("VG_VIT_LN_FOLD");
    v->ln_fold = dtype == 1 && width % 256 == 0 && !v->resid_h && !(fold && atoi(fold) == 0);
    { const char* hl = getenv("VG_VIT_RESID_HL"); v->resid_hl = v->ln_fold && v->gemm_w4 && v->cls_last && width <= 1024 && !(hl && atoi(hl) == 0); }
#ifdef VG_DEV
    if (getenv("VG_GEMM_V4")) v->resid_h = v->ln_fold = false;      // (k_gemm_f16 has neither epilogue)
#endif
#ifdef VG_DEV
    // k_gemm_f16_x2 serves K % 256 == 0 and merges at most X2_LN_MAXP partial statistics per row: a tower uses it for all of its
    // projection GEMMs or for none (the two kernels keep the folded LayerNorm's partials at different granularity)
    if (v->gemm_x2 && !(dtype == 1 && width % 256 == 0 && width / 64 <= X2_LN_MAXP && !v->resid_h && !getenv("VG_GEMM_V4"))) v->gemm_x2 = false;
#endif
    *out = v;
    return VG_OK;
}

void vg_vit_destroy(vg_vit* v) {
    if (!v) return;
    if (v->prof_init) for (int i = 0; i < 2 * VG_PROF_MAX; ++i) (void)hipEventDestroy(v->prof_ev[i]);
    for (auto& kv : v->w) (void)hipFree(kv.second);
    for (auto& kv : v->w32) (void)hipFree(kv.second);
    delete v;
}

/* upload one tensor by its reference state_dict name (prefix 'visual.' dropped); h_data is host float32.
 * GEMM weights are stored in the compute dtype, everything else stays float32. */
int vg_vit_set_weight(vg_vit* v, const char* name, const float* h_data, int64_t numel) {
    if (!v || !name || !h_data || numel <= 0) return VG_ERR_ARG;
    std::string n(name);
    std::lock_guard<std::mutex> lock(v->mtx);
    auto it = v->w.find(n);
    if (it != v->w.end()) {
        (void)hipFree(it->second);
        v->w.erase(it);
    }
    if (v->dtype == 1 && (n == "conv1.weight" || n == "positional_embedding")) {
        v->conv1_ready.store(false);             // the single-channel patch embedding's derived tensors are rebuilt at the next encode
        if (n == "conv1.weight") {
            auto o = v->w32.find(n);
            if (o != v->w32.end()) { (void)hipFree(o->second); v->w32.erase(o); }
            void* d32 = nullptr;
            VG_CHECK(hipMalloc(&d32, (size_t)numel * 4));
            VG_CHECK(hipMemcpy(d32, h_data, (size_t)numel * 4, hipMemcpyHostToDevice));
            v->w32[n] = d32;
        }
    }
    if (v->ln_fold && n.find("transformer.resblocks.") == 0) {
        v->fold_ready.store(false);              // a block's tensor changed: the folded LayerNorm tensors are rebuilt at the next encode
        if (n.find("in_proj_weight") != std::string::npos || n.find("c_fc.weight") != std::string::npos) {
            auto o = v->w32.find(n);
            if (o != v->w32.end()) { (void)hipFree(o->second); v->w32.erase(o); }
            void* d32 = nullptr;
            VG_CHECK(hipMalloc(&d32, (size_t)numel * 4));
            VG_CHECK(hipMemcpy(d32, h_data, (size_t)numel * 4, hipMemcpyHostToDevice));
            v->w32[n] = d32;
        }
    }
    void* d = nullptr;
    if (v->dtype == 1 && is_gemm_weight(n)) {
        std::vector<f16> tmp((size_t)numel);
        for (int64_t i = 0; i < numel; ++i) tmp[i] = (f16)h_data[i];
        VG_CHECK(hipMalloc(&d, (size_t)numel * 2));
        VG_CHECK(hipMemcpy(d, tmp.data(), (size_t)numel * 2, hipMemcpyHostToDevice));
    } else {
        VG_CHECK(hipMalloc(&d, (size_t)numel * 4));
        VG_CHECK(hipMemcpy(d, h_data, (size_t)numel * 4, hipMemcpyHostToDevice));
    }
    v->w[n] = d;
    v->numel[n] = (size_t)numel;
    v->weights_gen.fetch_add(1);
    return VG_OK;
}

// The last block's class-token rows, compacted: hc[c] = h[c * T] (attention output, fp16), xc[c] = x[c * T] (residual stream, fp32);
// rows n_crops .. Mc - 1 (padding of the GEMM row tile) are zeroed.  One workgroup per row.
__global__ __launch_bounds__(256) void k_gather_cls(const f16* __restrict__ h, const float* __restrict__ x, f16* __restrict__ hc,
                                                    float* __restrict__ xc, int n_crops, int T, int W,
                                                    const f16* __restrict__ pair_hi = nullptr, const f16* __restrict__ pair_lo = nullptr) {
    // pair_hi / pair_lo: the stream kept as an fp16 pair (EPI_BIAS_RESID_HL) -- the compact rows are fp32 either way
    const int c = blockIdx.x;
    for (int i = threadIdx.x; i < W; i += 256) {
        hc[(size_t)c * W + i] = c < n_crops ? h[(size_t)c * T * W + i] : (f16)0.f;
        if (pair_hi) xc[(size_t)c * W + i] = c < n_crops ? (float)pair_hi[(size_t)c * T * W + i] + (float)pair_lo[(size_t)c * T * W + i] : 0.f;
        else xc[(size_t)c * W + i] = c < n_crops ? x[(size_t)c * T * W + i] : 0.f;
    }
}

// The last block's QUERY projection needs the class-token rows only (round 5): their rows of the fp16 residual copy and of the folded
// LayerNorm's partial statistics, compacted (rows n_crops .. Mc - 1 zeroed: statistics (0, 0) give rstd = 1 / sqrt(eps), finite) ...
__global__ __launch_bounds__(256) void k_gather_cls_ln(const f16* __restrict__ x16, const LnPartial* __restrict__ st, f16* __restrict__ x16c,
                                                       LnPartial* __restrict__ stc, int n_crops, int T, int W, int nst) {
    const int c = blockIdx.x;
    for (int i = threadIdx.x; i < W; i += 256) x16c[(size_t)c * W + i] = c < n_crops ? x16[(size_t)c * T * W + i] : (f16)0.f;
    if ((int)threadIdx.x < nst) stc[(size_t)c * nst + threadIdx.x] = c < n_crops ? st[(size_t)c * T * nst + threadIdx.x] : LnPartial{0.f, 0.f};
}
// ... and the projected queries back into the class-token rows of the qkv buffer (columns [0, W)), where the attention reads them
__global__ __launch_bounds__(256) void k_scatter_cls_q(const f16* __restrict__ qc, f16* __restrict__ qkv, int T, int W, int ld) {
    const int c = blockIdx.x;
    for (int i = threadIdx.x; i < W; i += 256) qkv[(size_t)c * T * ld + i] = qc[(size_t)c * W + i];
}

static int64_t pad128(int64_t m) { return (m + 255) / 256 * 256; }   // GEMM row tile (256)

/* bytes of zero-initialised device workspace vg_vit_encode needs for n_crops */
int64_t vg_vit_workspace_bytes(const vg_vit* v, int n_crops) {
    if (!v || n_crops <= 0) return 0;
    int64_t Mp = pad128((int64_t)n_crops * v->T), W = v->width, es = v->dtype == 1 ? 2 : 4;
    int64_t Pp = pad128((int64_t)n_crops * (v->T - 1)), Kp = 3 * v->patch * v->patch;
    int64_t b = Mp * W * 4            // x (f32)
                + Mp * W * es         // h
                + Mp * (3 * W + 256) * es   // qkv (padded row stride)
                + Mp * 4 * W * es     // mlp
                + Pp * Kp * es        // patches
                + Pp * W * 4;         // patch-embed output (f32)
    if (v->ln_fold) b += Mp * W * 2 + Mp * (W / 64) * 8;      // fp16 copy of the residual + per-row partial statistics (folded LayerNorm)
    return b + 1024;
}

/* input_kind 0: f32 CHW crops [n,3,res,res]; 1: f16 CHW crops; d_feat: [n,out_dim] f32 */
int vg_vit_encode(vg_vit* v, const void* d_crops, int input_kind, int n_crops, void* d_workspace, float* d_feat,
                  void* stream) {
    if (!v || !d_crops || !d_workspace || !d_feat || n_crops <= 0 || input_kind < 0 || input_kind > 3) return VG_ERR_ARG;
    if (input_kind >= 2 && v->dtype != 1) return VG_ERR_ARG;
    if (input_kind == 3 && (v->patch * v->patch) % 128) return VG_ERR_ARG;       // K of the folded patch embedding: two 64-wide K-tiles at least
    hipStream_t st = (hipStream_t)stream;
    const int W = v->width, T = v->T, L = v->layers, H = v->heads;
    const int64_t M = (int64_t)n_crops * T, Mp = pad128(M), es = v->dtype == 1 ? 2 : 4;
    const int64_t P = (int64_t)n_crops * (T - 1), Pp = pad128(P), Kp = 3 * v->patch * v->patch;
    char* ws = (char*)d_workspace;
    float* x = (float*)ws;            ws += Mp * W * 4;
    void* h = ws;                     ws += Mp * W * es;
    // fp16: qkv rows padded by 64 halves.  The padding dates from k_gemm_f16, where 4608-byte rows aliased on the memory channels
    // (+20 %); with k_gemm_f16_pp64 the GEMM time no longer depends on it (13.50-13.58 ms per frame for 0 / 32 / 64 / 128 / 256),
    // but whole frames run 2.5 % faster with 0-128 than with 256 halves (57.0 vs 55.3 frames/s): less to write and to stride over.
    const int qkv_ld = v->dtype == 1 ? 3 * W + 64 : 3 * W;
    void* qkv = ws;                   ws += Mp * (3 * W + 256) * es;
    void* mlp = ws;                   ws += Mp * 4 * W * es;
    void* patches = ws;               ws += Pp * Kp * es;
    float* pe = (float*)ws;           ws += Pp * W * 4;
    const size_t pe_bytes = (size_t)(Pp * W * 4);      // dead once the embedding kernel has run: scratch of the residual GEMMs' split-K tails
    const bool fold = v->ln_fold;
    f16* x16 = (f16*)ws;              if (fold) ws += Mp * W * 2;
    LnPartial* lnst = (LnPartial*)ws;
    if (fold) { const int frc = vit_fold_ln(v, st); if (frc) return frc; }
    if (input_kind == 3) { const int frc = vit_fold_conv1(v, st); if (frc) return frc; }

    auto need = [&](const std::string& n) -> void* {
        auto it = v->w.find(n);
        return it == v->w.end() ? nullptr : it->second;
    };
    const char* top[] = {"conv1.weight", "class_embedding", "positional_embedding", "ln_pre.weight", "ln_pre.bias",
                         "ln_post.weight", "ln_post.bias", "proj"};
    for (const char* n : top)
        if (!need(n)) {
            fprintf(stderr, "[vilgod_hip] vg_vit_encode: weight %s not set\n", n);
            return VG_ERR_ARG;
        }
    // im2col (skipped when the renderer already wrote patch rows)
    if (input_kind >= 2) patches = const_cast<void*>(d_crops);
    else {
        int blocks = (int)((P * Kp + 255) / 256);
        if (blocks > 65535 * 8) blocks = 65535 * 8;
        if (v->dtype == 1) {
            if (input_kind == 0)
                hipLaunchKernelGGL((k_im2col<float, f16>), dim3(blocks), dim3(256), 0, st, (const float*)d_crops, (f16*)patches, n_crops, v->res, v->patch);
            else
                hipLaunchKernelGGL((k_im2col<f16, f16>), dim3(blocks), dim3(256), 0, st, (const f16*)d_crops, (f16*)patches, n_crops, v->res, v->patch);
        } else {
            if (input_kind == 0)
                hipLaunchKernelGGL((k_im2col<float, float>), dim3(blocks), dim3(256), 0, st, (const float*)d_crops, (float*)patches, n_crops, v->res, v->patch);
            else
                hipLaunchKernelGGL((k_im2col<f16, float>), dim3(blocks), dim3(256), 0, st, (const f16*)d_crops, (float*)patches, n_crops, v->res, v->patch);
        }
        VG_LAUNCH_CHECK();
    }
    // input_kind 3: single-channel patch rows [P, patch^2] times the folded K = patch^2 weight; its constant rides in the positional table
    const float* pos_tab = (const float*)need(input_kind == 3 ? "positional_embedding#1ch" : "positional_embedding");
    int rc = input_kind == 3 ? launch_gemm<EPI_NONE_F32>(v, patches, need("conv1.weight#1ch"), nullptr, pe, nullptr, (int)Pp, W, v->patch * v->patch, st)
                             : launch_gemm<EPI_NONE_F32>(v, patches, need("conv1.weight"), nullptr, pe, nullptr, (int)Pp, W, (int)Kp, st);
    if (rc) return rc;
    if (!pos_tab) return VG_ERR_ARG;
    const bool rh = v->resid_h;
    bool ln1_done = false;
    f16* xh = (f16*)x;                // the residual stream lives in the same workspace region, as fp16 when `rh`
    // the stream as an fp16 pair (VG_VIT_RESID_HL): hi = x16 (what in_proj / c_fc read anyway), lo in the fp32 stream's region; every full-row
    // residual GEMM is then a folded-LayerNorm producer (the last block runs on the compact fp32 class rows: the same condition as `cls_only`)
    const bool hl = v->resid_hl && fold && !rh && v->dtype == 1 && L > 1 && W % 256 == 0 &&
                    pad128(n_crops) * W * 16 + pad128(n_crops) * (W / 64) * 8 <= Mp * (3 * W + 256) * es;
    f16* xlo = (f16*)x;
    if (rh)
        hipLaunchKernelGGL((k_embed_lnpre<f16>), dim3((unsigned)((Mp + 3) / 4)), dim3(256), 0, st, pe, (const float*)need("class_embedding"),
                           pos_tab, (const float*)need("ln_pre.weight"),
                           (const float*)need("ln_pre.bias"), xh, (int)M, T, W, (int)Mp);
    else {
        // fp16 tower: ln_1 of block 0 is computed by the same kernel (writes h next to x)
        ln1_done = v->dtype == 1 && L > 0;
        const float* lw1 = ln1_done ? (const float*)need("transformer.resblocks.0.ln_1.weight") : nullptr;
        const float* lb1 = ln1_done ? (const float*)need("transformer.resblocks.0.ln_1.bias") : nullptr;
        if (ln1_done && (!lw1 || !lb1)) return VG_ERR_ARG;
        hipLaunchKernelGGL((k_embed_lnpre<float>), dim3((unsigned)((Mp + 3) / 4)), dim3(256), 0, st, pe, (const float*)need("class_embedding"),
                           pos_tab, (const float*)need("ln_pre.weight"),
                           (const float*)need("ln_pre.bias"), x, (int)M, T, W, (int)Mp, lw1, lb1, ln1_done ? (f16*)h : (f16*)nullptr,
                           hl ? (f16*)x16 : (f16*)nullptr, hl ? xlo : (f16*)nullptr);
    }
    VG_LAUNCH_CHECK();
    for (int l = 0; l < L; ++l) {
        std::string p = "transformer.resblocks." + std::to_string(l) + ".";
        const char* names[] = {"ln_1.weight", "ln_1.bias", "attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight",
                               "attn.out_proj.bias", "ln_2.weight", "ln_2.bias", "mlp.c_fc.weight", "mlp.c_fc.bias",
                               "mlp.c_proj.weight", "mlp.c_proj.bias"};
        void* wp[12];
        for (int i = 0; i < 12; ++i) {
            wp[i] = need(p + names[i]);
            if (!wp[i]) {
                fprintf(stderr, "[vilgod_hip] vg_vit_encode: weight %s%s not set\n", p.c_str(), names[i]);
                return VG_ERR_ARG;
            }
        }
        void *fw1 = nullptr, *fw2 = nullptr;
        const float *f1c1 = nullptr, *f1c2 = nullptr, *f2c1 = nullptr, *f2c2 = nullptr;
        if (fold) {
            fw1 = need(p + "attn.in_proj_weight#ln"); f1c1 = (const float*)need(p + "attn.in_proj_weight#c1"); f1c2 = (const float*)need(p + "attn.in_proj_weight#c2");
            fw2 = need(p + "mlp.c_fc.weight#ln");     f2c1 = (const float*)need(p + "mlp.c_fc.weight#c1");     f2c2 = (const float*)need(p + "mlp.c_fc.weight#c2");
            if (!fw1 || !f1c1 || !f1c2 || !fw2 || !f2c1 || !f2c2) return VG_ERR_ARG;
        }
        if (l == 0 && ln1_done) {
            // h already holds ln_1(x) of block 0 (k_embed_lnpre)
        } else if (fold) {
            // ln_1 rides in in_proj: raw fp16 residual (x16) and row statistics (lnst) from the previous block's c_proj epilogue
        } else if (rh)
            hipLaunchKernelGGL((k_layernorm<f16, f16>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, (const f16*)xh, (const float*)wp[0], (const float*)wp[1], (f16*)h, (int)M, W, 1);
        else if (v->dtype == 1)
            hipLaunchKernelGGL((k_layernorm<f16>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, x, (const float*)wp[0], (const float*)wp[1], (f16*)h, (int)M, W, 1);
        else
            hipLaunchKernelGGL((k_layernorm<float>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, x, (const float*)wp[0], (const float*)wp[1], (float*)h, (int)M, W, 1);
        VG_LAUNCH_CHECK();
        // The LAST block: only the class token's row of its output is ever read (ln_post(x[:, 0, :]) @ proj, model.py:235-238), and
        // after the attention every row is processed on its own (out_proj, ln_2, c_fc, QuickGELU, c_proj and the two residual adds).
        // So the block computes all keys and values, the class token's attention row, and then runs its three remaining GEMMs on
        // the n_crops class-token rows alone (compacted, padded to the row tile) instead of on n_crops x T rows: the same value
        // per element -- a row's dot products do not depend on which rows share its tile -- for 1 / T of the work.
        const int64_t Mc_ = pad128(n_crops);
        const bool cls_fits = Mc_ * W * 16 + Mc_ * (W / 64) * 8 <= Mp * (3 * W + 256) * es;       // the compact buffers live in the qkv buffer
                                                                                                   // (statistics sized like vg_vit_workspace_bytes: W / 64 partials per row, k_gemm_f16_x2's count)
        const bool cls_only = v->cls_last && fold && !rh && l == L - 1 && L > 1 && cls_fits;
        if (cls_only && W % 256 == 0) {
            // ... and of in_proj's three thirds that block needs K and V for every row but Q for the class-token rows only (round 5: a
            // third of the launch, 270 -> ~190 us per frame): in_proj's K / V rows (weight rows [W, 3W)) over all tokens, its Q rows over
            // the compacted class-token rows (the MLP buffer is free here), the queries scattered back to where the attention reads
            // them.  The other tokens' Q columns keep block L - 2's values: the attention's first query tile computes rows from them
            // that nobody reads (a query row never meets another query row).  Same kernel, same K loop per row: the same bits.
            const int nst = v->gemm_w4 ? W >> 7 : W >> 8;        // partial statistics per row: per 128 columns (k_gemm_f16_w4) or per 256
            char* cb = (char*)mlp;
            f16* x16q = (f16*)cb;              cb += Mc_ * W * 2;
            f16* qc = (f16*)cb;                cb += Mc_ * W * 2;
            LnPartial* lnq = (LnPartial*)cb;
            rc = launch_gemm<EPI_BIAS, 1>(v, x16, (const f16*)fw1 + (size_t)W * W, f1c2 + W, (f16*)qkv + W, nullptr, (int)Mp, 2 * W, W, st, qkv_ld,
                                          f1c1 + W, lnst);
            if (rc) return rc;
            hipLaunchKernelGGL(k_gather_cls_ln, dim3((unsigned)Mc_), dim3(256), 0, st, (const f16*)x16, (const LnPartial*)lnst, x16q, lnq, n_crops, T, W, nst);
            VG_LAUNCH_CHECK();
            rc = launch_gemm<EPI_BIAS, 1>(v, x16q, fw1, f1c2, qc, nullptr, (int)Mc_, W, W, st, 0, f1c1, lnq);
            if (rc) return rc;
            hipLaunchKernelGGL(k_scatter_cls_q, dim3((unsigned)n_crops), dim3(256), 0, st, (const f16*)qc, (f16*)qkv, T, W, qkv_ld);
            VG_LAUNCH_CHECK();
        } else
        if (fold && !(l == 0 && ln1_done))
            rc = launch_gemm<EPI_BIAS, 1>(v, x16, fw1, f1c2, qkv, nullptr, (int)Mp, 3 * W, W, st, qkv_ld, f1c1, lnst);
        else
            rc = launch_gemm<EPI_BIAS>(v, h, wp[2], (const float*)wp[3], qkv, nullptr, (int)Mp, 3 * W, W, st, qkv_ld);
        if (rc) return rc;
        if (v->dtype == 1) {
            {
                rc = launch_attention<false>((const f16*)qkv, (f16*)h, T, W, H, qkv_ld, n_crops * H, nullptr, st, cls_only ? 1 : 7, v->att_tr, v->att_stagger);
                if (rc) return rc;
            }
        } else {
            size_t lds = ((size_t)T * 65 + (size_t)T * 64 + 16 * (size_t)T) * sizeof(float);
            static bool attr = false;
            if (!attr) {
                VG_CHECK(hipFuncSetAttribute((const void*)k_attention_f32, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr = true;
            }
            hipLaunchKernelGGL(k_attention_f32, dim3(n_crops * H), dim3(256), lds, st, (const float*)qkv, (float*)h, T, W, H);
        }
        VG_LAUNCH_CHECK();
        if (cls_only) {
            // compact buffers inside the qkv buffer (dead once the attention has run): residual rows, attention rows, fp16 copy, hidden
            // activations, row statistics
            const int64_t Mc = pad128(n_crops);
            char* cb = (char*)qkv;
            float* xc = (float*)cb;            cb += Mc * W * 4;
            f16* hc = (f16*)cb;                cb += Mc * W * 2;
            f16* x16c = (f16*)cb;              cb += Mc * W * 2;
            f16* mlpc = (f16*)cb;              cb += Mc * 4 * W * 2;
            LnPartial* lnc = (LnPartial*)cb;
            hipLaunchKernelGGL(k_gather_cls, dim3((unsigned)Mc), dim3(256), 0, st, (const f16*)h, (const float*)x, hc, xc, n_crops, T, W,
                               hl ? (const f16*)x16 : (const f16*)nullptr, hl ? (const f16*)xlo : (const f16*)nullptr);
            VG_LAUNCH_CHECK();
            rc = launch_gemm<EPI_BIAS_RESID, 2>(v, hc, wp[4], (const float*)wp[5], nullptr, xc, (int)Mc, W, W, st, 0, nullptr, lnc, x16c);
            if (rc) return rc;
            rc = launch_gemm<EPI_BIAS_GELU, 1>(v, x16c, fw2, f2c2, mlpc, nullptr, (int)Mc, 4 * W, W, st, 0, f2c1, lnc);
            if (rc) return rc;
            rc = launch_gemm<EPI_BIAS_RESID>(v, mlpc, wp[10], (const float*)wp[11], nullptr, xc, (int)Mc, W, 4 * W, st);
            if (rc) return rc;
            hipLaunchKernelGGL((k_head<float>), dim3(n_crops), dim3(256), W * sizeof(float), st, (const float*)xc, (const float*)need("ln_post.weight"),
                               (const float*)need("ln_post.bias"), (const float*)need("proj"), d_feat, 1, W, v->out_dim);
            VG_LAUNCH_CHECK();
            return VG_OK;
        }
        rc = rh ? launch_gemm<EPI_BIAS_RESID_H>(v, h, wp[4], (const float*)wp[5], nullptr, x, (int)Mp, W, W, st)
           : hl ? launch_gemm<EPI_BIAS_RESID_HL, 2>(v, h, wp[4], (const float*)wp[5], nullptr, (float*)xlo, (int)Mp, W, W, st, 0, nullptr, lnst, x16)
           : fold ? launch_gemm<EPI_BIAS_RESID, 2>(v, h, wp[4], (const float*)wp[5], nullptr, x, (int)Mp, W, W, st, 0, nullptr, lnst, x16, pe, pe_bytes)
                  : launch_gemm<EPI_BIAS_RESID>(v, h, wp[4], (const float*)wp[5], nullptr, x, (int)Mp, W, W, st, 0, nullptr, nullptr, nullptr, pe, pe_bytes);
        if (rc) return rc;
        if (fold) {
            // ln_2 rides in c_fc
        } else if (rh)
            hipLaunchKernelGGL((k_layernorm<f16, f16>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, (const f16*)xh, (const float*)wp[6], (const float*)wp[7], (f16*)h, (int)M, W, 1);
        else if (v->dtype == 1)
            hipLaunchKernelGGL((k_layernorm<f16>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, x, (const float*)wp[6], (const float*)wp[7], (f16*)h, (int)M, W, 1);
        else
            hipLaunchKernelGGL((k_layernorm<float>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, x, (const float*)wp[6], (const float*)wp[7], (float*)h, (int)M, W, 1);
        VG_LAUNCH_CHECK();
        rc = fold ? launch_gemm<EPI_BIAS_GELU, 1>(v, x16, fw2, f2c2, mlp, nullptr, (int)Mp, 4 * W, W, st, 0, f2c1, lnst)
                  : launch_gemm<EPI_BIAS_GELU>(v, h, wp[8], (const float*)wp[9], mlp, nullptr, (int)Mp, 4 * W, W, st);
        if (rc) return rc;
        // the last block's c_proj has no LayerNorm consumer in a GEMM (ln_post reads the class token's fp32 row in k_head)
        rc = rh ? launch_gemm<EPI_BIAS_RESID_H>(v, mlp, wp[10], (const float*)wp[11], nullptr, x, (int)Mp, W, 4 * W, st)
           : (hl && l + 1 < L) ? launch_gemm<EPI_BIAS_RESID_HL, 2>(v, mlp, wp[10], (const float*)wp[11], nullptr, (float*)xlo, (int)Mp, W, 4 * W, st, 0, nullptr, lnst, x16)
           : (fold && l + 1 < L) ? launch_gemm<EPI_BIAS_RESID, 2>(v, mlp, wp[10], (const float*)wp[11], nullptr, x, (int)Mp, W, 4 * W, st, 0, nullptr, lnst, x16, pe, pe_bytes)
                  : launch_gemm<EPI_BIAS_RESID>(v, mlp, wp[10], (const float*)wp[11], nullptr, x, (int)Mp, W, 4 * W, st, 0, nullptr, nullptr, nullptr, pe, pe_bytes);
        if (rc) return rc;
    }
    if (rh)
        hipLaunchKernelGGL((k_head<f16>), dim3(n_crops), dim3(256), W * sizeof(float), st, (const f16*)xh, (const float*)need("ln_post.weight"),
                           (const float*)need("ln_post.bias"), (const float*)need("proj"), d_feat, T, W, v->out_dim);
    else
        hipLaunchKernelGGL((k_head<float>), dim3(n_crops), dim3(256), W * sizeof(float), st, x, (const float*)need("ln_post.weight"),
                           (const float*)need("ln_post.bias"), (const float*)need("proj"), d_feat, T, W, v->out_dim);
    VG_LAUNCH_CHECK();
    return VG_OK;
}

/* k_attention_f16 alone: softmax(q k^T / 8) v per (crop, head) on the padded qkv rows in_proj writes (model.py:175-187 via
 * nn.MultiheadAttention); exposed so that the kernel can be unit-tested against a plain fp32 attention. */
int vg_attention(const void* d_qkv, void* d_out, int n_crops, int T, int W, int heads, int ld, void* stream) {
    if (!d_qkv || !d_out || n_crops <= 0 || T > AT_MAXT || heads * 64 != W) return VG_ERR_ARG;
    const char* tr_env = getenv("VG_ATT_TR");           // handle-less test entry point: read per call, on the caller's thread
    const char* sg_env = getenv("VG_ATT_STAGGER");
    return launch_attention<false>((const f16*)d_qkv, (f16*)d_out, T, W, heads, ld, n_crops * heads, nullptr, (hipStream_t)stream, 7,
                                   !(tr_env && atoi(tr_env) == 0), !(sg_env && atoi(sg_env) == 0));
}

#ifdef VG_DEV      // development aids (tools/dev/vilgod_hip_dev.h): ablations, cycle-stamp traces
#include "dev/vit_dev_entry.inc"
#endif  // VG_DEV

/* C = X @ Wt^T (+ epilogue), exposed for unit tests / micro-benchmarks of the GEMM itself.
 * dtype 1: X,Wt f16, M%128==0, N%128==0, K%64==0; dtype 0: f32, M%64, N%64, K%16.
 * epi 0: +bias -> C (compute dtype)   1: +bias, QuickGELU -> C   2: resid(f32) += acc + bias   3: C f32, no bias
 * epi 4 (dtype 1, N % 256 == 0, K % 64 == 0): d_resid holds fp16 [M,N]: resid = f16(resid + f16(acc + bias)) */
int vg_gemm(int dtype, int epi, const void* d_X, const void* d_Wt, const float* d_bias, void* d_C, float* d_resid,
            int M, int N, int K, void* stream) {
    vg_vit v;
    v.dtype = dtype;
    hipStream_t st = (hipStream_t)stream;
    switch (epi) {
        case 0: return launch_gemm<EPI_BIAS>(&v, d_X, d_Wt, d_bias, d_C, d_resid, M, N, K, st);
        case 1: return launch_gemm<EPI_BIAS_GELU>(&v, d_X, d_Wt, d_bias, d_C, d_resid, M, N, K, st);
        case 2: return launch_gemm<EPI_BIAS_RESID>(&v, d_X, d_Wt, d_bias, d_C, d_resid, M, N, K, st);
        case 3: return launch_gemm<EPI_NONE_F32>(&v, d_X, d_Wt, d_bias, d_C, d_resid, M, N, K, st);
        case 4: return dtype == 1 ? launch_gemm<EPI_BIAS_RESID_H>(&v, d_X, d_Wt, d_bias, d_C, d_resid, M, N, K, st) : VG_ERR_ARG;
    }
    return VG_ERR_ARG;
}

/* vg_gemm epi 2 (dtype 1) with scratch for the split-K tail (launch_gemm / splitk_plan): what the tower's residual GEMMs run.  Row tiles
 * beyond the last complete round of n_cu tiles are computed K-split through d_scratch (>= 2 MB per tail tile; without enough the
 * launch is not split).  Opt-in: VG_GEMM_SPLITK=n (at most n parts per tile; 8) in the environment, read per launch. */
int vg_gemm_resid_splitk(const void* d_X, const void* d_Wt, const float* d_bias, float* d_resid, int M, int N, int K,
                         void* d_scratch, int64_t scratch_bytes, void* stream) {
    if (!d_X || !d_Wt || !d_bias || !d_resid || scratch_bytes < 0) return VG_ERR_ARG;
    vg_vit v;
    v.dtype = 1;
    return launch_gemm<EPI_BIAS_RESID>(&v, d_X, d_Wt, d_bias, nullptr, d_resid, M, N, K, (hipStream_t)stream, 0, nullptr, nullptr, nullptr,
                                       (float*)d_scratch, (size_t)scratch_bytes);
}

/* on != 0: record a HIP event pair around every projection-GEMM launch of vg_vit_encode (on the launch stream);
 * resets the collected samples. */
int vg_vit_profile(vg_vit* v, int on) {
    if (!v) return VG_ERR_ARG;
    if (on && !v->prof_init) {
        for (int i = 0; i < 2 * VG_PROF_MAX; ++i) VG_CHECK(hipEventCreate(&v->prof_ev[i]));
        v->prof_init = true;
    }
    v->prof_on = on;
    v->prof_n = 0;
    return VG_OK;
}

/* synchronises, then returns the number of GEMM launches sampled, their summed duration (ms) and algorithmic FLOPs;
 * kind -1: every projection GEMM, 0: k_gemm_f16 / k_gemm_f32, 1: k_gemm_f16_pp64 (pp16 when K % 64 != 0) */
int vg_vit_profile_read_kind(vg_vit* v, int kind, int32_t* h_launches, double* h_ms, double* h_flops) {
    if (!v || !h_launches || !h_ms || !h_flops) return VG_ERR_ARG;
    *h_launches = 0; *h_ms = 0; *h_flops = 0;
    for (int i = 0; i < v->prof_n; ++i) {
        if (kind >= 0 && v->prof_kind[i] != kind) continue;
        VG_CHECK(hipEventSynchronize(v->prof_ev[2 * i + 1]));
        float ms = 0.f;
        VG_CHECK(hipEventElapsedTime(&ms, v->prof_ev[2 * i], v->prof_ev[2 * i + 1]));
        *h_ms += ms;
        *h_flops += v->prof_flops[i];
        ++*h_launches;
    }
    return VG_OK;
}

int vg_vit_profile_read(vg_vit* v, int32_t* h_launches, double* h_ms, double* h_flops) {
    return vg_vit_profile_read_kind(v, -1, h_launches, h_ms, h_flops);
}

int vg_clip_scores(const float* d_feat, int n, int dim, const float* d_text, int n_classes, float* d_probs,
                   int32_t* d_top1, float* d_top1_score, void* stream);

/* ---- captured classification (SURVEY 7 step 7 / BASELINE config 5: hipGraph-captured loop) -----------------------------------
 * The launch-heavy, shape-stable part of a frame -- the ~150 kernels of vg_vit_encode + vg_clip_scores -- as ONE hipGraph per
 * distinct crop count, captured on first use and replayed afterwards.  The cache belongs to one worker (one stream, one set of
 * persistent buffers: patches, workspace, features, scores): the key is the crop count plus every pointer baked into the nodes. */
struct vg_graph_key {
    int n_crops, input_kind, dim, n_classes;
    const void *crops, *ws, *feat, *text, *probs, *top1, *score;
    const void* tower;            // the handle and the generation of its device tensors (weights are baked into the kernel nodes)
    uint64_t weights_gen;
    bool operator<(const vg_graph_key& o) const { return memcmp(this, &o, sizeof(*this)) < 0; }
};
struct vg_graph_entry { hipGraphExec_t exec; long last_use; };
struct vg_graph_cache {
    std::map<vg_graph_key, vg_graph_entry> graphs;
    long captured = 0, replayed = 0, evicted = 0, clock = 0;
    int limit = 32;               // graphs kept (least recently used one is destroyed beyond that): a real sequence has a new crop
                                  // count almost every frame, an unbounded cache would grow to hundreds of graphExecs per worker
};

int vg_graph_cache_create(vg_graph_cache** out) {
    if (!out) return VG_ERR_ARG;
    *out = new vg_graph_cache();
    return VG_OK;
}

void vg_graph_cache_destroy(vg_graph_cache* c) {
    if (!c) return;
    for (auto& kv : c->graphs) (void)hipGraphExecDestroy(kv.second.exec);
    delete c;
}

int vg_graph_cache_limit(vg_graph_cache* c, int max_graphs) {
    if (!c || max_graphs < 1) return VG_ERR_ARG;
    c->limit = max_graphs;
    return VG_OK;
}

int vg_graph_cache_stats2(const vg_graph_cache* c, int64_t* h_captured, int64_t* h_replayed, int64_t* h_evicted, int64_t* h_live) {
    if (!c) return VG_ERR_ARG;
    if (h_captured) *h_captured = c->captured;
    if (h_replayed) *h_replayed = c->replayed;
    if (h_evicted) *h_evicted = c->evicted;
    if (h_live) *h_live = (int64_t)c->graphs.size();
    return VG_OK;
}

int vg_graph_cache_stats(const vg_graph_cache* c, int64_t* h_captured, int64_t* h_replayed) {
    if (!c) return VG_ERR_ARG;
    if (h_captured) *h_captured = c->captured;
    if (h_replayed) *h_replayed = c->replayed;
    return VG_OK;
}

int vg_vit_classify_graph(vg_vit* v, vg_graph_cache* c, const void* d_crops, int input_kind, int n_crops, void* d_workspace, float* d_feat,
                          const float* d_text, int dim, int n_classes, float* d_probs, int32_t* d_top1, float* d_top1_score, void* stream) {
    if (!v || !c || n_crops <= 0 || !stream) return VG_ERR_ARG;          // capture needs a real (non-default) stream
    // the first encode of a process runs as plain launches: it sets the kernels' dynamic-LDS attributes (hipFuncSetAttribute), which
    // must not happen inside a capture
    // (per handle, and the other threads wait for it: it also builds the handle's derived tensors, vit_fold_ln / vit_fold_conv1)
    const auto cold = [&]() { return !v->warmed.load() || (v->ln_fold && !v->fold_ready.load()) || (input_kind == 3 && !v->conv1_ready.load()); };
    if (cold()) {
        static std::mutex warm_mtx;
        std::lock_guard<std::mutex> lock(warm_mtx);
        if (cold()) {
            int rc = vg_vit_encode(v, d_crops, input_kind, n_crops, d_workspace, d_feat, stream);
            if (!rc) rc = vg_clip_scores(d_feat, n_crops, dim, d_text, n_classes, d_probs, d_top1, d_top1_score, stream);
            if (!rc) VG_CHECK(hipStreamSynchronize((hipStream_t)stream));
            v->warmed.store(true);
            return rc;
        }
    }
    if (v->prof_on) {                                                   // event pairs cannot live in a graph: plain launches while profiling
        const int rc = vg_vit_encode(v, d_crops, input_kind, n_crops, d_workspace, d_feat, stream);
        return rc ? rc : vg_clip_scores(d_feat, n_crops, dim, d_text, n_classes, d_probs, d_top1, d_top1_score, stream);
    }
    hipStream_t st = (hipStream_t)stream;
    vg_graph_key key;
    memset(&key, 0, sizeof(key));
    key.n_crops = n_crops; key.input_kind = input_kind; key.dim = dim; key.n_classes = n_classes;
    key.crops = d_crops; key.ws = d_workspace; key.feat = d_feat; key.text = d_text; key.probs = d_probs; key.top1 = d_top1; key.score = d_top1_score;
    key.tower = v; key.weights_gen = v->weights_gen.load();
    auto it = c->graphs.find(key);
    if (it == c->graphs.end()) {
        // thread-local capture mode: the other workers keep launching on their streams meanwhile
        hipGraph_t graph = nullptr;
        VG_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        int rc = vg_vit_encode(v, d_crops, input_kind, n_crops, d_workspace, d_feat, stream);
        if (!rc) rc = vg_clip_scores(d_feat, n_crops, dim, d_text, n_classes, d_probs, d_top1, d_top1_score, stream);
        hipError_t e = hipStreamEndCapture(st, &graph);
        if (rc || e != hipSuccess || !graph) {
            if (graph) (void)hipGraphDestroy(graph);
            fprintf(stderr, "[vilgod_hip] vg_vit_classify_graph: capture failed (rc %d, %s)\n", rc, hipGetErrorString(e));
            return rc ? rc : VG_ERR_HIP;
        }
        hipGraphExec_t exec = nullptr;
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        VG_CHECK(e);
        while ((int)c->graphs.size() >= c->limit) {                    // least recently used out (its launches are stream-ordered
            auto lru = c->graphs.begin();                               // before this point; hipGraphExecDestroy defers the release)
            for (auto j = c->graphs.begin(); j != c->graphs.end(); ++j)
                if (j->second.last_use < lru->second.last_use) lru = j;
            VG_CHECK(hipStreamSynchronize(st));
            (void)hipGraphExecDestroy(lru->second.exec);
            c->graphs.erase(lru);
            c->evicted++;
        }
        it = c->graphs.emplace(key, vg_graph_entry{exec, 0}).first;
        c->captured++;
    }
    it->second.last_use = ++c->clock;
    VG_CHECK(hipGraphLaunch(it->second.exec, st));
    c->replayed++;
    return VG_OK;
}

int vg_clip_scores(const float* d_feat, int n, int dim, const float* d_text, int n_classes, float* d_probs,
                   int32_t* d_top1, float* d_top1_score, void* stream) {
    if (n <= 0) return VG_OK;
    if (!d_feat || !d_text || !d_probs || !d_top1 || !d_top1_score || n_classes <= 0 || n_classes > 64) return VG_ERR_ARG;
    hipLaunchKernelGGL(k_clip_scores, dim3(n), dim3(64), 0, (hipStream_t)stream, d_feat, d_text, d_probs, d_top1,
                       d_top1_score, n, dim, n_classes);
    VG_LAUNCH_CHECK();
    return VG_OK;
}

}  // extern "C"
