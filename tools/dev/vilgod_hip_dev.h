/* Development entry points of libvilgod_hip_dev.so (python -m vilgod_amd.build --dev: the product sources compiled with -DVG_DEV).
 * NOT part of the product ABI (include/vilgod_hip.h): ablation variants and cycle-stamp builds of the GEMM / attention kernels and
 * the superseded GEMM kernels they are compared with, used by tools/bench_gemm_*.py and tools/bench_attention.py only. */
#ifndef VILGOD_HIP_DEV_H
#define VILGOD_HIP_DEV_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* development aid: the f16 GEMM kernels by number, bias epilogue (k_gemm_f16: 0, ablations 1 no in-loop DMA, 2 DMA only,
 * 3 no epilogue; k_gemm_f16_pp (32x32x16, K-step 32): 22, 20 two phases per K-step, 21/23 five stages; k_gemm_f16_pp16: 30;
 * k_gemm_f16_pp64, the production kernel of rounds 2-5: 32) */
int vg_gemm_variant(int var, const void* d_X, const void* d_Wt, const float* d_bias, void* d_C, int M, int N, int K, int ldc,
                    void* stream);

/* development aid: k_gemm_f16_pp (var 20..23) and k_gemm_f16_pp64 (var 32 +bias, 33 +bias QuickGELU, 34 fp32 residual with d_C = the
 * float [M,N] stream, 35 fp16 residual with d_C = the half [M,N] stream) with per-wave cycle stamps.  d_trace receives, per
 * (workgroup, wave), eight int64: main-loop cycles, cycles in the counted vmcnt wait, cycles at barriers, prologue + epilogue
 * cycles, LOAD-segment cycles, MFMA-segment cycles, wave id, elapsed 100-MHz ticks.  var 50 (round 6): k_gemm_f16_w4's trace variant, twelve int64
 * per (tile, wave): entry -> block, block entry, K loop, M-wait / barrier / end-wait cycle sums, one stamp pair, epilogue, whole block, first, wave,
 * 100-MHz ticks (tools/dev/w4_trace.py; size d_trace for 48 * tiles).  Variants 32..35 append, after those
 * 64 * n_workgroups values, eight int64 per workgroup: entry and exit time (100-MHz ticks), XCC id << 32 | HW_ID, prologue
 * cycles, epilogue cycles, 1 -- size d_trace for 72 * (M/256) * (N/256) values (tools/bench_gemm_tiles.py). */
int vg_gemm_trace(int var, const void* d_X, const void* d_Wt, const float* d_bias, void* d_C, int64_t* d_trace, int M, int N,
                  int K, int ldc, void* stream);

/* k_gemm_f16_pp64 with the folded LayerNorm's consumer epilogue (epi 0 bias, 1 bias + QuickGELU) on caller-supplied row statistics
 * d_stats [M, K/256] (mean, m2) float pairs and d_c1 [N]: A/B against vg_gemm on the same operands (tools/bench_gemm_ln.py) */
int vg_gemm_ln_consumer(int epi, const void* d_X, const void* d_Wt, const float* d_bias, const float* d_c1, const void* d_stats,
                        void* d_C, int M, int N, int K, void* stream);

/* vg_attention (include/vilgod_hip.h) with cycle stamps: with d_trace != NULL the traced build runs and writes, per
 * (workgroup of the persistent grid min(n_crops*heads, 256), wave 0..6), eight int64 cycle sums: staging + barrier, next-item load
 * issue, S^T MFMA issue, max pass, exp pass, P/V^T/O^T issue, output, end barrier. */
int vg_attention_trace(const void* d_qkv, void* d_out, int n_crops, int T, int W, int heads, int ld, int64_t* d_trace, void* stream);


/* hierarchy stage on the device: 100 MHz time stamps of the last call's one-workgroup kernels (csrc/hdbscan_device.hip HD_STAMP):
 * [0..2] k_hd_tree_a start / nodes staged / Kruskal done; [4..13] k_hd_tree_bc start, staged, up sweeps, down sweeps, BFS numbering,
 * selection, epsilon, scan, owners, end */
int vg_hier_stamps(void* h, int64_t* h_out16);

#ifdef __cplusplus
}
#endif
#endif
