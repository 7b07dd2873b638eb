// Micro-benchmark: LDS-DMA (global_load_lds 16 B/lane) streaming rate per CU for two source patterns with the same bytes:
//   A: a 1-KB piece = 16 rows x  64 B (K-step 32 halves)      B: a 1-KB piece = 8 rows x 128 B (K-step 64 halves)
// 8 waves per workgroup, one workgroup per CU, every wave streams its own rows of a [rows][K] fp16 matrix that is
// L2/MALL resident (re-read), counted vmcnt keeps 8 pieces in flight per wave.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
template <int SEG>   // bytes per row segment: 64 or 128
__global__ __launch_bounds__(512, 1) void k_dma(const _Float16* __restrict__ X, int K, int rows_per_wg, int iters, int* sink, int shares) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = SEG / 16;                 // lanes per row
    constexpr int RPP = 64 / LPR;                 // rows per piece
    const int nkt = K * 2 / SEG;
    const _Float16* base = X + (size_t)((blockIdx.x % shares) * rows_per_wg + wave * (rows_per_wg / 8) + lane / LPR) * K + (lane % LPR) * 8;
    char* dst = smem + wave * 16384;
    int n = 0;
    for (int it = 0; it < iters; ++it)
        for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {          // 4 pieces = 4*RPP rows of this wave's share
                __builtin_amdgcn_global_load_lds((glb_void*)(base + (size_t)p * RPP * K + kt * (SEG / 2)),
                                                 (lds_void*)(dst + ((n++ & 15) * 1024)), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (sink && smem[threadIdx.x] == 123 && n == -1) *sink = 1;
}
int main() {
    const int K = 768, WG = 256, rows_per_wg = 8 * 64;      // every wave owns 64 rows: 4 pieces x 16 rows (A) or 2 x (4 x 8) (B)
    const size_t rows = (size_t)WG * rows_per_wg;
    _Float16* X; hipMalloc(&X, rows * K * 2); hipMemset(X, 0, rows * K * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shares : {256, 4})
    for (int seg : {64, 128, 64, 128}) {
        const int iters = 20;
        auto launch = [&]() {
            if (seg == 64) { hipFuncSetAttribute((const void*)k_dma<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
                hipLaunchKernelGGL(k_dma<64>, dim3(WG), dim3(512), 131072, 0, X, K, rows_per_wg, iters, nullptr, shares); }
            else { hipFuncSetAttribute((const void*)k_dma<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
                hipLaunchKernelGGL(k_dma<128>, dim3(WG), dim3(512), 131072, 0, X, K, rows_per_wg, iters, nullptr, shares); }
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // bytes: per wave per (it, kt): 4 pieces x 1 KB; rows covered: seg 64 -> 64 rows x 64 B; seg 128 -> 32 rows x 128 B
        const double bytes = (double)WG * 8 * iters * (K * 2 / seg) * 4 * 1024;
        printf("footprint %3d WG-shares (%.0f MB), segment %3d B: %.3f ms  %.2f TB/s  (%.1f B/clk/CU at 1.8 GHz)\n", shares, shares * rows_per_wg * K * 2 / 1e6, seg, ms, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 1.8e9);
    }
    return 0;
}
