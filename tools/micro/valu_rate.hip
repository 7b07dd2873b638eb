// Micro-benchmark: issue cost (cycles per wave-instruction) of the vector instructions the GEMM / attention epilogues are made of, on
// gfx950: plain and packed fp32 arithmetic, fp32 / fp16 transcendentals, conversions.  One workgroup per CU; 1, 2 or 4 waves per SIMD;
// every wave runs REP x 32 independent instructions of one kind on 8 register sets (no dependent chain) between two s_memtime stamps.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP 64
// 32 instructions per macro expansion, 8 independent destinations (v reads a 2-register pair for the packed forms)
#define BODY1(OP)                                                                                            \
    for (int it = 0; it < REP; ++it) {                                                                       \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                      \
            asm volatile(OP " %0, %0" : "+v"(a0)); asm volatile(OP " %0, %0" : "+v"(a1));                  \
            asm volatile(OP " %0, %0" : "+v"(a2)); asm volatile(OP " %0, %0" : "+v"(a3));                  \
            asm volatile(OP " %0, %0" : "+v"(a4)); asm volatile(OP " %0, %0" : "+v"(a5));                  \
            asm volatile(OP " %0, %0" : "+v"(a6)); asm volatile(OP " %0, %0" : "+v"(a7));                  \
        }                                                                                                    \
    }
#define BODY2(OP)                                                                                            \
    for (int it = 0; it < REP; ++it) {                                                                       \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                      \
            asm volatile(OP " %0, %0, %1" : "+v"(a0) : "v"(c)); asm volatile(OP " %0, %0, %1" : "+v"(a1) : "v"(c)); \
            asm volatile(OP " %0, %0, %1" : "+v"(a2) : "v"(c)); asm volatile(OP " %0, %0, %1" : "+v"(a3) : "v"(c)); \
            asm volatile(OP " %0, %0, %1" : "+v"(a4) : "v"(c)); asm volatile(OP " %0, %0, %1" : "+v"(a5) : "v"(c)); \
            asm volatile(OP " %0, %0, %1" : "+v"(a6) : "v"(c)); asm volatile(OP " %0, %0, %1" : "+v"(a7) : "v"(c)); \
        }                                                                                                    \
    }
#define BODY3(OP)                                                                                            \
    for (int it = 0; it < REP; ++it) {                                                                       \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                      \
            asm volatile(OP " %0, %0, %1, %1" : "+v"(a0) : "v"(c)); asm volatile(OP " %0, %0, %1, %1" : "+v"(a1) : "v"(c)); \
            asm volatile(OP " %0, %0, %1, %1" : "+v"(a2) : "v"(c)); asm volatile(OP " %0, %0, %1, %1" : "+v"(a3) : "v"(c)); \
            asm volatile(OP " %0, %0, %1, %1" : "+v"(a4) : "v"(c)); asm volatile(OP " %0, %0, %1, %1" : "+v"(a5) : "v"(c)); \
            asm volatile(OP " %0, %0, %1, %1" : "+v"(a6) : "v"(c)); asm volatile(OP " %0, %0, %1, %1" : "+v"(a7) : "v"(c)); \
        }                                                                                                    \
    }

typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k_rate(long long* out, float seed) {
    const int tid = threadIdx.x;
    long long t0, t1;
    if (KIND < 100) {
        float a0 = seed + tid, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, c = 1.0001f;
        t0 = __builtin_amdgcn_s_memtime();
        if (KIND == 0) { BODY2("v_add_f32") }
        if (KIND == 1) { BODY2("v_mul_f32") }
        if (KIND == 2) { BODY3("v_fma_f32") }
        if (KIND == 3) { BODY1("v_exp_f32") }
        if (KIND == 4) { BODY1("v_rcp_f32") }
        if (KIND == 5) { BODY1("v_exp_f16") }
        if (KIND == 6) { BODY1("v_rcp_f16") }
        if (KIND == 7) { BODY2("v_cvt_pk_f16_f32") }
        if (KIND == 8) { BODY2("v_pk_mul_f16") }
        if (KIND == 9) { BODY3("v_pk_fma_f16") }
        if (KIND == 10) { BODY1("v_mov_b32") }
        if (KIND == 11) { BODY2("v_max_f32") }
        if (KIND == 12) { BODY1("v_rsq_f32") }
        if (KIND == 13) { BODY1("v_cvt_f16_f32") }
        if (KIND == 14) { BODY1("v_cvt_f32_f16") }
        t1 = __builtin_amdgcn_s_memtime();
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0] = 1;
    } else {
        f2 a0 = {seed + tid, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f, c = {1.0001f, 0.9999f};
        t0 = __builtin_amdgcn_s_memtime();
        if (KIND == 100) { BODY2("v_pk_add_f32") }
        if (KIND == 101) { BODY2("v_pk_mul_f32") }
        if (KIND == 102) { BODY3("v_pk_fma_f32") }
        t1 = __builtin_amdgcn_s_memtime();
        if (a0.x + a1.x + a2.x + a3.x + a4.y + a5.y + a6.y + a7.y == 12345.678f) out[0] = 1;
    }
    if ((tid & 63) == 0) out[1 + blockIdx.x * (blockDim.x >> 6) + (tid >> 6)] = t1 - t0;
}

template <int KIND>
static void run(const char* name, long long* d_out) {
    for (int threads : {256, 512, 1024}) {
        const int waves = 256 * (threads / 64);
        hipMemset(d_out, 0, 8 * (1 + waves));
        hipLaunchKernelGGL(k_rate<KIND>, dim3(256), dim3(threads), 0, 0, d_out, 0.5f);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(256), dim3(threads), 0, 0, d_out, 0.5f);
        hipDeviceSynchronize();
        std::vector<long long> h(1 + waves);
        hipMemcpy(h.data(), d_out, 8 * (1 + waves), hipMemcpyDeviceToHost);
        std::sort(h.begin() + 1, h.end());
        const double med = (double)h[1 + waves / 2];
        // cycles per wave-instruction as the wave sees them, and the SIMD's issue cost = that / (waves per SIMD)
        printf("%-18s %d waves/SIMD: %7.2f cycles per instruction per wave  -> %6.2f per SIMD slot\n", name, threads / 256, med / (REP * 32.0),
               med / (REP * 32.0) / (threads / 256));
    }
}

int main() {
    long long* d_out;
    hipMalloc(&d_out, 8 * (1 + 256 * 16));
    run<0>("v_add_f32", d_out);
    run<1>("v_mul_f32", d_out);
    run<2>("v_fma_f32", d_out);
    run<11>("v_max_f32", d_out);
    run<10>("v_mov_b32", d_out);
    run<100>("v_pk_add_f32", d_out);
    run<101>("v_pk_mul_f32", d_out);
    run<102>("v_pk_fma_f32", d_out);
    run<3>("v_exp_f32", d_out);
    run<4>("v_rcp_f32", d_out);
    run<12>("v_rsq_f32", d_out);
    run<5>("v_exp_f16", d_out);
    run<6>("v_rcp_f16", d_out);
    run<7>("v_cvt_pk_f16_f32", d_out);
    run<13>("v_cvt_f16_f32", d_out);
    run<14>("v_cvt_f32_f16", d_out);
    run<8>("v_pk_mul_f16", d_out);
    run<9>("v_pk_fma_f16", d_out);
    return 0;
}
