// Where do the workgroups of a CU-masked stream run?  Launches 4096 workgroups on streams created with
// hipExtStreamCreateWithCUMask for the masks vilgod_amd/streams.py builds (low 8 r bits = "front", the rest = "tower") and prints, per
// XCD, which (shader engine, CU) pairs executed at least one workgroup.  Build: hipcc --offload-arch=gfx950 -O3 -o cu_mask_probe cu_mask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <set>
#include <vector>
__global__ void k_where(unsigned* out, int spin) {
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);        // HW_REG_XCC_ID
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_REG_HW_ID: cu [11:8], sh [12], se [15:13]
        out[blockIdx.x] = ((xcc & 0xF) << 16) | (hw & 0xFFFF);
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);                // keep the CU busy so the dispatcher has to spread out
}
int main() {
    int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    printf("device reports %d CUs\n", ncu);
    const int G = 4096; unsigned* d; hipMalloc(&d, G * 4); std::vector<unsigned> h(G);
    auto run = [&](const char* name, hipStream_t st) {
        hipMemsetAsync(d, 0xFF, G * 4, st);
        hipLaunchKernelGGL(k_where, dim3(G), dim3(256), 0, st, d, 200);
        hipStreamSynchronize(st);
        hipMemcpy(h.data(), d, G * 4, hipMemcpyDeviceToHost);
        std::set<unsigned> per[16]; std::set<unsigned> all;
        for (int i = 0; i < G; ++i) { const unsigned x = (h[i] >> 16) & 0xF, se = (h[i] >> 13) & 7, sh = (h[i] >> 12) & 1, cu = (h[i] >> 8) & 0xF;
            per[x].insert((se << 8) | (sh << 4) | cu); all.insert((x << 12) | (se << 8) | (sh << 4) | cu); }
        printf("%-28s: %3zu distinct CUs;", name, all.size());
        for (int x = 0; x < 8; ++x) printf(" xcd%d:%zu", x, per[x].size());
        printf("\n   xcd0 (se.cu):");
        for (unsigned v : per[0]) printf(" %u.%u", v >> 8, v & 0xF);
        printf("\n");
    };
    hipStream_t s0; hipStreamCreate(&s0); run("unrestricted", s0);
    for (int r : {1, 2, 4, 8}) {
        unsigned front[8] = {0}, tower[8];
        for (int i = 0; i < 8 * r; ++i) front[i / 32] |= 1u << (i % 32);
        for (int w = 0; w < 8; ++w) tower[w] = ~front[w];
        hipStream_t sf, st; char nm[64];
        if (hipExtStreamCreateWithCUMask(&sf, 8, front) != hipSuccess || hipExtStreamCreateWithCUMask(&st, 8, tower) != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed\n"); return 1; }
        snprintf(nm, sizeof nm, "front r=%d (low %d bits)", r, 8 * r); run(nm, sf);
        snprintf(nm, sizeof nm, "tower r=%d (the other bits)", r); run(nm, st);
    }
    return 0;
}
