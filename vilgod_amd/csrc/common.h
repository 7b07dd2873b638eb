// Shared helpers for the gfx950 kernels of libvilgod_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define VG_OK 0
#define VG_ERR_ARG 1
#define VG_ERR_HIP 2
#define VG_ERR_CAPACITY 3

#define VG_CHECK(expr)                                                                      \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            fprintf(stderr, "[vilgod_hip] %s:%d %s -> %s\n", __FILE__, __LINE__, #expr,     \
                    hipGetErrorString(_e));                                                 \
            return VG_ERR_HIP;                                                              \
        }                                                                                   \
    } while (0)

#define VG_LAUNCH_CHECK() VG_CHECK(hipGetLastError())

static inline int vg_div_up(int a, int b) { return (a + b - 1) / b; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, DEVICE): the attribute belongs to the device's code object, so a
// process-wide `static bool` would leave a second device of the process without it (its launches with > 64 KB of dynamic LDS fail).
// One bit per device in an atomic word; setting it twice is harmless, so racing threads need no lock.
#include <atomic>
struct VgPerDeviceOnce { std::atomic<unsigned long long> done{0}; };
static inline int vg_max_dynamic_lds(const void* kernel, int bytes, VgPerDeviceOnce& once) {
    int dev = 0;
    VG_CHECK(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (once.done.load(std::memory_order_acquire) & bit) return VG_OK;
    VG_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    once.done.fetch_or(bit, std::memory_order_release);
    return VG_OK;
}
#define VG_MAX_DYNAMIC_LDS(kernel, bytes)                                                   \
    do {                                                                                    \
        static VgPerDeviceOnce _once;                                                       \
        const int _rc = vg_max_dynamic_lds((const void*)(kernel), (int)(bytes), _once);    \
        if (_rc != VG_OK) return _rc;                                                       \
    } while (0)

#define WAVE 64

// order-preserving float <-> uint key (for radix select / atomic min-max on signed floats)
__device__ __forceinline__ uint32_t vg_fkey(float f) {
    uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float vg_fkey_inv(uint32_t k) {
    uint32_t b = k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    return __uint_as_float(b);
}

__device__ __forceinline__ float vg_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float vg_wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float vg_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
