// ABI version + small device utilities shared by the host side.
#include "common.h"
#include "vilgod_hip.h"

extern "C" int vg_abi_version(void) { return 1; }

// CU-masked streams (include/vilgod_hip.h, "execution resources")
extern "C" int vg_stream_create_cu_mask(void** out_stream, const uint32_t* h_cu_mask, int n_words) {
    if (!out_stream || !h_cu_mask || n_words < 1) return VG_ERR_ARG;
    bool any = false;
    for (int i = 0; i < n_words; ++i) any = any || h_cu_mask[i] != 0u;
    if (!any) return VG_ERR_ARG;                      // a stream that may run nowhere
    hipStream_t st = nullptr;
    VG_CHECK(hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, h_cu_mask));
    *out_stream = (void*)st;
    return VG_OK;
}

extern "C" int vg_stream_destroy(void* stream) {
    if (!stream) return VG_ERR_ARG;
    VG_CHECK(hipStreamDestroy((hipStream_t)stream));
    return VG_OK;
}

extern "C" int vg_device_cu_count(int32_t* h_count) {
    if (!h_count) return VG_ERR_ARG;
    int dev = 0, n = 0;
    VG_CHECK(hipGetDevice(&dev));
    VG_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    *h_count = n;
    return VG_OK;
}
