"""Interleaved A/B of the projection GEMM of TWO builds of the library in one process (development aid):
    python tools/bench_gemm_libs.py vilgod_amd/libvilgod_hip.so vilgod_amd/libvilgod_hip_<tag>.so     (tools/dev/build_variant.sh)
in_proj- and c_fc-shaped launches (epilogues 0 and 1) at M = CROPS x 197 (padded), median of 9 rounds of 5 launches, outputs compared."""
import ctypes, os, sys, statistics
import torch
dev = torch.device('cuda:0')
torch.zeros(1, device=dev)
libs = [ctypes.CDLL(os.path.abspath(p)) for p in sys.argv[1:3]]
for l in libs:
    l.vg_gemm.restype = ctypes.c_int
    l.vg_gemm.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
M = (int(os.environ.get('CROPS', '337')) * 197 + 255) // 256 * 256
st = torch.cuda.current_stream().cuda_stream
for N, K, epi in [(2304, 768, 0), (3072, 768, 1)]:
    X = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.05).half()
    b = torch.randn(N, device=dev)
    C = [torch.zeros(M, N, dtype=torch.float16, device=dev) for _ in libs]
    res = [[], []]
    for rnd in range(11):
        for k, l in enumerate(libs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                assert l.vg_gemm(1, epi, X.data_ptr(), W.data_ptr(), b.data_ptr(), C[k].data_ptr(), None, M, N, K, st) == 0
            e1.record(); torch.cuda.synchronize()
            if rnd >= 2: res[k].append(e0.elapsed_time(e1) / 5 * 1000)
    a_, b_ = statistics.median(res[0]), statistics.median(res[1])
    print(f'M={M} N={N} K={K} epi={epi}: {os.path.basename(sys.argv[1])} {a_:7.1f} us   {os.path.basename(sys.argv[2])} {b_:7.1f} us   ratio {b_ / a_:.3f}   '
          f'outputs equal: {torch.equal(C[0], C[1])}', flush=True)
