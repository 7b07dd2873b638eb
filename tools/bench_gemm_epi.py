"""Per-epilogue timing of the production f16 GEMM (vg_gemm) on the four ViT-B/16 projection shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilgod_amd._lib import lib, ptr, stream_ptr, check
dev = torch.device('cuda:0')
M = (int(os.environ.get('CROPS', '327')) * 197 + 255) // 256 * 256
for N, K, epis in [(2304, 768, (0,)), (3072, 768, (0, 1)), (768, 768, (0, 2)), (768, 3072, (0, 2)), (768, 768, (3,))]:
    X = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.05).half()
    b = torch.randn(N, device=dev)
    for epi in epis:
        C = torch.zeros(M, N, dtype=torch.float32 if epi == 3 else torch.float16, device=dev)
        R = torch.zeros(M, N, dtype=torch.float32, device=dev)
        for _ in range(3): check(lib.vg_gemm(1, epi, ptr(X), ptr(W), ptr(b), ptr(C), ptr(R), M, N, K, stream_ptr()))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): lib.vg_gemm(1, epi, ptr(X), ptr(W), ptr(b), ptr(C), ptr(R), M, N, K, stream_ptr())
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f'M={M} N={N} K={K} epi={epi}: {ms*1000:7.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF')
