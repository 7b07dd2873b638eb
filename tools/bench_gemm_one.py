"""One GEMM shape in a loop (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilgod_amd._lib import lib, ptr, stream_ptr, check
dev = torch.device('cuda:0')
M = (240 * 197 + 255) // 256 * 256
N, K, ldc = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
X = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.05).half()
b = torch.randn(N, device=dev); C = torch.zeros(M, ldc, dtype=torch.float16, device=dev)
for _ in range(6):
    check(lib.vg_gemm_variant(0, ptr(X), ptr(W), ptr(b), ptr(C), M, N, K, ldc, stream_ptr()))
torch.cuda.synchronize()
