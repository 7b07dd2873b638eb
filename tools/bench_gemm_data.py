"""Does the projection GEMM's speed depend on the operand VALUES (power / clock management) or on where the weights live?
c_fc shape (N=3072, K=768) with the shipped kernel: random / zero / constant / sparse activations, and weights at shifted addresses."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilgod_amd._lib import lib, ptr, stream_ptr, check
dev = torch.device('cuda:0')
M = (325 * 197 + 255) // 256 * 256
N, K = 3072, 768
b = torch.randn(N, device=dev); C = torch.zeros(M, N, dtype=torch.float16, device=dev)
Wr = (torch.randn(N, K, device=dev) * 0.05).half()
def run(X, W, tag):
    for _ in range(3): check(lib.vg_gemm(1, 1, ptr(X), ptr(W), ptr(b), ptr(C), None, M, N, K, stream_ptr()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): lib.vg_gemm(1, 1, ptr(X), ptr(W), ptr(b), ptr(C), None, M, N, K, stream_ptr())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f'{tag:46s} {ms * 1000:7.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF')
Xn = torch.randn(M, K, device=dev)
run(Xn.half(), Wr, 'X ~ N(0,1), W ~ N(0,0.05)')
run((Xn * 0.05).half(), Wr, 'X ~ N(0,0.05)')
run((Xn * 8).half(), Wr, 'X ~ N(0,8)')
run(torch.zeros(M, K, device=dev).half(), Wr, 'X = 0')
run(torch.ones(M, K, device=dev).half(), Wr, 'X = 1')
run((Xn * (torch.rand(M, K, device=dev) < 0.1)).half(), Wr, 'X 90 % zeros')
run(Xn.half(), torch.zeros(N, K, device=dev).half(), 'W = 0')
big = torch.zeros(N * K + 65536, dtype=torch.float16, device=dev)
for off in (0, 64, 1024, 2048 + 64, 32768):
    W2 = big[off:off + N * K].view(N, K); W2.copy_(Wr)
    run(Xn.half(), W2, f'W at +{off * 2} bytes')
