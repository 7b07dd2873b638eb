"""Is the folded LayerNorm's consumer epilogue (LN = 1) slower than the plain one on the SAME operands?  (development build)
In the pipeline in_proj with the fold reads 275 us per launch and the one unfolded launch of the first block 242: code or data?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dev'))
from devlib import lib, ptr, stream_ptr, check
dev = torch.device('cuda:0')
M = (int(os.environ.get('CROPS', '331')) * 197 + 255) // 256 * 256
for N, K, epi, scale in [(2304, 768, 0, 0.5), (3072, 768, 1, 0.5), (2304, 768, 0, 3.0), (3072, 768, 1, 3.0)]:
    X = (torch.randn(M, K, device=dev) * scale).half(); W = (torch.randn(N, K, device=dev) * 0.05).half()
    b = torch.randn(N, device=dev); c1 = torch.randn(N, device=dev) * 0.1
    stats = torch.stack([torch.randn(M, K // 256, device=dev) * 0.1, torch.rand(M, K // 256, device=dev) * 256.0 * scale * scale], dim=-1).contiguous()
    C = torch.zeros(M, N, dtype=torch.float16, device=dev)
    tot = {'plain': 0.0, 'ln': 0.0}
    for rep in range(5):
        for kind in ('plain', 'ln'):
            def launch():
                if kind == 'plain': return lib.vg_gemm(1, epi, ptr(X), ptr(W), ptr(b), ptr(C), None, M, N, K, stream_ptr())
                return lib.vg_gemm_ln_consumer(epi, ptr(X), ptr(W), ptr(b), ptr(c1), ptr(stats), ptr(C), M, N, K, stream_ptr())
            for _ in range(2): check(launch())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): launch()
            e1.record(); torch.cuda.synchronize()
            tot[kind] += e0.elapsed_time(e1) / 10 * 1000
    print(f'N={N} K={K} epi={epi} |x| ~ {scale}: plain {tot["plain"] / 5:7.1f} us   LN consumer {tot["ln"] / 5:7.1f} us   ({tot["ln"] / tot["plain"] - 1:+.1%})')
