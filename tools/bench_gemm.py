"""Times the four projection GEMM shapes of ViT-B/16 at a frame's worth of tokens (240 crops x 197)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilgod_amd._lib import lib, ptr, stream_ptr, check

def main():
    dev = torch.device('cuda:0')
    M = (240 * 197 + 255) // 256 * 256
    shapes = [('qkv', 0, 2304, 768), ('out_proj', 2, 768, 768), ('c_fc', 1, 3072, 768), ('c_proj', 2, 768, 3072)]
    tot_t = tot_f = 0
    for name, epi, N, K in shapes:
        X = (torch.randn(M, K, device=dev) * 0.5).half()
        W = (torch.randn(N, K, device=dev) * 0.05).half()
        b = torch.randn(N, device=dev)
        C = torch.zeros(M, N, dtype=torch.float16, device=dev)
        R = torch.zeros(M, N, dtype=torch.float32, device=dev)
        for _ in range(3):
            check(lib.vg_gemm(1, epi, ptr(X), ptr(W), ptr(b), ptr(C), ptr(R), M, N, K, stream_ptr()))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        it = 20
        for _ in range(it):
            lib.vg_gemm(1, epi, ptr(X), ptr(W), ptr(b), ptr(C), ptr(R), M, N, K, stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / it
        fl = 2.0 * M * N * K
        tot_t += ms; tot_f += fl
        print(f'{name:9s} M={M} N={N} K={K}: {ms*1000:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s')
    print(f'per layer GEMM: {tot_t*1000:.1f} us, {tot_f/tot_t/1e9:.1f} TFLOP/s; x12 = {tot_t*12:.2f} ms')

if __name__ == '__main__':
    main()
