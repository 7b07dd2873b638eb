"""Like-for-like ceiling of the ViT projection GEMMs: `vg_gemm` (k_gemm_f16_pp64, bias epilogue, fp16 out)
against torch.matmul / torch.addmm fp16 (hipBLASLt) on the SAME tensors, same M = 64256 (= 326 crops x 197 tokens padded to 256-row
tiles), N(0,1) activations, the four shapes, launches interleaved A/B in one process on one box.

    python tools/bench_blas_ceiling.py [--rounds 6] [--iters 20]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilgod_amd._lib import lib, ptr, stream_ptr, check   # noqa: E402


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=6)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--M', type=int, default=64256)
    ap.add_argument('--sigma', type=float, default=1.0)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    M = args.M
    shapes = [('in_proj', 0, 2304, 768), ('out_proj', 0, 768, 768), ('c_fc', 0, 3072, 768), ('c_proj', 0, 768, 3072)]
    out = {'M': M, 'activations': f'N(0,{args.sigma})', 'weights': 'N(0,0.02)', 'shapes': {}}
    g = torch.Generator(device=dev).manual_seed(0)
    for name, epi, N, K in shapes:
        X = (torch.randn(M, K, device=dev, generator=g) * args.sigma).half()
        W = (torch.randn(N, K, device=dev, generator=g) * 0.02).half()
        b32 = torch.randn(N, device=dev, generator=g)
        b16 = b32.half()
        C = torch.zeros(M, N, dtype=torch.float16, device=dev)
        R = torch.zeros(1, dtype=torch.float32, device=dev)
        Wt = W.t()

        def ours():
            check(lib.vg_gemm(1, epi, ptr(X), ptr(W), ptr(b32), ptr(C), ptr(R), M, N, K, stream_ptr()))

        def blas_mm():
            torch.matmul(X, Wt, out=C)

        def blas_addmm():
            torch.addmm(b16, X, Wt, out=C)

        for fn in (ours, blas_mm, blas_addmm):
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        # outputs agree (fp16 rounding of an fp32 accumulation on both sides)
        ours(); mine = C.clone(); blas_addmm(); ref = C.clone()
        err = (mine.float() - ref.float()).abs().max().item()
        t = {'vg_gemm': [], 'matmul': [], 'addmm': []}
        for _ in range(args.rounds):                     # interleaved: A, B, C, A, B, C ...
            t['vg_gemm'].append(timed(ours, args.iters))
            t['matmul'].append(timed(blas_mm, args.iters))
            t['addmm'].append(timed(blas_addmm, args.iters))
        fl = 2.0 * M * N * K
        rec = {k: {'us_median': round(sorted(v)[len(v) // 2] * 1000, 1), 'us_min': round(min(v) * 1000, 1),
                   'tflops_median': round(fl / sorted(v)[len(v) // 2] / 1e9, 1)} for k, v in t.items()}
        rec['max_abs_diff_vs_addmm'] = err
        out['shapes'][f'{name} N={N} K={K}'] = rec
        print(name, json.dumps(rec), flush=True)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
