"""k_gemm_f16_w4 (4 waves, assembly K loop; VG_GEMM_W4=1) against k_gemm_f16_pp64 (VG_GEMM_W4=0) and hipBLASLt (torch.addmm):
bit-equality on small shapes, then the four ViT-B/16 projection shapes at M = 64 256 with launches of the three interleaved in one
process on one box (VERDICT r5 task 1: in_proj <= 222 us, c_fc <= 300, c_proj <= 252, out_proj <= 75 with the bias epilogue).

    python tools/bench_gemm_w4.py [--rounds 7] [--iters 10] [--no-blas] [--json out.json]
"""
import argparse
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilgod_amd._lib import lib, ptr, stream_ptr, check   # noqa: E402


def run(epi, X, W, b, C, R, w4):
    os.environ['VG_GEMM_W4'] = '1' if w4 else '0'
    M, K = X.shape
    N = W.shape[0]
    check(lib.vg_gemm(1, epi, ptr(X), ptr(W), ptr(b), ptr(C), ptr(R), M, N, K, stream_ptr()))


def equality(dev):
    g = torch.Generator(device=dev).manual_seed(1)
    bad = 0
    for (M, N, K) in [(256, 256, 192), (256, 256, 256), (512, 512, 320), (512, 768, 768), (256, 2304, 768), (256, 768, 3072), (768, 3072, 768)]:
        X = (torch.randn(M, K, device=dev, generator=g)).half()
        W = (torch.randn(N, K, device=dev, generator=g) * 0.05).half()
        b = torch.randn(N, device=dev, generator=g)
        for epi in (0, 1, 2, 3):
            outs = []
            for w4 in (False, True):
                C16 = torch.zeros(M, N, dtype=torch.float16, device=dev)
                C32 = torch.zeros(M, N, dtype=torch.float32, device=dev)
                R = torch.arange(M * N, device=dev, dtype=torch.float32).reshape(M, N) * 1e-3
                run(epi, X, W, b, C32 if epi == 3 else C16, R, w4)
                torch.cuda.synchronize()
                outs.append((C32 if epi == 3 else (R if epi == 2 else C16)).clone())
            same = torch.equal(outs[0], outs[1])
            if not same:
                bad += 1
                d = (outs[0].float() - outs[1].float()).abs()
                print(f'  MISMATCH M={M} N={N} K={K} epi={epi}: max |d| {d.max().item():.4g}, {(d > 0).sum().item()} elements', flush=True)
    print('equality:', 'all bit-identical' if bad == 0 else f'{bad} cases differ', flush=True)
    return bad == 0


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--M', type=int, default=64256)
    ap.add_argument('--no-blas', action='store_true')
    ap.add_argument('--skip-equality', action='store_true')
    ap.add_argument('--json', default=None)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    out = {'M': args.M, 'equal': None, 'shapes': {}}
    if not args.skip_equality:
        out['equal'] = equality(dev)
    M = args.M
    g = torch.Generator(device=dev).manual_seed(0)
    for name, epi, N, K in [('in_proj', 0, 2304, 768), ('out_proj', 0, 768, 768), ('c_fc', 0, 3072, 768), ('c_proj', 0, 768, 3072),
                            ('c_fc+gelu', 1, 3072, 768), ('out_proj+resid', 2, 768, 768), ('c_proj+resid', 2, 768, 3072)]:
        X = torch.randn(M, K, device=dev, generator=g).half()
        W = (torch.randn(N, K, device=dev, generator=g) * 0.02).half()
        b32 = torch.randn(N, device=dev, generator=g)
        b16 = b32.half()
        C = torch.zeros(M, N, dtype=torch.float16, device=dev)
        R = torch.zeros(M if epi == 2 else 1, N if epi == 2 else 1, dtype=torch.float32, device=dev)
        Wt = W.t()
        fns = {'pp64': lambda: run(epi, X, W, b32, C, R, False), 'w4': lambda: run(epi, X, W, b32, C, R, True)}
        if not args.no_blas and epi == 0:
            fns['addmm'] = lambda: torch.addmm(b16, X, Wt, out=C)
        for fn in fns.values():
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        t = {k: [] for k in fns}
        for _ in range(args.rounds):
            for k, fn in fns.items():
                t[k].append(timed(fn, args.iters))
        fl = 2.0 * M * N * K
        rec = {k: {'us_median': round(statistics.median(v), 1), 'us_min': round(min(v), 1), 'tflops_median': round(fl / statistics.median(v) / 1e6, 1)}
               for k, v in t.items()}
        out['shapes'][f'{name} N={N} K={K} epi={epi}'] = rec
        print(name, json.dumps(rec), flush=True)
    if args.json:
        with open(args.json, 'w') as f:
            json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
