"""Residual GEMMs (N = 768: out_proj K = 768, c_proj K = 3072) with and without the split-K tail, interleaved on one box.
    python tools/bench_gemm_splitk.py [rows ...]        (row tiles of 256 token rows; default 256 257 258 261 266 272 280 298)
Prints us per launch for VG_GEMM_SPLITK=0 / 8 (median of 5 interleaved rounds of 10 launches)."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vilgod_amd._lib import lib, ptr, stream_ptr, check  # noqa: E402

rows_list = [int(a) for a in sys.argv[1:]] or [256, 257, 258, 261, 266, 272, 280, 298]
dev = torch.device('cuda:0')
N = 768
scratch = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
for K in (768, 3072):
    W = (torch.randn(N, K) * 0.05).half().to(dev)
    bias = torch.randn(N).to(dev)
    for rows in rows_list:
        M = rows * 256
        X = (torch.randn(M, K) * 0.5).half().to(dev)
        R = torch.zeros(M, N, device=dev)

        def run(mode, n=10):
            os.environ['VG_GEMM_SPLITK'] = mode
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                check(lib.vg_gemm_resid_splitk(ptr(X), ptr(W), ptr(bias), ptr(R), M, N, K, ptr(scratch), scratch.numel(), stream_ptr()))
            torch.cuda.synchronize()
            return 1e6 * (time.perf_counter() - t0) / n
        for m in ('0', '8'):
            run(m, 3)
        t = {'0': [], '8': []}
        for _ in range(5):
            for m in ('0', '8'):
                t[m].append(run(m))
        a, b = statistics.median(t['0']), statistics.median(t['8'])
        print(f'K {K:5d} rows {rows:4d} ({rows * 3:4d} tiles): unsplit {a:7.1f} us, split tail {b:7.1f} us ({100 * (b - a) / a:+5.1f} %)', flush=True)
        del X, R
