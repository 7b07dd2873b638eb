"""Interleaved A/B of two builds of the projection GEMM selected by an environment switch read per launch (development aid):
    python tools/bench_gemm_ab.py VG_GEMM_HS 1 0
Every ViT-B/16 shape and epilogue at M = CROPS x 197 (padded), launches of the two settings alternating in one process, median of 9."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilgod_amd._lib import lib, ptr, stream_ptr, check
var, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
dev = torch.device('cuda:0')
M = (int(os.environ.get('CROPS', '337')) * 197 + 255) // 256 * 256
for N, K, epis in [(2304, 768, (0,)), (3072, 768, (1,)), (768, 768, (2,)), (768, 3072, (2,))]:
    X = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.05).half()
    b = torch.randn(N, device=dev)
    for epi in epis:
        C = torch.zeros(M, N, dtype=torch.float16, device=dev)
        R = torch.zeros(M, N, dtype=torch.float32, device=dev)
        res = {va: [], vb: []}
        for rnd in range(11):
            for v in (va, vb):
                os.environ[var] = v
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): check(lib.vg_gemm(1, epi, ptr(X), ptr(W), ptr(b), ptr(C), ptr(R), M, N, K, stream_ptr()))
                e1.record(); torch.cuda.synchronize()
                if rnd >= 2: res[v].append(e0.elapsed_time(e1) / 5 * 1000)
        a_, b_ = statistics.median(res[va]), statistics.median(res[vb])
        print(f'M={M} N={N} K={K} epi={epi}: {var}={va} {a_:7.1f} us ({2.0*M*N*K/a_/1e6:6.1f} TF)   {var}={vb} {b_:7.1f} us ({2.0*M*N*K/b_/1e6:6.1f} TF)   ratio {a_/b_:.3f}', flush=True)
