"""Residual GEMMs (epi 2) of the ViT-B/16 shapes with and without RI (accumulators initialised from the residual tile, write-only
epilogue): launches of the two forms interleaved in one process, between them a 400 MB write that pushes the residual stream out of the
256 MB memory-side cache as the c_fc output does in the pipeline (COLD=1, default) or nothing (COLD=0)."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilgod_amd._lib import lib, ptr, stream_ptr, check
dev = torch.device('cuda:0')
M = (int(os.environ.get('CROPS', '337')) * 197 + 255) // 256 * 256
cold = os.environ.get('COLD', '1') != '0'
junk = torch.zeros(100 * 1024 * 1024, dtype=torch.float32, device=dev)
for N, K in [(768, 768), (768, 3072)]:
    X = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.05).half()
    b = torch.randn(N, device=dev)
    R = torch.randn(M, N, dtype=torch.float32, device=dev)
    res = {'1': [], '0': []}
    for rnd in range(12):
        for ri in ('1', '0'):
            os.environ['VG_GEMM_RI'] = ri
            if cold: junk.fill_(float(rnd))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib.vg_gemm(1, 2, ptr(X), ptr(W), ptr(b), None, ptr(R), M, N, K, stream_ptr()))
            e1.record(); torch.cuda.synchronize()
            if rnd >= 2: res[ri].append(e0.elapsed_time(e1) * 1000)
        R.mul_(0.5)
    for ri in ('1', '0'):
        us = statistics.median(res[ri])
        print(f'M={M} N={N} K={K} residual epilogue, RI={ri} ({"cold" if cold else "warm"}): median {us:7.1f} us  min {min(res[ri]):7.1f}  {2.0*M*N*K/us/1e6:7.1f} TF', flush=True)
