"""Development aid: the variants of k_gemm_f16_w4's K loop (csrc/gen_gemm_w4.py VARIANTS; VG_GEMM_W4 = 1 + index, development
library) interleaved with k_gemm_f16_pp64 (0) in one process, bias epilogue, M = 64 256.  Ablation variants compute wrong numbers.

    VILGOD_HIP_LIB=vilgod_amd/libvilgod_hip_dev.so python tools/dev/w4_variants.py 0 1 2 3 4 5 6
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vilgod_amd._lib import lib, ptr, stream_ptr, check   # noqa: E402

vals = sys.argv[1:] or ['0', '1']
dev = torch.device('cuda:0')
M = int(os.environ.get('M', '64256'))
g = torch.Generator(device=dev).manual_seed(0)
for name, N, K in [('in_proj', 2304, 768), ('out_proj', 768, 768), ('c_proj', 768, 3072)]:
    X = torch.randn(M, K, device=dev, generator=g).half()
    W = (torch.randn(N, K, device=dev, generator=g) * 0.02).half()
    b = torch.randn(N, device=dev, generator=g)
    C = torch.zeros(M, N, dtype=torch.float16, device=dev)
    R = torch.zeros(1, dtype=torch.float32, device=dev)
    t = {v: [] for v in vals}
    for rnd in range(9):
        for v in vals:
            os.environ['VG_GEMM_W4'] = v
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                check(lib.vg_gemm(1, 0, ptr(X), ptr(W), ptr(b), ptr(C), ptr(R), M, N, K, stream_ptr()))
            e1.record()
            torch.cuda.synchronize()
            if rnd >= 2:
                t[v].append(e0.elapsed_time(e1) / 8 * 1000)
    ref = None
    eq = {}
    for v in ['0'] + vals:
        os.environ['VG_GEMM_W4'] = v
        C.zero_()
        check(lib.vg_gemm(1, 0, ptr(X), ptr(W), ptr(b), ptr(C), ptr(R), M, N, K, stream_ptr()))
        torch.cuda.synchronize()
        if ref is None:
            ref = C.clone()
        eq[v] = 'eq' if torch.equal(ref, C) else 'NE'
    print(name, '  '.join(f'{v}: {statistics.median(t[v]):6.1f} {eq[v]}' for v in vals), flush=True)
