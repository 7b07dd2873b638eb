"""Per-wave cycle breakdown of the tiled GEMM variants (main loop / waiting for the DMA / barrier / epilogue)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dev'))
from devlib import lib, ptr, stream_ptr, check        # the development build (tools/dev)
dev = torch.device('cuda:0')
M = (240 * 197 + 255) // 256 * 256
vars_ = [int(v) for v in (sys.argv[1:] or ['22'])]
for N, K, ldc in [(2304, 768, 2560), (3072, 768, 3072), (768, 3072, 768)]:
    X = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.05).half()
    b = torch.randn(N, device=dev); C = torch.zeros(M, ldc, dtype=torch.float16, device=dev)
    for var in vars_:
        tr = torch.zeros(1 << 19, dtype=torch.int64, device=dev)
        for _ in range(int(os.environ.get('REPS', '3'))):
            check(lib.vg_gemm_trace(var, ptr(X), ptr(W), ptr(b), ptr(C), ptr(tr), M, N, K, ldc, stream_ptr()))
        torch.cuda.synchronize()
        nk = K // (64 if var in (32, 33, 40) else 32)
        if True:
            t = tr.view(-1, 8).cpu().double(); t = t[t[:, 7] > 0]
            for g in ((0,) if var == 40 else (0, 1)):
                tg = t[(t[:, 6] >= 4) == bool(g)]
                main, wait, bar, epi, load, mma = tg.mean(0).tolist()[:6]; ghz = ((tg[:, 0] + tg[:, 3]) / tg[:, 7]).median().item() * 0.1
                print(f'N={N} K={K} var={var} group {g}: per K-step: load {load/nk:5.0f}  wait {wait/nk:5.0f}  barriers {bar/nk:5.0f}  '
                      f'mma {mma/nk:5.0f}   K-step {main/nk:6.0f}  epilogue {epi:6.0f}  clock {ghz:.2f} GHz')
            continue
