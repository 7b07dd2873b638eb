import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dev'))
from devlib import lib, ptr, stream_ptr, check        # the development build (tools/dev)
dev = torch.device('cuda:0')
M = (int(os.environ.get("CROPS", "240")) * 197 + 255) // 256 * 256
vars_ = [int(v) for v in (sys.argv[1:] or ['0', '22', '30', '32'])]      # k_gemm_f16, _pp (32x32), _pp16, _pp64
for N, K, ldc in [(2304, 768, 2560), (3072, 768, 3072), (768, 768, 768), (768, 3072, 768)]:
    X = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.05).half()
    b = torch.randn(N, device=dev); C = torch.zeros(M, ldc, dtype=torch.float16, device=dev)
    for var in vars_:
        for _ in range(3): check(lib.vg_gemm_variant(var, ptr(X), ptr(W), ptr(b), ptr(C), M, N, K, ldc, stream_ptr()))
        torch.cuda.synchronize()
        ref = (X.float() @ W.float().t() + b); err = (C[:, :N].float() - ref).abs().max().item() if ldc >= N else -1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): lib.vg_gemm_variant(var, ptr(X), ptr(W), ptr(b), ptr(C), M, N, K, ldc, stream_ptr())
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f'N={N} K={K} ldc={ldc} var={var}: {ms*1000:7.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF  maxerr {err:.3g}')
