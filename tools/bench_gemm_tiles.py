"""Where a 256 x 256 tile's time goes in k_gemm_f16_pp64 (trace variants 32 = +bias f16, 33 = +bias GELU f16, 34 = residual f32, 35 = residual f16):
prologue (entry -> first MMA), main loop, epilogue, and the dead time on a CU between one workgroup's exit and the next one's entry
(s_memrealtime stamps, 10 ns units, grouped by XCC / SE / CU id)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dev'))
from devlib import lib, ptr, stream_ptr, check        # the development build (tools/dev)
dev = torch.device('cuda:0')
M = (int(os.environ.get('CROPS', '325')) * 197 + 255) // 256 * 256
for var, N, K in [(32, 2304, 768), (33, 3072, 768), (34, 768, 768), (34, 768, 3072), (35, 768, 768), (35, 768, 3072)]:
    X = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.05).half()
    b = torch.randn(N, device=dev)
    C = torch.zeros(M, N, dtype=torch.float32 if var == 34 else torch.float16, device=dev)
    grid = (M // 256) * (N // 256)
    tr = torch.zeros(grid * 72 + 64, dtype=torch.int64, device=dev)
    for _ in range(3):
        check(lib.vg_gemm_trace(var, ptr(X), ptr(W), ptr(b), ptr(C), ptr(tr), M, N, K, N, stream_ptr()))
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); check(lib.vg_gemm_trace(var, ptr(X), ptr(W), ptr(b), ptr(C), ptr(tr), M, N, K, N, stream_ptr())); ev1.record()
    torch.cuda.synchronize()
    w = tr[:grid * 64].view(-1, 8).cpu().double(); w = w[w[:, 7] > 0]
    e = tr[grid * 64: grid * 72].view(-1, 8).cpu()
    e = e[e[:, 5] == 1]
    e[:, 2] = (e[:, 2] & 0xFF00) | ((e[:, 2] >> 32) & 0xF) << 16        # XCC id | SE / SH / CU id (HW_ID bits 15:8)
    ghz = ((w[:, 0] + w[:, 3]) / w[:, 7]).median().item() * 0.1
    main = w[:, 0].mean().item()
    pro, epi = e[:, 3].double().median().item(), e[:, 4].double().median().item()
    gaps, busy, span = [], 0.0, 0.0
    for key in e[:, 2].unique():
        r = e[e[:, 2] == key]
        r = r[r[:, 0].argsort()]
        gaps += (r[1:, 0] - r[:-1, 1]).tolist()
    gaps = torch.tensor(gaps if gaps else [0.0], dtype=torch.float64)
    t0, t1 = e[:, 0].min().item(), e[:, 1].max().item()
    life = (e[:, 1] - e[:, 0]).double().mean().item()
    print(f'var {var} N={N} K={K}: {grid} tiles on {len(e[:, 2].unique())} CUs, launch {ev0.elapsed_time(ev1) * 1e3:.0f} us '
          f'(first entry -> last exit {(t1 - t0) / 100:.0f} us), clock {ghz:.2f} GHz | per tile: prologue {pro:.0f} cyc, main {main:.0f}, '
          f'epilogue {epi:.0f}; WG life {life / 100:.2f} us; gap exit->next entry on a CU: median {gaps.median().item() * 10:.0f} ns, '
          f'mean {gaps.mean().item() * 10:.0f} ns, p90 {gaps.quantile(0.9).item() * 10:.0f} ns | '
          f'TF {2 * M * N * K / (ev0.elapsed_time(ev1) * 1e-3) / 1e12:.0f}')
