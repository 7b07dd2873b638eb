"""Per-tile timeline of k_gemm_f16_pp64 (development build, vg_gemm_trace var 32..34): prologue / main loop / epilogue cycles of every
workgroup, and how the workgroups of one CU follow each other (gaps, lockstep across CUs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dev'))
from devlib import lib, ptr, stream_ptr, check
dev = torch.device('cuda:0')
M = (int(os.environ.get('CROPS', '327')) * 197 + 255) // 256 * 256
var = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for N, K in [(2304, 768), (768, 768), (768, 3072)]:
    ldc = N
    X = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * 0.05).half()
    b = torch.randn(N, device=dev)
    C = torch.zeros(M, ldc, dtype=torch.float32 if var == 34 else torch.float16, device=dev)
    tr = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
    for _ in range(3):
        check(lib.vg_gemm_trace(var, ptr(X), ptr(W), ptr(b), ptr(C), ptr(tr), M, N, K, ldc, stream_ptr()))
    torch.cuda.synchronize()
    nwg = (M // 256) * (N // 256)
    w = tr[:nwg * 64].view(nwg, 8, 8).cpu().double()          # per wave: main, wait, bar, rest, load, mma, wave, wall
    e = tr[nwg * 64: nwg * 64 + nwg * 8].view(nwg, 8).cpu()   # per workgroup: wall entry, wall end, hw id, prologue, epilogue
    t0 = e[:, 0].min().item()
    ent, end = (e[:, 0] - t0).double() * 0.01, (e[:, 1] - t0).double() * 0.01       # us
    pro, epi, main = e[:, 3].double(), e[:, 4].double(), w[:, 0, 0]
    hw = e[:, 2]
    cu = ((hw >> 32) & 0xFFFFFFFF) * 1000 + (hw & 0xFFFF)      # (XCC id, HW_ID) -> a key per CU
    print(f'N={N} K={K} var={var}: {nwg} tiles, launch {end.max().item():.1f} us; cycles per tile: prologue {pro.median().item():.0f} (p90 {pro.quantile(0.9).item():.0f}), '
          f'main {main.median().item():.0f}, epilogue {epi.median().item():.0f} (p10 {epi.quantile(0.1).item():.0f}, p90 {epi.quantile(0.9).item():.0f})')
    # rounds: tiles sorted by entry time; entry-time spread inside each round of 256
    order = ent.argsort()
    for r in range(0, min(nwg, 256 * 4), 256):
        seg = order[r:r + 256]
        print(f'   round {r // 256}: entries {ent[seg].min().item():7.1f} .. {ent[seg].max().item():7.1f} us (p10 {ent[seg].quantile(0.1).item():7.1f}, p90 {ent[seg].quantile(0.9).item():7.1f}), '
              f'ends {end[seg].min().item():7.1f} .. {end[seg].max().item():7.1f}, epilogue median {epi[seg].median().item():.0f} cycles')
