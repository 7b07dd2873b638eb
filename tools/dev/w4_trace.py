"""Cycle stamps of k_gemm_f16_w4's K loop (development build, vg_gemm_trace var 50 = the trace variant of csrc/gen_gemm_w4.py).
Per wave: entry -> asm block, asm prologue (DMA of the first tiles + first fragment reads), K loop, and inside the loop the cycles
at the M wait (vmcnt + lgkmcnt), at the barrier, at the end-of-iteration wait; each of the three minus the calibration pair.

    python tools/dev/w4_trace.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from devlib import lib, ptr, stream_ptr, check   # noqa: E402

dev = torch.device('cuda:0')
M = int(os.environ.get('M', '64256'))
for name, N, K in [('in_proj', 2304, 768), ('out_proj', 768, 768), ('c_proj', 768, 3072)]:
    X = torch.randn(M, K, device=dev).half()
    W = (torch.randn(N, K, device=dev) * 0.02).half()
    b = torch.randn(N, device=dev)
    C = torch.zeros(M, N, dtype=torch.float16, device=dev)
    nwg = (M // 256) * (N // 256)
    tr = torch.zeros(nwg * 4 * 12, dtype=torch.int64, device=dev)
    for _ in range(3):
        check(lib.vg_gemm_trace(50, ptr(X), ptr(W), ptr(b), ptr(C), ptr(tr), M, N, K, N, stream_ptr()))
    torch.cuda.synchronize()
    t = tr.view(-1, 12).cpu().double()
    npk = K // 64
    med = t.median(0).values.tolist()
    cal = med[6]
    print(f'{name}: entry->asm {med[0]:.0f}  asm prologue {med[1]:.0f}  K loop {med[2]:.0f} = {med[2] / npk:.0f} per K-tile (2048 = MFMA issue only)  '
          f'epilogue {med[7]:.0f}  whole asm {med[8]:.0f}', flush=True)
    print(f'    per K-tile: M wait {med[3] / (npk - 1) - cal:.0f}  barrier {med[4] / (npk - 1) - cal:.0f}  end wait {med[5] / (npk - 1) - cal:.0f}  '
          f'(stamp pair alone {cal:.0f}; each figure includes ~16-32 cycles of the MFMA between its stamps)', flush=True)
    first = t[:256 * 4].median(0).values.tolist(); rest = t[256 * 4:].median(0).values.tolist()
    print(f'    first round of workgroups: prologue {first[1]:.0f}  loop/tile {first[2] / npk:.0f}  epilogue {first[7]:.0f};  later rounds: prologue {rest[1]:.0f}  loop/tile {rest[2] / npk:.0f}  epilogue {rest[7]:.0f}', flush=True)
    for w in range(4):
        tw = t[t[:, 10] == w].median(0).values.tolist()
        print(f'    wave {w}: loop {tw[2] / npk:.0f}/tile  M wait {tw[3] / (npk - 1) - cal:.0f}  barrier {tw[4] / (npk - 1) - cal:.0f}  end {tw[5] / (npk - 1) - cal:.0f}', flush=True)
