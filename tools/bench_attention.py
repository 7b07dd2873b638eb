"""k_attention_f16 alone: time per launch and the per-wave phase split (vg_attention_trace)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dev'))
from devlib import lib, ptr, stream_ptr, check        # the development build (tools/dev)
dev = torch.device('cuda:0')
n, T, W, H = int(os.environ.get('CROPS', '325')), 197, 768, 12
ld = 3 * W + 64
qkv = (torch.randn(n * T, ld, device=dev) * 1.0).half()
out = torch.zeros(n * T, W, dtype=torch.float16, device=dev)
tr = torch.zeros(256 * 7 * 8, dtype=torch.int64, device=dev)
for use_trace in (False, True):
    t = ptr(tr) if use_trace else None
    for _ in range(3): check(lib.vg_attention_trace(ptr(qkv), ptr(out), n, T, W, H, ld, t, stream_ptr()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): lib.vg_attention_trace(ptr(qkv), ptr(out), n, T, W, H, ld, t, stream_ptr())
    e1.record(); torch.cuda.synchronize()
    print(f'{"traced" if use_trace else "plain "} launch: {e0.elapsed_time(e1) / 20 * 1000:.1f} us for {n * H} items')
items_per_wg = n * H / 256
t = tr.view(256, 7, 8).cpu().double() / items_per_wg
names = ['stage+barrier', 'next loads', 'S^T mfma issue', 'max (mfma wait)', 'exp', 'P,V^T,O^T issue', 'output', 'end barrier']
for w in range(7):
    print(f'wave {w}: ' + '  '.join(f'{nm} {t[:, w, i].mean().item():6.0f}' for i, nm in enumerate(names)) + f'  | total {t[:, w].sum(1).mean().item():7.0f} cycles per item')
