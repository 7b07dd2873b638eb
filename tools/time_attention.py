"""vg_attention alone (product library): time per launch on ViT-B/16 shapes; VG_ATT_W4=0 selects the 7-wave kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilgod_amd._lib import lib, ptr, stream_ptr, check
dev = torch.device('cuda:0')
n, T, W, H = int(os.environ.get('CROPS', '331')), 197, 768, 12
ld = 3 * W + 64
qkv = (torch.randn(n * T, ld, device=dev) * 1.0).half()
out = torch.zeros(n * T, W, dtype=torch.float16, device=dev)
for _ in range(3): check(lib.vg_attention(ptr(qkv), ptr(out), n, T, W, H, ld, stream_ptr()))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): lib.vg_attention(ptr(qkv), ptr(out), n, T, W, H, ld, stream_ptr())
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1000
print(f'VG_ATT_W4={os.environ.get("VG_ATT_W4", "(default)")}: {us:.1f} us per launch, {n * H} items, {4.0 * n * H * T * T * 64 / us / 1e6:.0f} TFLOP/s (QK^T + PV)')
