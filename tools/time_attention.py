"""vg_attention alone (product library), ViT-B/16 shapes, time per launch.  A/B: run once with VG_ATT_TR=0 (the transposed V image) and
once without in the SAME gpurun call -- boxes differ by +-5 %, so only interleaved or same-box numbers compare."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilgod_amd._lib import lib, ptr, stream_ptr, check
dev = torch.device('cuda:0')
n, T, W, H = int(os.environ.get('CROPS', '331')), 197, 768, 12
ld = 3 * W + 64
qkv = (torch.randn(n * T, ld, device=dev) * 1.0).half()
out = torch.zeros(n * T, W, dtype=torch.float16, device=dev)
tot = 0.0
for rep in range(6):
    for _ in range(2): check(lib.vg_attention(ptr(qkv), ptr(out), n, T, W, H, ld, stream_ptr()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): lib.vg_attention(ptr(qkv), ptr(out), n, T, W, H, ld, stream_ptr())
    e1.record(); torch.cuda.synchronize()
    tot += e0.elapsed_time(e1) / 20 * 1000
print(f'VG_ATT_TR={os.environ.get("VG_ATT_TR", "(default 1)")}: {tot / 6:.1f} us per launch ({n * H} items), output checksum {out.float().sum().item():.6e}')
