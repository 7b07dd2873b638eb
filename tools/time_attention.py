"""vg_attention alone (product library), ViT-B/16 shapes, time per launch, variants INTERLEAVED in one process (boxes differ by +-5 %):
stagger on / off (VG_ATT_STAGGER, k_attention_f16 STAG) and the transposed V image (VG_ATT_TR=0); the switches are read per call by
this handle-less entry point.   CROPS=337 python tools/time_attention.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilgod_amd._lib import lib, ptr, stream_ptr, check
dev = torch.device('cuda:0')
n, T, W, H = int(os.environ.get('CROPS', '337')), 197, 768, 12
ld = 3 * W + 64
qkv = (torch.randn(n * T, ld, device=dev) * 1.0).half()
out = torch.zeros(n * T, W, dtype=torch.float16, device=dev)
variants = [('stagger', {'VG_ATT_STAGGER': '1', 'VG_ATT_TR': '1'}), ('no stagger', {'VG_ATT_STAGGER': '0', 'VG_ATT_TR': '1'}),
            ('transposed V image', {'VG_ATT_STAGGER': '0', 'VG_ATT_TR': '0'})]
tot = {name: [] for name, _ in variants}
sums = {}
for rep in range(6):
    for name, env in variants:
        os.environ.update(env)
        for _ in range(2): check(lib.vg_attention(ptr(qkv), ptr(out), n, T, W, H, ld, stream_ptr()))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): lib.vg_attention(ptr(qkv), ptr(out), n, T, W, H, ld, stream_ptr())
        e1.record(); torch.cuda.synchronize()
        tot[name].append(e0.elapsed_time(e1) / 20 * 1000)
        sums[name] = out.float().sum().item()
for name, _ in variants:
    v = sorted(tot[name])
    print(f'{name:20s}: median {v[len(v) // 2]:.1f} us per launch (min {v[0]:.1f}, max {v[-1]:.1f}; {n * H} items), output checksum {sums[name]:.6e}')
