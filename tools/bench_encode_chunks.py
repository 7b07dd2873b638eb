"""ViT encode time of one frame's crops as one batch vs in sub-batches (activation working set vs the 256 MB MALL)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilgod_amd import clip_weights as cw
from vilgod_amd.clip_wrapper import VitEncoder
dev = torch.device('cuda:0')
enc = VitEncoder(cw.synthetic_vit_weights(0, **cw.VIT_B16), dtype='f16', device=dev)
n = int(os.environ.get('CROPS', '327'))
patches = (torch.randn(n * 196, 768, device=dev) * 0.5).half()
for chunk in (n, 218, 164, 128, 109, 82, 64):
    def run():
        outs = []
        for a in range(0, n, chunk):
            b = min(n, a + chunk)
            outs.append(enc.encode_patches(patches[a * 196:b * 196], b - a))
        return outs
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    print(f'{n} crops in chunks of {chunk:4d}: {e0.elapsed_time(e1) / 5:7.2f} ms per frame')
