"""fp16 tower vs the fp32 oracle on renderer-like crops: feature error and probability error (VG_VIT_RESID16=1 selects the
opt-in fp16 residual stream)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root
from oracle import vit_oracle as vo
from vilgod_amd import clip_weights as cw
from vilgod_amd.clip_wrapper import VitEncoder, clip_scores
n = int(os.environ.get('CROPS', '24'))
rng = np.random.default_rng(0)
x = torch.zeros(n, 3, 224, 224)
for i in range(n):
    for _ in range(30):
        cx, cy = rng.integers(40, 180, 2); r = rng.integers(3, 25)
        x[i, :, cy - r:cy + r, cx - r:cx + r] = float(rng.random())
x = (x - 0.45) / 0.27
wd = cw.synthetic_vit_weights(0, **cw.VIT_B16)
text = torch.as_tensor(cw.synthetic_text_features(0, 24, 512))
with torch.no_grad():
    ref = vo.vit_forward({k: torch.as_tensor(v) for k, v in wd.items()}, x, 12)
    pr = vo.clip_probabilities(ref, text)
enc = VitEncoder(wd, dtype='f16', device='cuda:0')
f = enc.encode(x.cuda().contiguous()).cpu()
p = clip_scores(f.cuda(), text.cuda())[0].cpu()
print('residual stream', 'fp16' if os.environ.get('VG_VIT_RESID16') else 'fp32', '| feature rel L2', float((f - ref).norm() / ref.norm()),
      '| max prob err', float((p - pr).abs().max()), '| top-1 agree', int((p.argmax(1) == pr.argmax(1)).sum()), '/', n)
