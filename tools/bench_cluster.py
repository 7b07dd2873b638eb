"""Clustering stage breakdown on one 150k-pt synthetic frame: GPU MST vs D2H vs host tree vs packing."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilgod_amd import synthetic
from vilgod_amd.pipeline import PseudoLabelPipeline
from vilgod_amd.frame_state import pack_clusters
pipe = PseudoLabelPipeline(device='cuda:0', max_points=160_000, clip_model_path='/nonexistent')
pts = pipe.upload(synthetic.make_frame(1, 150_000))
mask = pipe.ground(pts)
ref = pipe.to_ref(pts, np.eye(4))
X = ref[mask == 0].contiguous()
n = X.shape[0]
def t(f, reps=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3, r
ms, (lo, hi, w2) = t(lambda: pipe.cluster_model.mst(X))
print(f'n={n} mst (GPU incl. per-round syncs): {ms:.2f} ms, rounds {pipe.cluster_model.n_rounds_}')
ms, arrs = t(lambda: (lo.cpu().numpy(), hi.cpu().numpy(), w2.cpu().numpy()))
print(f'D2H of sorted edges: {ms:.2f} ms')
ms, (labels, probs, nc) = t(lambda: pipe.cluster_model.tree(*arrs, n))
print(f'host tree: {ms:.2f} ms  ({nc} clusters)')
ms, _ = t(lambda: pack_clusters(labels, probs, 0.3))
print(f'pack_clusters (numpy): {ms:.2f} ms')
