"""Where the two-frame (5-D) clustering of the default configuration spends its time: input rows, MST (GPU), tree (host),
label transfer; and the same for the 3-D single-frame clustering of the same frame."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilgod_amd import synthetic
from vilgod_amd.pipeline import PseudoLabelPipeline, default_preprocessor_cfg
from vilgod_amd.entropy import EntropyScorer, TwoFrameClusterer, full_scores

n_frames = int(os.environ.get('FRAMES', '8'))
frames, poses = synthetic.make_sequence(seed=0, n_frames=n_frames, n_points=150000, n_objects=60)
pipe = PseudoLabelPipeline(default_preprocessor_cfg(), device='cuda:0', vit_dtype='f16', clip_model_path='/nonexistent')
d_frames = [pipe.upload(f) for f in frames]
prepared = [pipe.prepare(f, poses[i], poses[0], fnr=i) for i, f in enumerate(d_frames)]
X = [p[2] for p in prepared]
scorer = EntropyScorer(pipe.cluster_model)
H = scorer.score_sequence(X)
ent = []
for (fs, _, d_X, _), h in zip(prepared, H):
    s, i = scorer.reduce(h)
    ent.append(full_scores(d_X.shape[0], s, i, device=pipe.device))
two = TwoFrameClusterer(pipe.cluster_model)
m = pipe.cluster_model
sync = torch.cuda.synchronize
acc = {}
def tick(k, t0):
    sync(); t = time.perf_counter(); acc[k] = acc.get(k, 0.0) + (t - t0); return t
for rep in range(2):
    acc.clear()
    two.reset()
    for f in range(n_frames):
        sync(); t = time.perf_counter()
        seq = two.cluster_input(f, X, ent); t = tick('input rows', t)
        lo, hi, w2 = m.mst(seq, dim=5); t = tick('mst 5-D', t)
        a, b, c = lo.cpu().numpy(), hi.cpu().numpy(), w2.cpu().numpy(); t = tick('copy', t)
        lab, prob, _ = m.tree(a, b, c, seq.shape[0]); t = tick('tree (host)', t)
        m.grid(seq); idx, _ = m.nearest(X[f], two.gate); idx = idx.cpu(); t = tick('transfer', t)
        lo, hi, w2 = m.mst(X[f][:, :3].contiguous(), dim=3); t = tick('mst 3-D (same frame)', t)
        a, b, c = lo.cpu().numpy(), hi.cpu().numpy(), w2.cpu().numpy()
        sync(); t = time.perf_counter()
        m.tree(a, b, c, X[f].shape[0]); t = tick('tree 3-D (host)', t)
    print(f'rep {rep}: rows/frame 5-D {seq.shape[0]}, 3-D {X[-1].shape[0]} | ms per frame: ' +
          '  '.join(f'{k} {1000 * v / n_frames:.2f}' for k, v in acc.items()))
