"""k_render alone on the clusters of one synthetic 150k-point frame (the product path: single-channel fp16 patch rows), us per launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vilgod_amd import synthetic
from vilgod_amd.pipeline import PseudoLabelPipeline
from vilgod_amd.frame_state import pack_clusters
dev = torch.device('cuda:0')
pipe = PseudoLabelPipeline(device=dev, max_points=151_024, clip_model_path='/nonexistent')
poses = synthetic.make_poses(3)
for seed in (300, 301):
    pts = pipe.upload(synthetic.make_frame(seed, 150_000, n_objects=60))
    fs, d_ref, d_X, gidx = pipe.prepare(pts, poses[1], poses[0], fnr=0)
    labels, pr = pipe.cluster(d_X)
    ids, index, seg = pack_clusters(labels, pr, pipe.prob_threshold)
    d_index, d_seg = torch.from_numpy(index).to(dev), torch.from_numpy(seg).to(dev)
    valid, _ = pipe.filter(d_X, d_index, d_seg, pipe.ground_plane(d_ref, gidx))
    vrows = np.flatnonzero(valid.cpu().numpy())
    parts = [index[seg[c]:seg[c + 1]] for c in vrows]
    d_vi = torch.from_numpy(np.concatenate(parts)).to(dev)
    d_vs = torch.from_numpy(np.r_[0, np.cumsum([len(p) for p in parts])].astype(np.int32)).to(dev)
    for _ in range(3):
        out = pipe.projection.render_frame(d_X, d_vi, d_vs, fs.transform_to_ego, out='patch16c1')
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = pipe.projection.render_frame(d_X, d_vi, d_vs, fs.transform_to_ego, out='patch16c1')
    e1.record(); torch.cuda.synchronize()
    print(f'frame seed {seed}: {len(vrows)} clusters x 4 views, render_frame (gather, medians, origin transform, k_render) {e0.elapsed_time(e1) / 20 * 1000:.1f} us per call; '
          f'checksum {out.float().sum().item():.6e}', flush=True)
