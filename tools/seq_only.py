"""process_sequence (the reference's default stage order) alone on a coherent synthetic sequence: for kernel traces.
    python tools/seq_only.py [frames=48] [workers=6]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilgod_amd import synthetic
from vilgod_amd.pipeline import PseudoLabelPipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 6
frames, poses = synthetic.make_sequence(seed=0, n_frames=n, n_points=150_000, n_objects=60)
pipe = PseudoLabelPipeline(device='cuda:0', max_points=151_024, clip_model_path='/nonexistent')
d = [pipe.upload(f) for f in frames]
pipe.process_sequence(d[:6], poses[:6], poses[0], n_workers=nw)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = pipe.process_sequence(d, poses, poses[0], n_workers=nw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'process_sequence {n} frames, {nw} workers: {1000 * dt / n:.2f} ms per frame = {n / dt:.1f} frames/s', flush=True)
