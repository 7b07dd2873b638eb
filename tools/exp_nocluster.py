"""Experiment (not part of the product): how much of the step time is the clustering stage's GPU work?  The four distinct frames of
the benchmark are clustered once and their labels cached by point count afterwards."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vilgod_amd import synthetic
from vilgod_amd.pipeline import PseudoLabelPipeline
dev = torch.device('cuda:0')
pipe = PseudoLabelPipeline(device=dev, max_points=151_024, clip_model_path='/nonexistent')
frames = [pipe.upload(synthetic.make_frame(1 + i, 150_000, n_objects=60)) for i in range(4)]
poses = synthetic.make_poses(200)
def run(K):
    pipe.new_sequence()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pipe.process_frames([frames[i % 4] for i in range(K)], [poses[i + 1] for i in range(K)], poses[0], n_workers=6)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
run(12)
print('normal       ms/step', run(96), run(96))
cache = {}
orig = PseudoLabelPipeline.cluster
def cached(self, d_X):
    k = d_X.shape[0]
    if k not in cache: cache[k] = orig(self, d_X)
    return cache[k]
PseudoLabelPipeline.cluster = cached
run(12)
print('no clustering ms/step', run(96), run(96))
# variant: GPU part of the clustering kept (MST), host part (edge D2H, hierarchy, packing) served from the cache
def gpu_only(self, d_X):
    k = d_X.shape[0]
    self.cluster_model.mst(d_X)
    return cache[k]
PseudoLabelPipeline.cluster = gpu_only
run(12)
print('GPU MST only  ms/step', run(96), run(96))
PseudoLabelPipeline.cluster = orig
run(12)
print('normal again  ms/step', run(96), run(96))
