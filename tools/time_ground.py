"""Ground pass alone: ms per pass on the four synthetic benchmark frames and on the KITTI fixture scans (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vilgod_amd import synthetic
from vilgod_amd.pipeline import PseudoLabelPipeline
pipe = PseudoLabelPipeline(device='cuda:0', max_points=200_000, clip_model_path='/nonexistent', box_workers=0)
sets = {'synthetic 150k': [pipe.upload(synthetic.make_frame(1 + i, 150_000, n_objects=60)) for i in range(4)]}
kd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
ks = sorted(f for f in os.listdir(kd) if f.startswith('kitti_') and f.endswith('.bin'))
if ks:
    sets['kitti fixture scans'] = [pipe.upload(np.fromfile(os.path.join(kd, f), dtype=np.float32).reshape(-1, 4)) for f in ks]
for name, frames in sets.items():
    pipe.new_sequence()
    for f in frames: pipe.ground(f)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 0
    for rep in range(10):
        for f in frames:
            pipe.ground(f); n += 1
    torch.cuda.synchronize()
    print(f'{name}: {1000 * (time.perf_counter() - t0) / n:.3f} ms per ground pass ({frames[0].shape[0]} points)')
