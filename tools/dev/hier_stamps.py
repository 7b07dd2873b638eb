"""Phase times of the hierarchy stage's one-workgroup kernels (development library): python tools/dev/hier_stamps.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from devlib import lib, ptr, stream_ptr, check   # noqa: E402
os.environ['VILGOD_HIP_LIB'] = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'vilgod_amd', 'libvilgod_hip_dev.so')
from vilgod_amd import synthetic                                   # noqa: E402
from vilgod_amd.pipeline import PseudoLabelPipeline                 # noqa: E402

dev = torch.device('cuda:0')
pipe = PseudoLabelPipeline(device=dev, max_points=160_000, clip_model_path='/nonexistent')
pts = pipe.upload(synthetic.make_frame(1, 150_000))
mask = pipe.ground(pts)
X = pipe.to_ref(pts, np.eye(4))[mask == 0].contiguous()
n = X.shape[0]
lo, hi, w2 = pipe.cluster_model.mst(X)
h = ctypes.c_void_p()
check(lib.vg_hier_create(ctypes.byref(h), 160_000))
labels = torch.empty(n, dtype=torch.int32, device=dev); probs = torch.empty(n, dtype=torch.float64, device=dev); nc = torch.zeros(1, dtype=torch.int32, device=dev)
names = {0: 'tree_a start', 1: 'nodes staged', 2: 'Kruskal done', 4: 'tree_bc start', 5: 'staged', 6: 'up sweeps', 7: 'down sweeps', 8: 'BFS numbering', 9: 'selection',
         10: 'epsilon', 11: 'scan', 12: 'owners', 13: 'end'}
for rep in range(4):
    check(lib.vg_hdbscan_tree_device(h, ptr(lo), ptr(hi), ptr(w2), n, 15, 0.15, ptr(labels), ptr(probs), ptr(nc), stream_ptr()))
    torch.cuda.synchronize()
    out = np.zeros(16, np.int64)
    check(lib.vg_hier_stamps(h, out.ctypes.data_as(ctypes.c_void_p)))
    keys = sorted(names)
    print(f'rep {rep}: ' + '  '.join(f'{names[b]} +{(out[b] - out[a]) / 100.0:.1f} us' for a, b in zip(keys[:-1], keys[1:]) if b not in (4,)) + f'   shader clock during k_hd_tree_bc {(out[15] - out[14]) / max(1, out[13] - out[4]) * 100.0:.0f} MHz', flush=True)
