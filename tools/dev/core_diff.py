"""Debug aid: GPU core distances vs the oracle on a small LiDAR-like scene; prints where they differ."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import hdbscan_oracle as ho
from test_cluster import lidar_scene, _gpu_model
cuda = torch.device('cuda:0')
X = lidar_scene(1, int(sys.argv[1]) if len(sys.argv) > 1 else 4000)
model = _gpu_model(cuda, len(X))
lo, hi, w2, core2 = model.mst(torch.from_numpy(X).to(cuda), want_core=True)
core2 = core2.cpu().numpy()
want = ho.core_distances_sq(X)
bad = np.flatnonzero(core2 != want)
print('n', len(X), 'differ', len(bad))
for i in bad[:12]:
    print(i, X[i], 'got', core2[i], 'want', want[i], 'ratio', core2[i] / want[i])
if len(bad):
    print('got<want', int((core2[bad] < want[bad]).sum()), 'got>want', int((core2[bad] > want[bad]).sum()), 'inf', int(np.isinf(core2[bad]).sum()))
