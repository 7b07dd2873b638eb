"""Stress of the device hierarchy stage (races only show on the GPU: the CPU emulation runs the phases' threads one after the other):
the same real trees again and again, random trees of many shapes, every result compared with the host stage bit for bit.
    python tools/dev/hier_stress.py [repeats=150] [random_cases=120]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from vilgod_amd import synthetic                                   # noqa: E402
from vilgod_amd.hdbscan import HDBSCAN, DeviceHierarchy             # noqa: E402
from vilgod_amd.pipeline import PseudoLabelPipeline                 # noqa: E402
from test_hierarchy import random_tree, host_tree                   # noqa: E402

repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 150
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 120
dev = torch.device('cuda:0')
pipe = PseudoLabelPipeline(device=dev, max_points=210_000, clip_model_path='/nonexistent', hierarchy='host')
hier = DeviceHierarchy(max_points=210_000, device=dev)
bad = 0
t0 = time.time()
for f, npts in enumerate((150_000, 200_000, 60_000)):
    pts = pipe.upload(synthetic.make_frame(11 + f, npts, n_objects=60 + 30 * f))
    mask = pipe.ground(pts)
    X = pipe.to_ref(pts, np.eye(4))[mask == 0].contiguous()
    n = X.shape[0]
    lo, hi, w2 = pipe.cluster_model.mst(X)
    L0, P0, c0 = pipe.cluster_model.tree(lo.cpu().numpy(), hi.cpu().numpy(), w2.cpu().numpy(), n)
    for rep in range(repeats):
        dl, dp, dn = hier.tree_async(lo, hi, w2, n, 15, 0.15)
        ok = int(dn.item()) == c0 and np.array_equal(L0, dl.cpu().numpy()) and np.array_equal(P0.view(np.uint64), dp.cpu().numpy().view(np.uint64))
        bad += not ok
    print(f'real tree {f}: n {n}, clusters {c0}, {repeats} repetitions, differing so far {bad}', flush=True)
rng = np.random.default_rng(2026)
for i in range(cases):
    kind = int(rng.integers(0, 5))
    n = int(rng.choice([50, 700, 5000, 20_000, 60_000, 120_000]))
    if kind == 1:
        n = min(n, 2500)                                             # (the blob generator builds a dense distance matrix)
    mcs = int(rng.choice([2, 3, 5, 15, 15, 15, 32]))
    eps = float(rng.choice([0.0, 0.15, 0.5]))
    lo, hi, w2 = random_tree(rng, n, kind)
    L0, P0, c0 = host_tree(lo, hi, w2, n, mcs, eps)
    for rep in range(3):
        dl, dp, dn = hier.tree_async(torch.from_numpy(lo).to(dev), torch.from_numpy(hi).to(dev), torch.from_numpy(w2).to(dev), n, mcs, eps)
        ok = int(dn.item()) == c0 and np.array_equal(L0, dl.cpu().numpy()) and np.array_equal(P0.view(np.uint64), dp.cpu().numpy().view(np.uint64))
        if not ok:
            bad += 1
            print(f'DIFFERENT: case {i} kind {kind} n {n} mcs {mcs} eps {eps} rep {rep}', flush=True)
print(f'{cases} random trees x 3, {3 * repeats} repetitions of real trees: {bad} differing results, {time.time() - t0:.0f} s')
sys.exit(1 if bad else 0)
