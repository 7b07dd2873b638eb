"""Hierarchy stage of HDBSCAN on one 150k-point synthetic frame: the host stage (copy of the tree to the host + csrc/hdbscan_tree.cpp) against
the device stage (csrc/hdbscan_device.hip + copy of labels and probabilities to the host), same tree, results compared bit for bit.

    python tools/bench_hierarchy.py [--points 150000] [--frames 4] [--reps 20]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilgod_amd import synthetic                                   # noqa: E402
from vilgod_amd.hdbscan import HDBSCAN, DeviceHierarchy             # noqa: E402
from vilgod_amd.pipeline import PseudoLabelPipeline                 # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--points', type=int, default=150_000)
    ap.add_argument('--frames', type=int, default=4)
    ap.add_argument('--reps', type=int, default=20)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    pipe = PseudoLabelPipeline(device=dev, max_points=args.points + 10_000, clip_model_path='/nonexistent')
    model = pipe.cluster_model
    hier = DeviceHierarchy(max_points=args.points + 10_000, device=dev)
    mcs, eps = model.min_cluster_size, model.cluster_selection_epsilon
    for f in range(args.frames):
        pts = pipe.upload(synthetic.make_frame(1 + f, args.points))
        mask = pipe.ground(pts)
        X = pipe.to_ref(pts, np.eye(4))[mask == 0].contiguous()
        n = X.shape[0]
        lo, hi, w2 = model.mst(X)
        torch.cuda.synchronize()
        host_ms, dev_ms, dev_kernel_ms = [], [], []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            h = lo.cpu().numpy(), hi.cpu().numpy(), w2.cpu().numpy()
            L0, P0, c0 = model.tree(*h, n)
            host_ms.append((time.perf_counter() - t0) * 1e3)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            dl, dp, dn = hier.tree_async(lo, hi, w2, n, mcs, eps)
            e1.record()
            L1, P1 = dl.cpu().numpy(), dp.cpu().numpy()
            dev_ms.append((time.perf_counter() - t0) * 1e3)
            dev_kernel_ms.append(e0.elapsed_time(e1))
        extra = ''
        emul = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'emul', '_build', 'libhd_emul.so')
        if os.path.exists(emul):          # (built by tests/test_hierarchy.py: the CPU emulation of the kernels' bodies also reports the split count)
            import ctypes
            el = ctypes.CDLL(emul)
            p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
            nc_, ns_, sw_ = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
            el.hd_emul_tree(p(h[0]), p(h[1]), p(h[2]), n, mcs, ctypes.c_double(eps), p(np.empty(n, np.int32)), p(np.empty(n, np.float64)),
                            ctypes.byref(nc_), ctypes.byref(ns_), ctypes.byref(sw_))
            extra = f'  splits {ns_.value}  sweeps over the cluster tree {sw_.value}'
        same = int(dn.item()) == c0 and np.array_equal(L0, L1) and np.array_equal(P0.view(np.uint64), P1.view(np.uint64))
        med = lambda a: float(np.median(a))
        print(f'frame {f}: n {n}  clusters {c0}  host stage (D2H of the tree + hdbscan_tree.cpp) {med(host_ms):.2f} ms   device stage + D2H of labels '
              f'{med(dev_ms):.2f} ms (kernels {med(dev_kernel_ms):.2f} ms)   results {"bit-identical" if same else "DIFFERENT"}{extra}', flush=True)


if __name__ == '__main__':
    main()
