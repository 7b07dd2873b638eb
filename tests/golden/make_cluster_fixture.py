"""Full-size clustering fixture: the CPU oracle (oracle/hdbscan_oracle.py, Prim in C) on the ~80k
non-ground points of ONE synthetic 150k-point frame -- the benchmark's size -- frozen as sha256 digests.

    python tests/golden/make_cluster_fixture.py            # build container, ~1-2 min; writes cluster_full_golden.json

The frame is regenerated from its seed on both sides (vilgod_amd/synthetic.py), the ground set comes from the C++
ground oracle (deterministic: -ffp-contract=off, same image), pose == ref pose so that points_ref == points exactly.
tests/test_cluster.py::test_hip_cluster_full_size_equals_oracle_fixture rebuilds X the same way on the GPU box, checks
the digest of X first (a host-dependent input would otherwise look like a kernel bug) and then the HIP result.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SEED, N_POINTS, N_OBJECTS = 7, 150_000, 60


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def build_input():
    """-> X float32 [M,3]: non-ground points of the fixture frame (oracle ground set, identity transform)."""
    from oracle import patchworkpp as opw
    from vilgod_amd import synthetic
    pts = synthetic.make_frame(SEED, N_POINTS, n_objects=N_OBJECTS)
    p = opw.Parameters()
    p.min_range = 1.5
    gidx = opw.mask_ground_points(pts, opw.patchworkpp(p), 1.723)
    gm = np.zeros(len(pts), bool)
    gm[gidx] = True
    return np.ascontiguousarray(pts[~gm][:, :3])


def digests(X, core2, lo, hi, w2, labels, probs):
    """Everything the GPU test compares, as digests.  lo/hi/w2: MST edges sorted by (w2, lo, hi)."""
    from oracle import hdbscan_oracle as ho
    dets = ho.detections_from_labels(labels, probs)
    return {
        'n': int(len(X)), 'x_sha256': sha(X), 'core2_sha256': sha(core2.astype(np.float64)),
        'mst_w2_sha256': sha(w2.astype(np.float64)), 'mst_edges_sha256': sha(np.stack([lo, hi], 1).astype(np.int64)),
        'labels_sha256': sha(labels.astype(np.int64)), 'canonical_labels_sha256': sha(ho.canonical(labels)),
        'probs_sha256': sha(probs.astype(np.float64)),
        'n_clusters': int(labels.max() + 1), 'n_noise': int((labels < 0).sum()), 'n_detections': len(dets),
        'detection_sizes_sha256': sha(np.array([len(d) for d in dets], np.int64)),
        'w2_sum': float(np.sum(w2)),
    }


def main():
    from oracle import hdbscan_oracle as ho
    t0 = time.time()
    X = build_input()
    n = len(X)
    core2 = ho.core_distances_sq(X)
    edges, ew2 = ho.mst_prim_c(X, core2)
    e, w2s = ho.sort_edges(edges, ew2)
    labels, probs = ho.tree_from_mst(e, w2s, n)
    out = digests(X, core2, e[:, 0], e[:, 1], w2s, labels, probs)
    out.update(seed=SEED, n_points=N_POINTS, n_objects=N_OBJECTS, seconds=round(time.time() - t0, 1),
               generator='tests/golden/make_cluster_fixture.py (oracle/hdbscan_oracle.py: cKDTree core distances, Prim in C, '
                         'Python tree stages)')
    path = os.path.join(ROOT, 'tests', 'golden', 'cluster_full_golden.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
