#!/usr/bin/env python3
"""Write a seeded data set in OpenPCDet's Waymo / Argoverse 2 on-disk layout (vilgod_amd/fixture_data.py), to try the real-data
adapters without the real data:

    python tools/make_fixture_dataset.py waymo /tmp/wd --sequences 2 --frames 20 --points 150000 --objects 60
    python tools/preprocess_data.py preprocessor=waymo dataset=waymo_openpcdet dataset.DATA_PATH=/tmp/wd
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vilgod_amd import fixture_data as fx      # noqa: E402

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('kind', choices=['waymo', 'argoverse'])
    ap.add_argument('root')
    ap.add_argument('--sequences', type=int, default=2)
    ap.add_argument('--frames', type=int, default=20)
    ap.add_argument('--points', type=int, default=150000)
    ap.add_argument('--objects', type=int, default=60)
    ap.add_argument('--seed', type=int, default=0)
    a = ap.parse_args()
    w = fx.write_waymo if a.kind == 'waymo' else fx.write_argo2
    print('\n'.join(w(a.root, n_sequences=a.sequences, n_frames=a.frames, n_points=a.points, n_objects=a.objects, seed=a.seed)))
