"""Regression pin of the ground oracle (oracle/patchworkpp_oracle.cpp) on the SIX KITTI scans the reference ships as demo data
(third_party/patchwork-plusplus/data/00000[0-5].bin, copied to tests/golden/kitti_00000[0-5].bin as data fixtures), run
SEQUENTIALLY on one stateful object like the reference's demo and ViLGOD's mask_ground_points (zero_shot_detector.py:137-146).

    python tests/golden/make_ground_pin.py        # writes ground_kitti.json

NOT reference truth: Patchwork++ cannot be built here (Eigen3 absent) and the reference holds no expected outputs.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import patchworkpp as opw  # noqa: E402

G = os.path.join(ROOT, 'tests', 'golden')
params = opw.Parameters()
params.min_range = 1.5
pp = opw.patchworkpp(params)
frames = []
for i in range(6):
    pts = np.fromfile(f'{G}/kitti_00000{i}.bin', dtype=np.float32).reshape(-1, 4)
    m = pp.estimateGround(pts)
    st = pp.state()
    frames.append({'n': int(len(pts)), 'n_ground': int(m.sum()), 'mask_sha256': hashlib.sha256(m.tobytes()).hexdigest(),
                   'sensor_height': float(st['sensor_height']), 'elevation_thr': [float(x) for x in st['elevation_thr']],
                   'flatness_thr': [float(x) for x in st['flatness_thr']]})
json.dump({'note': 'regression pin of oracle/patchworkpp_oracle.cpp on the six KITTI scans, one stateful object, in order (NOT reference '
                   'truth: the reference cannot be built here)', 'frames': frames}, open(f'{G}/ground_kitti.json', 'w'), indent=1)
print(json.dumps([(f['n'], f['n_ground']) for f in frames]))
