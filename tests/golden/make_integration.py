"""Integration golden (SURVEY §8c last row): the reference's OWN stage harness run end to end on a seeded sequence.

    python tests/golden/make_integration.py [hot] [default]

Runs, unchanged, `/root/reference/tools/preprocess_data.py::main` (the sequence loop, :73-103) which builds the reference's
`ZeroShotDetector` and calls `process()` (zero_shot_detector.py:58-69) over
  hot      mask_ground_points, spatial_clustering (n_frames=1), filter_detections, classification,
           fit_bounding_boxes_simple, evaluate_sequence                      (the hot-path stage list, 3 frames x 20k points)
  default  the shipped 9-stage list of preprocessing.yaml:50 (entropy scores, two-frame clustering, tracking, track boxes,
           label propagation)                                               (6 frames x 20k points)
with the absent third-party packages replaced as listed in oracle/refharness.py, on the repo's seeded synthetic dataset and
config tree (same keys as the reference's), and freezes what the reference WROTE: the sequence-state pickle
(zero_shot_detector.py:105-114) and the two result pickles (preprocess_data.py:98-103).  tests/test_integration.py runs this
repo's CLI with the same overrides on the GPU and compares key sets, key order, dtypes, index sets, names, boxes and scores.

Only in the build container (needs /root/reference).  Output: tests/golden/integration_<which>.pkl.gz (data only).
"""
import gzip
import hashlib
import logging
import os
import pickle
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))

HOT_STAGES = ['mask_ground_points', 'spatial_clustering', 'filter_detections', 'classification', 'fit_bounding_boxes_simple',
              'evaluate_sequence']

# the overrides BOTH sides run with (the test appends device.* knobs that have no counterpart in the reference)
COMMON = ['preprocessor=waymo', 'dataset.SYNTHETIC.points_per_frame=20000', 'dataset.SYNTHETIC.objects_per_frame=10',
          'dataset.SYNTHETIC.n_sequences=1', 'end_sequence=0', 'paths.clip_model=/nonexistent']
CASES = {
    'hot': COMMON + ['dataset.SYNTHETIC.frames_per_sequence=3', 'dataset.SYNTHETIC.seed=3',
                     'pipeline_active=[' + ','.join(HOT_STAGES) + ']', 'pipeline.2.args.n_frames=1'],
    # the reference indexes n_neighbouring_frames frames unconditionally (:168-171): the window must fit the 6-frame sequence
    'default': COMMON + ['dataset.SYNTHETIC.frames_per_sequence=6', 'pipeline.1.args.n_neighbouring_frames=5'],
}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_case(which):
    from vilgod_amd import config as vconfig
    from oracle import refharness
    tmp = tempfile.mkdtemp(prefix=f'vg_integration_{which}_')
    overrides = CASES[which] + [f'dataset.DATA_PATH={tmp}/data', 'preprocessor.clustering.model._target_=hdbscan.HDBSCAN']
    cfg = vconfig.load(os.path.join(ROOT, 'tools', 'configs'), 'preprocessing', overrides)
    dev = cfg.get('device', {})
    ref_main = refharness.install(cfg, plane_seed=dev.get('plane_seed', 666), subsample_seed=dev.get('subsample_seed', 0))
    logging.basicConfig(level=logging.INFO, stream=sys.stdout)
    os.makedirs(f'{tmp}/tools', exist_ok=True)
    cwd = os.getcwd()
    os.chdir(f'{tmp}/tools')                      # classification creates ../output_images/<seq> relative to the cwd (:331-332)
    t0 = time.time()
    try:
        ref_main.main(cfg)
    finally:
        os.chdir(cwd)
    print(f'[{which}] reference harness: {time.time() - t0:.1f} s')
    seq = 'synthetic_train_0000'
    stages = '_'.join(cfg.pipeline_active)
    with open(f'{tmp}/data/preprocessed_data/vilgod_mi355x_seq/{seq}.pkl', 'rb') as f:
        state = pickle.load(f)
    with open(f'{tmp}/data/preprocessed_data/results/vilgod_mi355x/{stages}/{seq}.pkl', 'rb') as f:
        results = pickle.load(f)
    with open(f'{tmp}/data/preprocessed_data/results/vilgod_mi355x/{stages}/{seq}_indices.pkl', 'rb') as f:
        indices = pickle.load(f)
    # the inputs, by digest (both sides regenerate them from the seed)
    ds = vconfig.instantiate(cfg.dataset_class, logger=None, training=True, start_sequence=0, end_sequence=0)
    next(iter(ds.next_sequence()))
    inputs = [sha(ds.get_lidar_points(f)) for f in range(ds.sequence_length)]
    out = dict(overrides=CASES[which], stages=list(cfg.pipeline_active), sequence=seq, state=state, results=results,
               indices=indices, input_sha256=inputs, numpy=np.__version__)
    path = os.path.join(OUT, f'integration_{which}.pkl.gz')
    with gzip.open(path, 'wb', compresslevel=9) as f:
        pickle.dump(out, f, protocol=4)
    nd = [len(s.get('_detections', [])) for s in state]
    nv = [sum(d['valid'] for d in s.get('_detections', [])) for s in state]
    print(f'[{which}] {path}: {os.path.getsize(path) / 1e6:.2f} MB; detections per frame {nd}, valid {nv}, '
          f'labelled {[len(r["name"]) for r in results]}, names {[list(r["name"]) for r in results]}')
    print(f'[{which}] frame keys {[list(s) for s in state][:1]}; detection keys {list(state[0]["_detections"][0]) if nd[0] else None}')


if __name__ == '__main__':
    for w in (sys.argv[1:] or ['hot', 'default']):
        run_case(w)
