"""Integration parity (SURVEY §8c last row): this repo's entry point against what the REFERENCE's own stage harness wrote.

tests/golden/integration_{hot,default}.pkl.gz hold the sequence-state pickle and the two result pickles written by the
reference's `tools/preprocess_data.py::main` + `ZeroShotDetector.process()` (run unchanged in the build container under the
stand-ins of oracle/refharness.py, see tests/golden/make_integration.py).  The GPU tests run `tools/preprocess_data.py` of this
repo with the same overrides in parity mode (fp32 ViT, reference box mode) and require: identical key sets AND key order of
frames, detections and result dicts, identical dtypes, identical index sets / ids / flags / names, boxes within 1e-9 (state,
ref frame) and 1e-8 (results, ego frame), scores within 1e-3.
"""
import gzip
import os
import pickle
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

KEY = 'clip_a_point_representation_of_a'
PARITY = ['device.vit_dtype=f32', 'device.box_mode=reference', 'device.max_points=24000']


def _golden(which):
    with gzip.open(os.path.join(GOLDEN, f'integration_{which}.pkl.gz'), 'rb') as f:
        return pickle.load(f)


def _kind(x):
    """dtype / container signature of a pickled value."""
    if isinstance(x, np.ndarray):
        return f'ndarray[{x.dtype.kind}{x.dtype.itemsize if x.dtype.kind != "U" else ""}]'
    if isinstance(x, np.generic):
        return f'np.{type(x).__name__}'
    return type(x).__name__


@pytest.mark.parametrize('which', ['hot', 'default'])
def test_integration_golden_layout(which):
    """CPU: the frozen reference output has the two pickle families' layout of SURVEY §8b (guards the fixture itself)."""
    g = _golden(which)
    n = len(g['state'])
    assert n == len(g['results']) == len(g['indices']) == len(g['input_sha256'])
    base = ['_detections', '_ground_point_indices'] + (['_entropy_scores', '_entropy_indices'] if which == 'default' else []) \
        + ['_gt_cluster_mapping']
    for st, res in zip(g['state'], g['results']):
        assert list(st) == base                                                     # lidar_frame.py:42-44 order
        assert list(res) == ['boxes_lidar', 'name', 'score', 'moving']              # zero_shot_detector.py:852-857
        assert res['boxes_lidar'].dtype == np.float64 and res['boxes_lidar'].shape[1] == 7
        for d in st['_detections']:
            assert list(d)[:1] == ['cluster_id'] and 'cluster_points_index' in d
            if d['valid'] and KEY in d.get('object_class', {}):
                assert len(d['object_class_predictions'][KEY]) == 4
    assert sum(len(r['name']) for r in g['results']) > 0


def _compare(g, res, idx, state):
    """-> list of human-readable mismatches (empty = parity)."""
    bad = []

    def chk(cond, msg):
        if not cond:
            bad.append(msg)

    chk(idx == g['indices'], f'indices {idx} != {g["indices"]}')
    chk(len(state) == len(g['state']) and len(res) == len(g['results']), 'frame count')
    for f, (a, b) in enumerate(zip(state, g['state'])):
        chk(list(a) == list(b), f'frame {f}: keys {list(a)} != {list(b)}')
        for k in b:
            if k not in a:
                continue
            if k != '_detections':
                chk(_kind(a[k]) == _kind(b[k]), f'frame {f} {k}: type {_kind(a[k])} != {_kind(b[k])}')
        if '_ground_point_indices' in a and '_ground_point_indices' in b:
            # the index set (order is the native module's patch order upstream; consumers use it as a set, lidar_frame.py:82-87)
            chk(np.array_equal(np.sort(a['_ground_point_indices']), np.sort(b['_ground_point_indices'])), f'frame {f}: ground set')
        if '_entropy_indices' in b and '_entropy_indices' in a:
            chk(np.array_equal(a['_entropy_indices'], b['_entropy_indices']), f'frame {f}: entropy indices')
            if len(a['_entropy_scores']) == len(b['_entropy_scores']):
                chk(np.abs(a['_entropy_scores'] - b['_entropy_scores']).max(initial=0) <= 1e-9, f'frame {f}: entropy scores')
        chk(a.get('_gt_cluster_mapping') == b.get('_gt_cluster_mapping'), f'frame {f}: gt mapping')
        da, db = a.get('_detections', []), b.get('_detections', [])
        chk(len(da) == len(db), f'frame {f}: {len(da)} detections != {len(db)}')
        for c, (d, e) in enumerate(zip(da, db)):
            w = f'frame {f} det {c}'
            chk(list(d) == list(e), f'{w}: keys {list(d)} != {list(e)}')
            for k in e:
                if k not in d:
                    continue
                x, y = d[k], e[k]
                if isinstance(y, dict):
                    chk(isinstance(x, dict) and list(x) == list(y), f'{w} {k}: dict keys {list(x) if isinstance(x, dict) else x} != {list(y)}')
                    if not isinstance(x, dict):
                        continue
                    for kk in y:
                        if kk in x:
                            chk(_kind(x[kk]) == _kind(y[kk]), f'{w} {k}[{kk}]: type {_kind(x[kk])} != {_kind(y[kk])}')
                            if k in ('object_class_predictions', 'object_class_predictions_detailed', 'object_class'):
                                chk(np.array_equal(x[kk], y[kk]), f'{w} {k}: {x[kk]} != {y[kk]}')
                            else:
                                chk(np.shape(x[kk]) == np.shape(y[kk]) and np.abs(np.asarray(x[kk], np.float64) - np.asarray(y[kk], np.float64)).max() <= 1e-3,
                                    f'{w} {k}: {x[kk]} != {y[kk]}')
                else:
                    chk(_kind(x) == _kind(y), f'{w} {k}: type {_kind(x)} != {_kind(y)}')
                    if k == '_bounding_box':
                        chk(np.shape(x) == np.shape(y) and np.abs(np.asarray(x) - np.asarray(y)).max() <= 1e-9, f'{w} box {x} != {y}')
                    elif k == 'cluster_points_index':
                        chk(np.array_equal(x, y), f'{w}: cluster_points_index ({len(x)} vs {len(y)} points)')
                    else:
                        chk(x == y, f'{w} {k}: {x!r} != {y!r}')
    for f, (a, b) in enumerate(zip(res, g['results'])):
        chk(list(a) == list(b), f'result {f}: keys {list(a)}')
        for k in b:
            if k in a:
                chk(_kind(a[k]) == _kind(b[k]) or len(b[k]) == 0, f'result {f} {k}: type {_kind(a[k])} != {_kind(b[k])}')
        if len(a['name']) != len(b['name']):
            chk(False, f'result {f}: {len(a["name"])} labels != {len(b["name"])}')
            continue
        chk(np.array_equal(a['name'], b['name']), f'result {f}: names {a["name"]} != {b["name"]}')
        chk(np.array_equal(a['moving'], b['moving']), f'result {f}: moving')
        chk(a['boxes_lidar'].shape == b['boxes_lidar'].shape and np.abs(a['boxes_lidar'] - b['boxes_lidar']).max(initial=0) <= 1e-8,
            f'result {f}: boxes differ by {np.abs(a["boxes_lidar"] - b["boxes_lidar"]).max(initial=0) if a["boxes_lidar"].shape == b["boxes_lidar"].shape else "shape"}')
        chk(np.abs(np.asarray(a['score'], np.float64) - np.asarray(b['score'], np.float64)).max(initial=0) <= 1e-3, f'result {f}: scores')
    return bad


def _report(bad):
    """Mismatches grouped by kind (frame / detection numbers replaced), with a count and the first instance of each."""
    import re
    groups = {}
    for m in bad:
        groups.setdefault(re.sub(r'\d+', '#', m.split(':')[0]) + ':' + re.sub(r'[-\d.e+]+', '#', ':'.join(m.split(':')[1:]))[:80], []).append(m)
    lines = [f'{len(bad)} mismatches against the reference harness, {len(groups)} kinds:']
    for k, v in groups.items():
        lines.append(f'  x{len(v)}  {v[0][:400]}')
    return '\n'.join(lines)


def _run_cli(g, root, extra=()):
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import preprocess_data
    from vilgod_amd import config as vconfig
    import hashlib
    ov = list(g['overrides']) + [f'dataset.DATA_PATH={root}'] + PARITY + list(extra)
    # same inputs as the generator saw (the synthetic dataset is seeded; guard against generator drift)
    cfg = vconfig.load(os.path.join(ROOT, 'tools', 'configs'), 'preprocessing', ov)
    ds = vconfig.instantiate(cfg.dataset_class, logger=None, training=True, start_sequence=0, end_sequence=0)
    next(iter(ds.next_sequence()))
    got = [hashlib.sha256(np.ascontiguousarray(ds.get_lidar_points(f)).tobytes()).hexdigest() for f in range(ds.sequence_length)]
    assert got == g['input_sha256'], 'the seeded synthetic inputs differ from the ones the golden was made from'
    preprocess_data.main(ov)
    stages = '_'.join(g['stages'])
    seq = g['sequence']
    with open(f'{root}/preprocessed_data/results/vilgod_mi355x/{stages}/{seq}.pkl', 'rb') as f:
        res = pickle.load(f)
    with open(f'{root}/preprocessed_data/results/vilgod_mi355x/{stages}/{seq}_indices.pkl', 'rb') as f:
        idx = pickle.load(f)
    with open(f'{root}/preprocessed_data/vilgod_mi355x_seq/{seq}.pkl', 'rb') as f:
        state = pickle.load(f)
    return res, idx, state


@pytest.mark.gpu
@pytest.mark.parametrize('which', ['hot', 'default'])
def test_cli_reproduces_reference_harness_pickles(cuda, tmp_path, which):
    g = _golden(which)
    res, idx, state = _run_cli(g, str(tmp_path / which))
    bad = _compare(g, res, idx, state)
    assert not bad, _report(bad)


@pytest.mark.gpu
def test_cli_reproduces_reference_harness_pickles_stage_after_stage(cuda, tmp_path):
    """The reference's execution order (one stage over all frames, state pickle rewritten after every stage) gives the same files."""
    g = _golden('default')
    res, idx, state = _run_cli(g, str(tmp_path / 'staged'), extra=['device.fuse_stages=False', 'device.sync_every_stage=True',
                                                                   'device.frames_in_flight=1'])
    bad = _compare(g, res, idx, state)
    assert not bad, _report(bad)
