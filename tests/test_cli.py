"""Drop-in boundary on the GPU: the `preprocessor=` CLI, the stage dispatcher with the reference's stage names,
stage-granular resume, the two pickle families, and the frame-sharded N=2 path (two processes on
one GPU, gloo) producing the same pickles as N=1."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

OVR = ['dataset.SYNTHETIC.frames_per_sequence=6',          # tracks shorter than 5 frames are dropped by propagate_labels
       'dataset.SYNTHETIC.points_per_frame=20000',
       'dataset.SYNTHETIC.objects_per_frame=10', 'dataset.SYNTHETIC.n_sequences=1', 'end_sequence=0',
       'device.max_points=24000', 'paths.clip_model=/nonexistent']


DEFAULT_STAGES = ['mask_ground_points', 'calculate_entropy_scores', 'spatial_clustering', 'filter_detections', 'track_clusters',
                  'classification', 'fit_bounding_boxes_simple', 'propagate_labels', 'evaluate_sequence']      # reference default
SINGLE_STAGES = ['mask_ground_points', 'spatial_clustering', 'filter_detections', 'classification', 'fit_bounding_boxes_simple',
                 'evaluate_sequence']


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return str(s.getsockname()[1])


def _wait_all(procs, timeout=600):
    """Wait for all ranks; when one exits non-zero the others would wait in a collective forever: end them (by PID)."""
    import time
    t0 = time.time()
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs) or time.time() - t0 > timeout:
            time.sleep(2.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    for p in procs:
        p.wait(timeout=30)


def _load(root, seq='synthetic_train_0000', stage_list=DEFAULT_STAGES):
    stages = '_'.join(stage_list)
    with open(f'{root}/preprocessed_data/results/vilgod_mi355x/{stages}/{seq}.pkl', 'rb') as f:
        res = pickle.load(f)
    with open(f'{root}/preprocessed_data/results/vilgod_mi355x/{stages}/{seq}_indices.pkl', 'rb') as f:
        idx = pickle.load(f)
    with open(f'{root}/preprocessed_data/vilgod_mi355x_seq/{seq}.pkl', 'rb') as f:
        state = pickle.load(f)
    return res, idx, state


@pytest.mark.gpu
def test_cli_single_process_and_resume(cuda, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import preprocess_data
    root = str(tmp_path / 'd1')
    res = preprocess_data.main(['preprocessor=waymo', f'dataset.DATA_PATH={root}', 'export_pseudo_labels=True'] + OVR)
    out, idx, state = _load(root)
    from vilgod_amd import export
    import glob
    npz = glob.glob(f'{root}/**/pseudo_labels_*/synthetic_train_0000.npz', recursive=True)
    assert len(npz) == 1
    lab = export.read_npz(npz[0])
    assert len(lab) == 6 and [len(a['annos']['name']) for a in lab] == [len(fr['name']) for fr in out]
    assert len(out) == 6 and idx == [0, 1, 2, 3, 4, 5] and len(state) == 6
    for fr in out:
        assert set(fr) == {'boxes_lidar', 'name', 'score', 'moving'}
        assert fr['boxes_lidar'].shape[1] == 7 and fr['boxes_lidar'].dtype == np.float64
        assert len(fr['name']) == len(fr['score']) == len(fr['moving']) == len(fr['boxes_lidar'])
        assert set(fr['name']) <= {'Vehicle', 'Pedestrian', 'Cyclist'}
    key = 'clip_a_point_representation_of_a'
    n_moving = 0
    for st in state:
        assert set(st) >= {'_detections', '_ground_point_indices', '_entropy_scores', '_entropy_indices', '_gt_cluster_mapping'}
        assert len(st['_entropy_scores']) == len(st['_entropy_indices']) and np.all(st['_entropy_scores'] < 0.9)
        n_moving += sum(not d['static'] for d in st['_detections'])
        for d in st['_detections']:
            assert {'cluster_id', 'valid', 'static', 'gt_assigned', 'cluster_points_index', 'tid'} <= set(d)
            if d['valid']:
                assert len(d['_bounding_box']) == 7 and key in d['object_class'] and len(d['object_class_predictions'][key]) == 4
    assert sum(len(fr['name']) for fr in out) > 0
    assert n_moving > 0                                   # the synthetic world has moving objects: some clusters are not static
    tracked = [d for st in state for d in st['_detections'] if 'static_track' in d]
    assert tracked and any(not d['static_track'] for d in tracked)      # track_clusters + the track branch of the box fit ran
    assert any(fr['moving'].any() for fr in out)
    # the single-frame preset (no entropy stage, n_frames=1) on independent frames
    root3 = str(tmp_path / 'd3')
    preprocess_data.main(['preprocessor=waymo', f'dataset.DATA_PATH={root3}', 'pipeline_active=[' + ','.join(SINGLE_STAGES) + ']',
                          'pipeline.2.args.n_frames=1', 'dataset.SYNTHETIC.coherent=False'] + OVR)
    out3, idx3, state3 = _load(root3, stage_list=SINGLE_STAGES)
    assert len(out3) == 6 and '_entropy_scores' not in state3[0] and sum(len(fr['name']) for fr in out3) > 0
    # second run of the same preset: every stage finds its output in the sequence pickle and skips (zero_shot_detector.py resume
    # logic).  (The default stage list is not idempotent upstream either: propagate_labels invalidates short tracks' detections,
    # which changes what a second track_clusters sees.)
    os.remove(f'{root3}/preprocessed_data/results/vilgod_mi355x/' + '_'.join(SINGLE_STAGES) + '/synthetic_train_0000.pkl')
    preprocess_data.main(['preprocessor=waymo', f'dataset.DATA_PATH={root3}', 'pipeline_active=[' + ','.join(SINGLE_STAGES) + ']',
                          'pipeline.2.args.n_frames=1', 'dataset.SYNTHETIC.coherent=False', 'pipeline.6.args.force=False'] + OVR)
    out4, _, _ = _load(root3, stage_list=SINGLE_STAGES)
    for a, b in zip(out3, out4):
        assert np.array_equal(a['name'], b['name']) and np.allclose(a['boxes_lidar'], b['boxes_lidar'])


def _same_outputs(a, sa, b, sb):
    for x, y in zip(a, b):
        assert np.array_equal(x['name'], y['name']) and np.array_equal(x['moving'], y['moving'])
        assert np.array_equal(x['boxes_lidar'], y['boxes_lidar']) and np.array_equal(x['score'], y['score'])
    for x, y in zip(sa, sb):
        assert np.array_equal(x['_ground_point_indices'], y['_ground_point_indices'])
        assert len(x['_detections']) == len(y['_detections'])
        for d, e in zip(x['_detections'], y['_detections']):
            assert np.array_equal(d['cluster_points_index'], e['cluster_points_index']) and d['valid'] == e['valid']
            assert d['static'] == e['static'] and d.get('static_track') == e.get('static_track') and d['tid'] == e['tid']
            assert ('_bounding_box' in d) == ('_bounding_box' in e) and ('_bounding_box' not in d or np.array_equal(d['_bounding_box'], e['_bounding_box']))
            assert set(d.get('object_class', {})) == set(e.get('object_class', {}))
            for k in d.get('object_class_predictions', {}):
                assert np.array_equal(d['object_class_predictions'][k], e['object_class_predictions'][k])
                assert np.array_equal(d['object_class_predictions_score'][k], e['object_class_predictions_score'][k])


@pytest.mark.gpu
def test_cli_fused_stage_pass_equals_stage_after_stage(cuda, tmp_path):
    """device.fuse_stages (filter + classification of a frame ride along in spatial_clustering's frame pass) changes only WHEN the
    work runs: both pickle families equal those of the reference's stage-after-stage order, on the default 9-stage list."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import preprocess_data
    roots = {}
    for fuse in (True, False):
        roots[fuse] = str(tmp_path / f'fuse_{fuse}')
        preprocess_data.main(['preprocessor=waymo', f'dataset.DATA_PATH={roots[fuse]}', f'device.fuse_stages={fuse}'] + OVR)
        stages = preprocess_data.LAST_RUN['sequences'][0]['stage_ms_per_frame']
        assert set(DEFAULT_STAGES) <= set(stages)                   # every stage still ran in its turn
    a, ia, sa = _load(roots[True])
    b, ib, sb = _load(roots[False])
    assert ia == ib and len(a) == len(b) == 6 and sum(len(fr['name']) for fr in a) > 0
    _same_outputs(a, sa, b, sb)


@pytest.mark.gpu
def test_cli_sequence_tail_leaves_the_thread_when_it_would_touch_the_gpu(cuda, tmp_path):
    """device.overlap_sequences runs a sequence's host-only tail (track boxes, label propagation, pickles) on a thread under the next
    sequence's GPU stages -- only when it really is host-only (ZeroShotDetector.back_is_host_only): with the default box
    mode the static boxes were requested during classification and the tail runs on the thread; with device.box_mode=fast, or on a
    resumed run whose classification is skipped (nothing prefetched), the box stage launches kernels, so the tail stays on the main
    thread.  Either way both pickle families equal the run without overlap."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import preprocess_data
    two = [o for o in OVR if 'n_sequences' not in o and 'end_sequence' not in o] + ['dataset.SYNTHETIC.n_sequences=2', 'end_sequence=1']
    seqs = ('synthetic_train_0000', 'synthetic_train_0001')

    def run(name, *extra):
        root = str(tmp_path / name)
        preprocess_data.main(['preprocessor=waymo', f'dataset.DATA_PATH={root}'] + two + list(extra))
        return root, [q['tail_on_thread'] for q in preprocess_data.LAST_RUN['sequences']]
    base, on_thread = run('overlap')
    assert on_thread == [True, True]
    plain, on_thread = run('no_overlap', 'device.overlap_sequences=False')
    assert on_thread == [False, False]
    for q in seqs:
        a, _, sa = _load(base, q)
        b, _, sb = _load(plain, q)
        assert sum(len(fr['name']) for fr in a) > 0
        _same_outputs(a, sa, b, sb)
    # fast box mode: the tracked rows' rectangles come from vg_cluster_boxes inside the box stage -> main thread
    fast, on_thread = run('fast', 'device.box_mode=fast')
    assert on_thread == [False, False]
    fast2, _ = run('fast_no_overlap', 'device.box_mode=fast', 'device.overlap_sequences=False')
    for q in seqs:
        a, _, sa = _load(fast, q)
        b, _, sb = _load(fast2, q)
        _same_outputs(a, sa, b, sb)
    # resumed run: the first run stops after classification (state pickle on disk), the second finds every frame classified and
    # skips the stage -- no static-box requests were sent, the box stage fetches the points and asks itself -> main thread
    head = DEFAULT_STAGES[:6]
    root = str(tmp_path / 'resume')
    preprocess_data.main(['preprocessor=waymo', f'dataset.DATA_PATH={root}', 'pipeline_active=[' + ','.join(head) + ']'] + two)
    preprocess_data.main(['preprocessor=waymo', f'dataset.DATA_PATH={root}'] + two)
    assert [q['tail_on_thread'] for q in preprocess_data.LAST_RUN['sequences']] == [False, False]
    for q in seqs:
        a, _, sa = _load(root, q)
        b, _, sb = _load(plain, q)
        _same_outputs(a, sa, b, sb)


@pytest.mark.gpu
def test_cli_two_ranks_equal_one_rank(cuda, tmp_path):
    root1, root2 = str(tmp_path / 'one'), str(tmp_path / 'two')
    cli = os.path.join(ROOT, 'tools', 'preprocess_data.py')
    r = subprocess.run([sys.executable, cli, 'preprocessor=waymo', f'dataset.DATA_PATH={root1}'] + OVR,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=_free_port(), WORLD_SIZE='2', VILGOD_DIST_BACKEND='gloo')
    # the ranks log to files: with pipes, reading one rank's output while the other's pipe fills up would stall a collective
    logs = [open(tmp_path / f'rank{k}.log', 'w') for k in range(2)]
    procs = [subprocess.Popen([sys.executable, cli, 'preprocessor=waymo', f'dataset.DATA_PATH={root2}'] + OVR,
                              env=dict(env, RANK=str(k), LOCAL_RANK='0'), stdout=logs[k], stderr=subprocess.STDOUT, text=True)
             for k in range(2)]
    _wait_all(procs)
    for k, p in enumerate(procs):
        logs[k].close()
        assert p.returncode == 0, open(tmp_path / f'rank{k}.log').read()[-3000:]
    a, ia, sa = _load(root1)
    b, ib, sb = _load(root2)
    assert ia == ib and len(a) == len(b) == 6
    for x, y in zip(a, b):
        assert np.array_equal(x['name'], y['name'])
        assert np.array_equal(x['boxes_lidar'], y['boxes_lidar']) and np.array_equal(x['score'], y['score'])
    for x, y in zip(sa, sb):
        assert np.array_equal(x['_ground_point_indices'], y['_ground_point_indices'])
        assert np.array_equal(x['_entropy_indices'], y['_entropy_indices']) and np.array_equal(x['_entropy_scores'], y['_entropy_scores'])
        assert len(x['_detections']) == len(y['_detections'])
        for d, e in zip(x['_detections'], y['_detections']):
            assert np.array_equal(d['cluster_points_index'], e['cluster_points_index']) and d['valid'] == e['valid']
            assert d['static'] == e['static'] and d.get('static_track') == e.get('static_track')
            assert ('_bounding_box' in d) == ('_bounding_box' in e) and ('_bounding_box' not in d or np.array_equal(d['_bounding_box'], e['_bounding_box']))
    # one sequence on two ranks = frame sharding (device.shard=auto): the part that does not shrink with the number of ranks --
    # the sequence-level stages every rank repeats -- is measured and logged by the entry point
    rep = [ln for ln in open(tmp_path / 'rank0.log').read().splitlines() if 'repeated on every rank' in ln]
    assert rep, 'the entry point reports the replicated sequence-level stages of a frame-sharded run'
    print(rep[-1].split(' - ')[-1].strip())


@pytest.mark.gpu
def test_cli_sequence_sharding_equals_one_rank(cuda, tmp_path):
    """device.shard=sequences: rank r processes sequences r, r + N, ... on its own (no exchange inside a sequence) -- both pickle
    families of every sequence equal those of a one-rank run, and rank 0's final evaluation sees all sequences in order."""
    ovr = [o for o in OVR if not o.startswith(('dataset.SYNTHETIC.n_sequences', 'end_sequence'))] + \
        ['dataset.SYNTHETIC.n_sequences=2', 'end_sequence=1']
    root1, root2 = str(tmp_path / 'one'), str(tmp_path / 'two')
    cli = os.path.join(ROOT, 'tools', 'preprocess_data.py')
    r = subprocess.run([sys.executable, cli, 'preprocessor=waymo', f'dataset.DATA_PATH={root1}'] + ovr, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=_free_port(), WORLD_SIZE='2', VILGOD_DIST_BACKEND='gloo')
    logs = [open(tmp_path / f'rank{k}.log', 'w') for k in range(2)]
    procs = [subprocess.Popen([sys.executable, cli, 'preprocessor=waymo', f'dataset.DATA_PATH={root2}', 'device.shard=sequences'] + ovr,
                              env=dict(env, RANK=str(k), LOCAL_RANK='0'), stdout=logs[k], stderr=subprocess.STDOUT, text=True)
             for k in range(2)]
    _wait_all(procs)
    for k, p in enumerate(procs):
        logs[k].close()
        assert p.returncode == 0, open(tmp_path / f'rank{k}.log').read()[-3000:]
    for seq in ('synthetic_train_0000', 'synthetic_train_0001'):
        a, ia, sa = _load(root1, seq=seq)
        b, ib, sb = _load(root2, seq=seq)
        assert ia == ib and len(a) == len(b) == 6 and sum(len(fr['name']) for fr in a) > 0
        _same_outputs(a, sa, b, sb)
    log0 = open(tmp_path / 'rank0.log').read()
    assert 'Evaluate all Sequences' in log0 and 'Evaluate all Sequences' not in open(tmp_path / 'rank1.log').read()
    # the one-rank run and rank 0 of the sharded run evaluate the same 12 frames: the same summary line
    line = [ln.split('Summary over all sequences:')[1] for ln in r.stdout.splitlines() if 'Summary over all sequences:' in ln]
    line0 = [ln.split('Summary over all sequences:')[1] for ln in log0.splitlines() if 'Summary over all sequences:' in ln]
    assert line and line == line0


@pytest.mark.gpu
def test_cli_two_processes_per_gpu(cuda, tmp_path):
    """device.processes_per_gpu=2 under torch.distributed.run (tools/time_cli.py PROCS=2 starts it the way a user would): two ranks
    share cuda:0, rank r walks sequences r, r + 2, ...; gloo process group chosen by the entry point itself."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'time_cli.py'), '6', '20000', 'dataset.SYNTHETIC.objects_per_frame=10'],
                       env=dict(os.environ, PROCS='2', SEQUENCES='2', SEED_STRIDE='1', TIME_CLI_JSON='1'), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    run = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('TIME_CLI_JSON ')][-1][len('TIME_CLI_JSON '):])
    assert run['processes'] == 2 and [q['frames'] for q in run['sequences']] == [6, 6]
    assert [q['name'] for q in run['sequences']] == ['synthetic_train_0000', 'synthetic_train_0001'] and run['loop_seconds'] > 0


@pytest.mark.gpu
@pytest.mark.parametrize('handoff', ['replicate', 'relay', 'chain'])
def test_bench_two_ranks_on_one_gpu(cuda, handoff):
    """bench.py's N > 1 path (barriers, padded all-gather of the score matrices, max-over-ranks timing, one JSON line from
    rank 0) with two processes sharing the single GPU of the test box over gloo; the driver's multi-GPU runs use RCCL.
    replicate (default): frames dealt round-robin, every rank runs all ground passes itself; chain: blocks + state hand-off."""
    import json
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=_free_port(), WORLD_SIZE='2', VILGOD_DIST_BACKEND='gloo')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--points', '30000',
           '--objects', '12', '--no-cpu-baseline', '--no-sequence-pass', '--no-extras', '--ground-handoff', handoff]
    import tempfile
    tmp = tempfile.mkdtemp()
    files = [(open(f'{tmp}/o{k}', 'w'), open(f'{tmp}/e{k}', 'w')) for k in range(2)]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(k), LOCAL_RANK='0'), stdout=files[k][0], stderr=files[k][1], text=True)
             for k in range(2)]
    _wait_all(procs)
    for fo, fe in files:
        fo.close(); fe.close()
    outs = [(open(f'{tmp}/o{k}').read(), open(f'{tmp}/e{k}').read()) for k in range(2)]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, o[-2000:] + e[-3000:]
    lines = [l for l in outs[0][0].splitlines() if l.startswith('{')]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith('{')]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['scaling'] == 'weak' and d['value'] > 0 and d['unit'] == 'frames/s'
    assert d['value'] == pytest.approx(2 * 3 / (d['ms_per_step'] * 3 / 1000.0), rel=1e-3)       # whole-job frames / max-rank time
    assert 0 < d['roofline']['frac'] < 1 and 'cpu_baseline' not in d


@pytest.mark.gpu
def test_bench_two_ranks_over_rccl_when_two_gpus_are_visible(cuda):
    """The driver's multi-GPU command line with backend nccl (= RCCL) on two devices: init_process_group with device_id, the barriers,
    all_gather + all_gather_into_tensor on device tensors, the all-reduce of the times -- so that the 8-GPU scaling run is not the first
    execution of that code.  RCCL refuses two ranks on one device: skipped on the one-GPU test boxes."""
    import json
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs (RCCL refuses two ranks on one device); the gloo twin of this test runs on one')
    env = dict(os.environ, MASTER_PORT=_free_port())
    env.pop('VILGOD_DIST_BACKEND', None)
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '1', '--points', '30000',
                        '--objects', '12', '--no-cpu-baseline', '--no-sequence-pass', '--no-extras'], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 4 and d['scaling'] == 'weak' and d['value'] > 0


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['waymo', 'argoverse'])
def test_cli_on_openpcdet_layout(cuda, tmp_path, kind, caplog):
    """the default 9-stage configuration over a data set in OpenPCDet's on-disk layout (written from the seeded
    generator), read by the real-data adapters, evaluated with the AP / APH evaluation against the generator's ground truth."""
    import logging
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import preprocess_data
    from vilgod_amd import fixture_data as fx
    root = str(tmp_path / kind)
    writer = fx.write_waymo if kind == 'waymo' else fx.write_argo2
    names = writer(root, n_sequences=1, n_frames=6, n_points=20000, n_objects=10, seed=0)
    with caplog.at_level(logging.INFO):
        res = preprocess_data.main([f'preprocessor={kind}', f'dataset={kind}_openpcdet', f'dataset.DATA_PATH={root}', 'end_sequence=1',
                                    'device.max_points=24000', 'paths.clip_model=/nonexistent', 'export_pseudo_labels=True',
                                    'pipeline.8.args.detection_3d.class_agnostic=True', 'pipeline.8.args.eval_range=[-75.,-75.,75.,75.]'])
    assert len(res) == 6 and sum(len(fr['name']) for fr in res) > 0
    out, idx, state = _load(root, seq=names[0])
    assert sorted(idx) == [0, 1, 2, 3, 4, 5] and len(state) == 6       # Argoverse infos are stored out of time order
    text = caplog.text
    assert 'Vehicle AP  L2:' in text and 'Vehicle APH L2:' in text
    ap = float(text.split('Vehicle AP  L2:')[1].split()[0])
    assert 0.0 <= ap <= 100.0
    # what the entry point's evaluation must print for detections whose quality is known: the generator's own ground truth
    # handed in as detections (load_detection_results) is AP = APH = 100.00; the ground truth of the first three frames only is
    # precision 1 up to recall n_first / n_total
    from vilgod_amd import config as vconfig, evaluation as ev
    ov = [f'preprocessor={kind}', f'dataset={kind}_openpcdet', f'dataset.DATA_PATH={root}', 'end_sequence=1', 'device.max_points=24000',
          'paths.clip_model=/nonexistent', 'pipeline.8.args.detection_3d.class_agnostic=True', 'pipeline.8.args.eval_range=[-75.,-75.,75.,75.]']
    cfg = vconfig.load(os.path.join(ROOT, 'tools', 'configs'), 'preprocessing', ov)
    ds = vconfig.instantiate(cfg.dataset_class, logger=logging.getLogger('t'), training=True, start_sequence=0, end_sequence=1)
    ds.training = False
    next(iter(ds.next_sequence()))
    truth = []
    for i in ds.sequence_indices:
        a = ds.infos[i]['annos']
        keep = np.isin(a['name'], ds.class_names) & (a['num_points_in_gt'] > 0)
        truth.append({'boxes_lidar': a['gt_boxes_lidar'][keep][:, :7].astype(np.float64), 'name': a['name'][keep],
                      'score': np.full(int(keep.sum()), 0.9, np.float32), 'moving': np.zeros(int(keep.sum()), bool)})
    assert sum(len(t['name']) for t in truth) > 10
    first = [t if k < 3 else {'boxes_lidar': np.zeros((0, 7)), 'name': np.array([]), 'score': np.array([]), 'moving': np.array([], bool)}
             for k, t in enumerate(truth)]
    _, gts = ev.filter_for_evaluation(ds, truth, ds.class_names, indices=ds.sequence_indices, class_agnostic=True,
                                      eval_range=[-75., -75., 75., 75.], moving=False, static=False, score_thresh=0.0, sampling_rate=1)
    n_gt = [int((g['num_points_in_gt'] > 0).sum()) for g in gts]
    for name_, dets, want in (('all', truth, 100.0), ('first3', first, 100.0 * sum(n_gt[:3]) / sum(n_gt))):
        path = str(tmp_path / f'{kind}_{name_}.pkl')
        with open(path, 'wb') as f:
            pickle.dump(dets, f)
        caplog.clear()
        with caplog.at_level(logging.INFO):
            preprocess_data.main(ov + ['load_detection_results=True', f'result_path={path}'])
        for key in ('Vehicle AP  L2:', 'Vehicle APH L2:'):
            got = float(caplog.text.split(key)[1].split()[0])
            assert abs(got - want) <= 0.006, (name_, key, got, want)
    # load_detection_results: evaluate the stored result pickle without processing anything (preprocess_data.py:66-70,108-110)
    stages = '_'.join(DEFAULT_STAGES)
    caplog.clear()
    with caplog.at_level(logging.INFO):
        res2 = preprocess_data.main([f'preprocessor={kind}', f'dataset={kind}_openpcdet', f'dataset.DATA_PATH={root}', 'end_sequence=1',
                                     'device.max_points=24000', 'paths.clip_model=/nonexistent', 'load_detection_results=True',
                                     f'result_path={root}/preprocessed_data/results/vilgod_mi355x/{stages}/{names[0]}.pkl',
                                     'pipeline.8.args.detection_3d.class_agnostic=True', 'pipeline.8.args.eval_range=[-75.,-75.,75.,75.]'])
    assert len(res2) == 6 and 'stage classification' not in caplog.text
    assert float(caplog.text.split('Vehicle AP  L2:')[1].split()[0]) == ap
