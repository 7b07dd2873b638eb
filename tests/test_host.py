"""Host logic that needs no GPU: config loader (Hydra syntax of the reference CLI), dataset duck-type, frame
sharding, and the N>1 score/result gather on a world_size-2 gloo group."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from conftest import ROOT
from vilgod_amd import config as vconfig
from vilgod_amd import dist as vdist

CFG = os.path.join(ROOT, 'tools', 'configs')


def test_config_defaults_and_resolvers():
    c = vconfig.load(CFG)
    assert c.preprocessor.name == 'waymo' and c.dataset.DATASET == 'SyntheticDataset'
    key = [t for t in c.pipeline if t['name'] == 'classification'][0].args.key
    assert key == 'clip_a_point_representation_of_a'                    # the resolved classification key
    assert c.preprocessor.lidar_image_projection.maxpool.kernel_size == (1, 5, 5)
    assert c.preprocessor.clustering.model.min_cluster_size == 15 and c.preprocessor.clustering.propability_threshold == 0.3
    assert c.paths.pseudo_label.endswith('pseudo_labels_vilgod_waymo')
    assert len(c.preprocessor.clip.class_list) == 24 and set(c.preprocessor.clip.class_mapping.values()) == {'Vehicle', 'Pedestrian', 'Cyclist', 'Background'}


def test_config_cli_overrides_like_hydra():
    c = vconfig.load(CFG, 'preprocessing', ['preprocessor=argoverse', 'start_sequence=3', 'end_sequence=7', 'split=val',
                                            'pipeline_active=[mask_ground_points,spatial_clustering]',
                                            'preprocessor.clustering.model.min_cluster_size=20', 'pipeline.2.args.force=True',
                                            'paths.clip_model=/models', 'device.vit_dtype=f32'])
    assert c.preprocessor.name == 'argo2' and c.dataset.SYNTHETIC.points_per_frame == 200000
    assert (c.start_sequence, c.end_sequence, c.split) == (3, 7, 'val')
    assert c.pipeline_active == ['mask_ground_points', 'spatial_clustering']
    assert c.preprocessor.clustering.model.min_cluster_size == 20 and c.pipeline[2].args.force is True
    assert c.paths.clip_model == '/models' and c.device.vit_dtype == 'f32'


def test_dataset_duck_type():
    c = vconfig.load(CFG, overrides=['dataset.SYNTHETIC.frames_per_sequence=3', 'dataset.SYNTHETIC.points_per_frame=5000',
                                     'dataset.SYNTHETIC.objects_per_frame=4', 'end_sequence=1'])
    ds = vconfig.instantiate(c.dataset_class, logger=None, training=True, start_sequence=c.start_sequence, end_sequence=c.end_sequence)
    names = list(ds.next_sequence())
    assert len(names) == 2 and ds.sequence_length == 3 and len(ds.sequence_indices) == 3
    pts = ds.get_lidar_points(1)
    assert pts.shape == (5000, 5) and pts.dtype == np.float32
    assert np.array_equal(pts, ds.get_lidar_points(1))                  # deterministic
    assert ds.sequence_infos[2]['pose'].shape == (4, 4) and set(ds.get_annos(0)) >= {'gt_names', 'moving'}
    assert ds.class_names == ['Vehicle', 'Pedestrian', 'Cyclist']


def test_shard_frames_partition():
    for n in [0, 1, 7, 199]:
        for w in [1, 2, 3, 8]:
            parts = [vdist.shard_frames(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
            assert all(p == list(range(p[0], p[0] + len(p))) for p in parts if p)


WORKER = textwrap.dedent('''
    import os, sys, torch, numpy as np
    sys.path.insert(0, %r)
    import torch.distributed as dist
    from vilgod_amd import dist as vdist
    rank, ws = vdist.init_from_env('gloo')
    assert ws == 2
    frames = vdist.shard_frames(5, rank, ws)
    g = torch.Generator().manual_seed(100)
    allm = {f: torch.rand((3 + f, 24), generator=g) for f in range(5)}       # same on both ranks
    local = {f: allm[f] for f in frames}
    got = vdist.gather_scores(local, 24, torch.device('cpu'))
    assert sorted(got) == list(range(5))
    for f in range(5):
        assert torch.equal(got[f], allm[f]), f
    objs = vdist.gather_objects({f: {'name': np.array(['Vehicle'] * f)} for f in frames})
    merged = {}
    for o in objs:
        merged.update(o)
    assert sorted(merged) == list(range(5)) and len(merged[4]['name']) == 4
    # empty shard on one rank
    got = vdist.gather_scores(local if rank == 0 else {}, 24, torch.device('cpu'))
    assert sorted(got) == vdist.shard_frames(5, 0, 2)
    # ground-state hand-off down the rank chain (vilgod_amd/dist.py): rank 1 receives what rank 0 exported AFTER its block
    from vilgod_amd._lib import lib
    nbytes = int(lib.vg_ground_state_bytes())
    class FakeGround:
        def __init__(self): self.state, self.ran = None, False
        def export_state(self):
            assert self.ran
            return bytes([7 + rank]) * nbytes
        def set_state(self, blob): self.state = bytes(blob)
    fg = FakeGround()
    def block():
        assert (fg.state is not None) == (rank > 0)
        fg.ran = True
        return rank
    assert vdist.chain_ground_state(fg, block) == rank
    assert fg.state == (bytes([7]) * nbytes if rank == 1 else None)
    dist.barrier()
    print('rank', rank, 'ok')
''')


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return str(s.getsockname()[1])


def test_two_rank_gather_gloo(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=_free_port(), WORLD_SIZE='2', OMP_NUM_THREADS='1')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f'rank {r} ok' in o


RELAY_WORKER = """
import os, sys
sys.path.insert(0, %r)
import torch
import torch.distributed as dist
from vilgod_amd import dist as vdist
from vilgod_amd._lib import lib
vdist.init_from_env('gloo')
rank, ws = vdist.world()
nbytes = int(lib.vg_ground_state_bytes())

class FakeGround:                      # the state is the list of frames whose "ground pass" has run, in order
    def __init__(self): self.seen = []
    def export_state(self):
        b = bytearray(nbytes); b[0] = len(self.seen)
        for i, g in enumerate(self.seen): b[1 + i] = g
        return bytes(b)
    def set_state(self, blob): self.seen = list(blob[1:1 + blob[0]])

gm = FakeGround()
n_total = 7
for g in range(n_total):
    if g %% ws != rank:
        continue
    vdist.relay_recv_state(gm, g)
    assert gm.seen == list(range(g)), (rank, g, gm.seen)       # every pass before frame g, in order, whoever ran it
    gm.seen.append(g)                                            # this rank's ground pass of frame g
    vdist.relay_send_state(gm, g, n_total)
dist.barrier()
print(f'rank {rank} ok', flush=True)
"""


def test_two_rank_ground_state_relay_gloo(tmp_path):
    """bench.py --ground-handoff relay (round 6): frames dealt round-robin, the Patchwork++ state relayed frame by frame -- rank g % N
    takes it from the owner of frame g - 1 before its ground pass of frame g and hands it on behind it.  Two gloo ranks, seven frames,
    a stand-in ground model whose state is the list of passes run so far: every pass sees exactly the passes before it, in order."""
    script = tmp_path / 'relay_worker.py'
    script.write_text(RELAY_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=_free_port(), WORLD_SIZE='2', OMP_NUM_THREADS='1')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f'rank {r} ok' in o


def test_pseudo_label_exporter_roundtrip(tmp_path):
    """N3 exporter: result dicts -> OpenPCDet infos-style pickle + NPZ and back (empty frames included)."""
    import pickle
    import numpy as np
    from vilgod_amd import export
    rng = np.random.default_rng(0)
    results = []
    for n in (3, 0, 5):
        results.append({'boxes_lidar': rng.normal(size=(n, 7)), 'name': np.array(['Vehicle', 'Pedestrian', 'Cyclist', 'Vehicle', 'Sign'][:n]),
                        'score': rng.uniform(0.2, 1.0, size=n), 'moving': np.zeros(n, bool)})
    pkl, npz = export.write_sequence(tmp_path, 'seq0', results, ['a', 'b', 'c'], [10, 11, 12], score_thresh=0.0,
                                     class_names=['Vehicle', 'Pedestrian', 'Cyclist'])
    infos = pickle.load(open(pkl, 'rb'))
    assert [i['frame_id'] for i in infos] == ['a', 'b', 'c'] and [i['sample_idx'] for i in infos] == [10, 11, 12]
    assert [len(i['annos']['name']) for i in infos] == [3, 0, 4]                  # 'Sign' is not a training class
    assert infos[0]['annos']['gt_boxes_lidar'].dtype == np.float32 and infos[0]['annos']['gt_boxes_lidar'].shape == (3, 7)
    back = export.read_npz(npz)
    for a, b in zip(infos, back):
        assert a['frame_id'] == b['frame_id'] and np.array_equal(a['annos']['name'], b['annos']['name'])
        assert np.array_equal(a['annos']['gt_boxes_lidar'], b['annos']['gt_boxes_lidar'])
        assert np.array_equal(a['annos']['score'], b['annos']['score'])


def test_stage_fusion_plan_follows_the_stage_list():
    """ZeroShotDetector._fusion_plan: which later stages ride along in spatial_clustering's frame pass.  Only stages separated from
    it by nothing but track_clusters, never a stage with `force` (it would redo the work), nothing with per-stage state files."""
    from types import SimpleNamespace
    from vilgod_amd.zero_shot_detector import ZeroShotDetector
    cfg_dir = os.path.join(ROOT, 'tools', 'configs')

    def plan(*overrides, sync=False):
        cfg = vconfig.load(cfg_dir, 'preprocessing', ['preprocessor=waymo'] + list(overrides))
        return list(ZeroShotDetector._fusion_plan(SimpleNamespace(cfg=cfg, sync_every_stage=sync)))

    assert plan() == ['filter_detections', 'classification']                       # the reference's default 9-stage list
    assert plan('device.fuse_stages=False') == [] and plan(sync=True) == []
    assert plan('pipeline.5.args.force=True') == ['filter_detections']             # classification forced: runs in its own turn
    assert plan('pipeline.3.args.force=True') == []                                # filter forced: valid_only classification needs it first
    single = 'pipeline_active=[mask_ground_points,spatial_clustering,filter_detections,classification,fit_bounding_boxes_simple,evaluate_sequence]'
    assert plan(single) == ['filter_detections', 'classification']
    assert plan('pipeline_active=[mask_ground_points,spatial_clustering,classification]') == []          # valid_only without the filter stage
    assert plan('pipeline_active=[mask_ground_points,filter_detections,classification]') == []           # no clustering stage to ride in


def test_box_helper_pool_survives_a_dead_helper():
    """box_mode='reference': a helper process that died is replaced and the request repeated; results equal the in-thread ones."""
    from vilgod_amd import boxes
    rng = np.random.default_rng(0)
    xy = rng.normal(size=(300, 2)).astype(np.float32)
    seg = np.array([0, 100, 180, 300], np.int64)
    zmin, zmax = np.zeros(3, np.float32), np.ones(3, np.float32)
    want = boxes.reference_boxes_packed(xy, seg, zmin, zmax)
    pool = boxes.BoxWorkerPool(1)
    try:
        assert np.array_equal(pool.submit(xy, seg, zmin, zmax).result(), want)
        pool.procs[0].kill()
        pool.procs[0].wait(timeout=10)
        assert np.array_equal(pool.submit(xy, seg, zmin, zmax).result(), want)
        assert pool.respawned == 1
        pool.grow(2)
        assert len(pool.procs) == 2
        futs = [pool.submit(xy, seg, zmin, zmax) for _ in range(6)]
        assert all(np.array_equal(f.result(), want) for f in futs)
    finally:
        pool.close()


def _frame_state_with_results(seed=0):
    from vilgod_amd.frame_state import FrameState, pack_clusters
    rng = np.random.default_rng(seed)
    fs = FrameState(3, np.eye(4), np.eye(4))
    labels = rng.integers(-1, 7, 400)
    fs.set_clusters(*pack_clusters(labels, rng.random(400), 0.1))
    C = fs.n_detections
    fs.ground_point_indices = np.sort(rng.choice(1000, 300, replace=False))
    fs.entropy_scores, fs.entropy_indices = rng.random(50), np.sort(rng.choice(400, 50, replace=False))
    fs.valid = rng.random(C) > 0.3
    fs.filtered = True
    fs.static = rng.random(C) > 0.5
    fs.static_track[:] = rng.integers(-1, 2, C)
    fs.boxes = rng.random((C, 7))
    fs.boxes[0] = np.nan
    names = np.array(['Background', 'Cyclist', 'Pedestrian', 'Vehicle'], dtype=object)
    which = fs.valid.copy()
    n = int(which.sum())
    fs.set_classes('clip', which, names[rng.integers(0, 4, (n, 4))], names[rng.integers(0, 4, (n, 4))], rng.random((n, 4)).astype(np.float32),
                   names[rng.integers(0, 4, n)], rng.random(n).astype(np.float32))
    fs.set_final_score('clip', int(np.flatnonzero(which)[0]), 0.7)            # a propagated python-float score
    return fs


def _same(a, b):
    if isinstance(a, dict):
        return isinstance(b, dict) and list(a) == list(b) and all(_same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return type(a) is type(b) and len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray):
        return isinstance(b, np.ndarray) and a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b, equal_nan=a.dtype.kind == 'f')
    return type(a) is type(b) and (a == b or (a != a and b != b))


def test_frame_state_compact_form_serialises_identically_and_is_detached():
    """FrameState.compact() -> from_compact().serialize (what the state-writer process does) gives the dict FrameState.serialize gives:
    same keys in the same order, same value types (numpy bool / numpy str / float32-vs-python-float scores) and values; and later
    in-place changes of the live state (what propagate_labels does) do not reach a compact form taken before them."""
    from vilgod_amd.frame_state import FrameState
    fs = _frame_state_with_results()
    want = fs.serialize
    c = fs.compact()
    assert _same(FrameState.from_compact(c).serialize, want)
    fs.boxes[1] = 5.0
    fs.valid[:] = False
    fs.static_track[:] = 1
    fs.cls['clip']['name'][np.flatnonzero(fs.cls['clip']['has'])[0]] = 'Vehicle'
    fs.set_final_score('clip', int(np.flatnonzero(fs.cls['clip']['has'])[1]), 1.0)
    assert _same(FrameState.from_compact(c).serialize, want)
    assert not _same(fs.serialize, want)


def test_state_writer_process_writes_the_reference_layout(tmp_path):
    """python -m vilgod_amd.state_writer (the helper process of device.async_state_write) on compact frame states: the file holds
    exactly [FrameState.serialize ...], appears under its final name only when complete, and a second request is served too."""
    import pickle
    from vilgod_amd import zero_shot_detector as zsd
    frames = [_frame_state_with_results(s) for s in range(3)]
    want = [f.serialize for f in frames]
    try:
        for name in ('a.pkl', 'b.pkl'):
            path = tmp_path / name
            zsd._submit_state_write(path, [f.compact() for f in frames])
            zsd.wait_state_writes()
            assert zsd._STATE_WRITER['proc'] is not None and zsd._STATE_WRITER['proc'].poll() is None     # served by the helper, which lives on
            with open(path, 'rb') as fp:
                assert _same(pickle.load(fp), want)
            assert not (tmp_path / (name + '.tmp')).exists()
    finally:
        zsd.shutdown_state_writer()


def test_cu_mask_words_partition_the_device():
    """vilgod_amd/streams.py: the front mask is the low 8 r bits (r CUs of every XCD under amdkfd's round-robin deal), the tower mask
    everything else of the device's CUs; together they cover every CU exactly once."""
    from vilgod_amd.streams import cu_mask_words
    for n_cu, r in ((256, 1), (256, 2), (256, 5), (304, 3), (64, 7)):
        f, t = cu_mask_words(n_cu, r, 'front'), cu_mask_words(n_cu, r, 'tower')
        assert f.dtype == np.uint32 and len(f) == len(t) == (n_cu + 31) // 32
        bits_f = np.unpackbits(f.view(np.uint8), bitorder='little')[:n_cu]
        bits_t = np.unpackbits(t.view(np.uint8), bitorder='little')[:n_cu]
        assert bits_f.sum() == 8 * r and bits_f[:8 * r].all()
        assert np.array_equal(bits_f + bits_t, np.ones(n_cu, np.uint8))
        assert not np.unpackbits(t.view(np.uint8), bitorder='little')[n_cu:].any()
    with pytest.raises(ValueError):
        cu_mask_words(256, 32, 'front')
    with pytest.raises(ValueError):
        cu_mask_words(256, 0, 'tower')
    with pytest.raises(ValueError):
        cu_mask_words(100, 2, 'front')          # not a multiple of 8 XCDs: another partition mode, the mask layout is unknown


def test_pack_clusters_host_equals_numpy_form():
    """frame_state.pack_clusters (vg_pack_clusters_host: one counting sort in C++) against the numpy grouping it replaced
    (lidar_frame.py:163-167, 230-237): ids, packed indices and offsets equal, dtypes too -- random labels incl. empty input, all noise,
    gaps in the label range and thresholded probabilities."""
    from vilgod_amd.frame_state import pack_clusters, pack_clusters_numpy
    rng = np.random.default_rng(5)
    cases = [(np.zeros(0, np.int64), None), (np.full(7, -1), rng.random(7)), (np.array([3, 3, -1, 9, 3, 9]), None),
             (np.array([2_000_000_000, 5, 2_000_000_000, -1, 70_000, 5]), None)]          # label values far beyond the point count
    for n in (1, 17, 1000, 50_000):
        lab = rng.integers(-1, 40, n)
        lab[lab == 5] = 7                                  # a label nobody owns
        cases += [(lab, rng.random(n)), (lab.astype(np.int32), None)]
    for lab, pr in cases:
        got, want = pack_clusters(lab, pr, 0.3), pack_clusters_numpy(lab, pr, 0.3)
        for g, w in zip(got, want):
            assert g.dtype == w.dtype and np.array_equal(g, w)


@pytest.mark.parametrize('family', ['pp64', 'w4'])
def test_profile_summary_splits_the_projection_gemm_per_kind(tmp_path, family):
    """tools/summarize_profiles.py (VERDICT r4 task 2: FETCH / WRITE / SQ counters per GEMM KIND): a fabricated rocprofv3 run -- two frames of two
    blocks each, the instantiations' mangled names as the product library emits them, out_proj and c_proj sharing one instantiation and
    alternating in dispatch order, plus the small class-token launches of the last block -- must come out with each kind's own time, bytes
    and MFMA utilisation, and the small launches must not be counted."""
    import csv
    import json
    src, dst = tmp_path / 'prof', tmp_path / 'out'
    for d in ('trace', 'fetch', 'write', 'sq'):
        (src / d).mkdir(parents=True)
    # (round 6: k_gemm_f16_w4 launches a persistent grid of one workgroup per CU -- the rows of a launch come from the encode's
    # k_embed_lnpre dispatch in front of it)
    if family == 'pp64':
        name = lambda epi, ln: f'_Z15k_gemm_f16_pp64ILi{epi}ELb0ELb0ELi{ln}ELb0EEvPKDF16_S1_PKfPvPfiiiiiPxS3_P9LnPartialPDF16_i'
    else:
        name = lambda epi, ln: f'_Z13k_gemm_f16_w4ILi{epi}ELi{ln}ELi0EEvPKDF16_S1_PKfPvPfiiiiiS3_P9LnPartialPDF16_Px'
    dom = 'k_gemm_f16_pp64' if family == 'pp64' else 'k_gemm_f16_w4'
    rows_m = 65536                                            # 256 row tiles
    kinds = {'in_proj': (name(0, 1), 9, 270_000, 100.0, 300.0, 0.45), 'out_proj': (name(2, 2), 3, 160_000, 330.0, 300.0, 0.22),
             'c_fc': (name(1, 1), 12, 400_000, 110.0, 400.0, 0.42), 'c_proj': (name(2, 2), 3, 350_000, 740.0, 301.0, 0.46)}
    disp = []
    for frame in range(2):
        disp.append(('lnpre', '_Z13k_embed_lnpreIDF16_EvPKfS1_S1_S1_S1_PT_iiiiS1_S1_PDF16_', rows_m // 4 * 256))
        for block in range(2):
            for k in ('in_proj', 'out_proj', 'c_fc', 'c_proj'):
                disp.append((k, kinds[k][0], (rows_m // 256) * kinds[k][1] * 512 if family == 'pp64' else 256 * 256))
        disp += [('small', name(2, 2), 2 * 3 * 512 if family == 'pp64' else 8 * 256), ('small', name(1, 1), 2 * 12 * 512 if family == 'pp64' else 24 * 256)]      # class-token rows of the last block
    trace, fetch, write, sq = [], [], [], []
    t = 1_000_000
    for did, (k, nm, grid) in enumerate(disp, start=1):
        dur = kinds[k][2] if k in kinds else 30_000
        trace.append({'Kind': 'KERNEL_DISPATCH', 'Dispatch_Id': did, 'Kernel_Name': nm, 'Start_Timestamp': t, 'End_Timestamp': t + dur, 'Grid_Size_X': grid})
        base = {'Dispatch_Id': did, 'Kernel_Name': nm, 'Grid_Size': grid, 'Start_Timestamp': t, 'End_Timestamp': t + 2 * dur}
        f_mb, w_mb, util = (kinds[k][3], kinds[k][4], kinds[k][5]) if k in kinds else (1.0, 1.0, 0.05)
        fetch.append(dict(base, Counter_Name='FETCH_SIZE', Counter_Value=f_mb * 1e6 / 2048))             # KB, under-reported x2 on gfx950
        write.append(dict(base, Counter_Name='WRITE_SIZE', Counter_Value=w_mb * 1e6 / 1024))
        gui = 8 * 2 * dur * 2.0                                                                          # 8 XCDs x ns x GHz
        for cn, cv in (('GRBM_GUI_ACTIVE', gui), ('SQ_VALU_MFMA_BUSY_CYCLES', util * gui / 8 * 1024), ('SQ_WAVE_CYCLES', 1000.0), ('SQ_WAIT_ANY', 300.0),
                       ('SQ_WAIT_INST_ANY', 400.0), ('SQ_ACTIVE_INST_ANY', 300.0), ('SQ_BUSY_CYCLES', 1.0)):
            sq.append(dict(base, Counter_Name=cn, Counter_Value=cv))
        t += 3 * dur

    def dump(path, rows):
        with open(path, 'w', newline='') as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(rows)
    dump(src / 'trace' / 'bench_kernel_trace.csv', trace)
    dump(src / 'fetch' / 'bench_counter_collection.csv', fetch)
    dump(src / 'write' / 'bench_counter_collection.csv', write)
    dump(src / 'sq' / 'bench_counter_collection.csv', sq)
    stats = {}
    for r in trace:
        st = stats.setdefault(r['Kernel_Name'], [0, 0])
        st[0] += 1
        st[1] += r['End_Timestamp'] - r['Start_Timestamp']
    dump(src / 'trace' / 'bench_kernel_stats.csv', [{'Name': k, 'Calls': c, 'TotalDurationNs': d, 'AverageNs': d / c} for k, (c, d) in stats.items()])
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'summarize_profiles.py'), 'tst'], capture_output=True, text=True,
                       env=dict(os.environ, VG_PROFILE_SRC=str(src), VG_PROFILE_OUT=str(dst)))
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.load(open(dst / 'tst_pmc_summary.json'))[dom + '_by_kind']
    assert set(out) == set(kinds)
    for k, (_, ntn, dur, f_mb, w_mb, util) in kinds.items():
        o = out[k]
        assert o['launches'] == 4 and o['avg_rows'] == rows_m and abs(o['avg_launch_us'] - dur / 1e3) < 0.1
        assert abs(o['fetch_bytes_corrected_x2'] - f_mb * 1e6) < 1e3 and abs(o['write_bytes'] - w_mb * 1e6) < 1e3
        assert abs(o['mfma_utilisation'] - util) < 1e-3 and abs(o['sustained_ghz'] - 2.0) < 1e-3
    assert json.load(open(dst / 'gemm_traffic.json'))['by_kind']['out_proj']['avg_launch_us'] == out['out_proj']['avg_launch_us']
