"""End-to-end hot path [A]-[F] on one frame: GPU pipeline vs the chained CPU oracle (BASELINE config 1 shape:
single synthetic 20k-point frame).  fp32 ViT: probabilities within 1e-3, identical classes / valid flags /
ground set / cluster labels; boxes equal to the reference-rule oracle (box_mode='reference')."""
import numpy as np
import pytest
import torch

from vilgod_amd import synthetic, clip_weights as cw


@pytest.mark.gpu
def test_pipeline_matches_oracle_20k(cuda):
    from vilgod_amd.pipeline import PseudoLabelPipeline, default_preprocessor_cfg
    from vilgod_amd.clip_wrapper import ClipWrapper
    from oracle.pipeline_oracle import OraclePipeline
    from oracle import segment_oracle as so
    cfg = default_preprocessor_cfg()
    pts = synthetic.make_frame(3, 20_000, n_objects=12)
    poses = synthetic.make_poses(2, seed=4)
    pipe = PseudoLabelPipeline(cfg, device=cuda, vit_dtype='f32', max_points=25_000, clip_model_path='/nonexistent',
                               angle_mode='reference')          # the view angle from this host's numpy, like the oracle's
    assert pipe.clip.weights_source.startswith('synthetic')
    assert pipe.cls_key == 'clip_a_point_representation_of_a'
    fs, res = pipe.process_frame(pts, poses[1], poses[0], fnr=1)
    wd = cw.synthetic_vit_weights(0, **cw.VIT_B16)
    text = cw.synthetic_text_features(0, 24, 512)
    orc = OraclePipeline(wd, text, cfg['clip']['class_list'], cfg['clip']['class_mapping'], box_all_edges=False)   # the reference's rule
    o = orc.process_frame(pts, poses[1], poses[0])
    assert pipe.box_mode == 'reference'
    assert np.array_equal(np.sort(fs.ground_point_indices), o['ground_idx'])
    assert [int(c) for c in fs.cluster_ids] == [c for c, _ in o['dets']]
    for c, (_, idx) in enumerate(o['dets']):
        assert np.array_equal(fs.cluster_index(c), idx)
    assert np.array_equal(pipe.ground_plane(pipe.to_ref(pipe.upload(pts), fs.transform_to_ref),
                                            torch.from_numpy(o['ground_idx']).to(cuda)), o['plane'])
    assert np.array_equal(fs.valid, o['valid'])
    got_p = pipe.last_probs.cpu().numpy()
    assert got_p.shape == o['probs_clip'].shape and got_p.shape[0] == 4 * int(o['valid'].sum())
    err = np.abs(got_p - o['probs_clip']).max()
    print('clusters', len(o['dets']), 'valid', int(o['valid'].sum()), 'max |dp|', err)
    # parity mode (fp32 tower, reference box rule, this host's view angle): EVERY crop within north_star's 1e-3, every name equal
    assert err < 1e-3
    e = fs.cls[pipe.cls_key]
    rows = np.flatnonzero(fs.valid)
    assert [str(e['name'][r]) for r in rows] == list(o['names'])
    assert np.abs(np.array([e['final'][r] for r in rows], np.float64) - np.array(o['scores'], np.float64)).max() < 1e-3
    # box_mode='reference' (the default): the reference's boxes (closing hull edge dropped), same numpy on the same host
    assert np.abs(fs.boxes[rows] - o['boxes_ref']).max() <= 1e-9
    assert set(res.keys()) == {'boxes_lidar', 'name', 'score', 'moving'}
    assert res['boxes_lidar'].shape[1] == 7 and len(res['name']) == len(res['score']) == len(res['boxes_lidar'])
    # serialisation survives a round trip
    import pickle
    from vilgod_amd.frame_state import FrameState
    st = FrameState(1, poses[1], poses[0])
    st.sync(pickle.loads(pickle.dumps(fs.serialize)))
    assert np.array_equal(st.valid, fs.valid) and np.allclose(st.boxes[rows], fs.boxes[rows])


@pytest.mark.gpu
def test_pipeline_fp16_full_size_frame_runs(cuda):
    """BASELINE config 3 shape: 150k points, fp16 ViT, 4 views; sanity on outputs + per-stage times."""
    from vilgod_amd.pipeline import PseudoLabelPipeline
    pipe = PseudoLabelPipeline(device=cuda, vit_dtype='f16', max_points=160_000, clip_model_path='/nonexistent')
    poses = synthetic.make_poses(3)
    for f in range(3):
        fs, res = pipe.process_frame(synthetic.make_frame(f, 150_000), poses[f], poses[0], fnr=f, timing=True)
        print(f, 'clusters', fs.n_detections, 'valid', int(fs.valid.sum()), 'labelled', len(res['name']),
              {k: round(v * 1000, 2) for k, v in pipe.timings.items()})
        assert fs.n_detections > 20 and fs.valid.sum() > 10
        assert 0.35 < len(fs.ground_point_indices) / 150_000 < 0.6
        assert np.isfinite(res['boxes_lidar']).all() and (res['score'] > 0).all()


@pytest.mark.gpu
def test_sequence_pipeline_matches_oracle(cuda):
    """SURVEY 8f N1, the reference's default stage order on a short coherent sequence: ground -> entropy scores ->
    two-frame clustering + label transfer -> filters -> classification -> boxes; GPU pipeline vs the chained oracle.
    Exact: ground set, kept entropy indices, cluster membership, static flags, valid flags; names as in the 1-frame test."""
    from vilgod_amd.pipeline import PseudoLabelPipeline, default_preprocessor_cfg
    from oracle.pipeline_oracle import OraclePipeline
    cfg = default_preprocessor_cfg()
    frames, poses = synthetic.make_sequence(seed=2, n_frames=4, n_points=12_000, n_objects=8)
    pipe = PseudoLabelPipeline(cfg, device=cuda, vit_dtype='f32', max_points=25_000, clip_model_path='/nonexistent',
                               angle_mode='reference')
    ent_args = dict(n_neighbouring_frames=3, skip_frames=0)
    got = pipe.process_sequence(frames, poses, poses[0], entropy_args=ent_args, n_frames=2, seed=0)
    wd = cw.synthetic_vit_weights(0, **cw.VIT_B16)
    text = cw.synthetic_text_features(0, 24, 512)
    orc = OraclePipeline(wd, text, cfg['clip']['class_list'], cfg['clip']['class_mapping'])
    want = orc.process_sequence(frames, poses, poses[0], n_neighbouring_frames=3, skip_frames=0, n_frames=2, seed=0)
    n_static = n_moving = 0
    for (fs, res), o in zip(got, want):
        assert np.array_equal(np.sort(fs.ground_point_indices), o['ground_idx'])
        assert np.array_equal(fs.entropy_indices, o['entropy_indices'])
        assert np.allclose(fs.entropy_scores, o['entropy_scores'], rtol=0, atol=1e-12)
        assert [int(c) for c in fs.cluster_ids] == [c for c, _ in o['dets']]
        for c, (_, idx) in enumerate(o['dets']):
            assert np.array_equal(fs.cluster_index(c), idx)
        assert np.array_equal(fs.static, o['static'])
        assert np.array_equal(fs.valid, o['valid'])
        n_static += int(fs.static.sum()); n_moving += int((~fs.static).sum())
        rows = np.flatnonzero(fs.valid)
        if len(rows):
            e = fs.cls[pipe.cls_key]
            assert [str(e['name'][r]) for r in rows] == list(o['names'])
        ser = fs.serialize
        assert list(ser)[:4] == ['_detections', '_ground_point_indices', '_entropy_scores', '_entropy_indices']
    assert n_static > 0 and n_moving > 0
    # frames in flight (worker streams + shared, precomputed clustering rows) give the same states
    got2 = pipe.process_sequence(frames, poses, poses[0], entropy_args=ent_args, n_frames=2, seed=0, n_workers=2)
    for (fa, ra), (fb, rb) in zip(got, got2):
        assert np.array_equal(fa.index, fb.index) and np.array_equal(fa.seg_off, fb.seg_off)
        assert np.array_equal(fa.valid, fb.valid) and np.array_equal(fa.static, fb.static)
        assert np.array_equal(ra['name'], rb['name']) and np.allclose(ra['boxes_lidar'], rb['boxes_lidar'])


@pytest.mark.gpu
def test_dense_frames_six_views_in_flight_equal_sequential(cuda):
    """BASELINE config 5 shape (dense 200k-point frames, ~120 objects; 6 rendered views as in config 3): no CPU run at this size,
    so the check is a property -- processing the frames several at a time on worker streams (own handles per worker, shared
    weights) gives the same states and results as processing them one by one, and the outputs are well formed."""
    from vilgod_amd.pipeline import PseudoLabelPipeline
    pipe = PseudoLabelPipeline(device=cuda, vit_dtype='f16', n_views=6, max_points=210_000, clip_model_path='/nonexistent')
    poses = synthetic.make_poses(7)
    frames = [pipe.upload(synthetic.make_frame(40 + f, 200_000, n_objects=120)) for f in range(5)]
    pipe.new_sequence()
    seq = []
    for f in range(5):
        fs, res = pipe.process_frame(frames[f], poses[f + 1], poses[0], fnr=f)
        seq.append((fs, res, pipe.last_probs.cpu().numpy()))
    pipe.new_sequence()
    par = pipe.process_frames(frames, poses[1:6], poses[0], n_workers=4)
    assert len(par) == 5
    for (fa, ra, pa), (fb, rb, pb) in zip(seq, par):
        assert np.array_equal(np.sort(fa.ground_point_indices), np.sort(fb.ground_point_indices))
        assert np.array_equal(fa.index, fb.index) and np.array_equal(fa.seg_off, fb.seg_off) and np.array_equal(fa.valid, fb.valid)
        assert np.array_equal(pa, pb.cpu().numpy())                       # the same kernels on the same crops: bit-identical scores
        assert np.array_equal(ra['name'], rb['name']) and np.array_equal(ra['boxes_lidar'], rb['boxes_lidar'])
        assert pa.shape[0] == 6 * int(fa.valid.sum()) and np.allclose(pa.sum(1), 1.0, atol=1e-5)
        assert fa.n_detections > 60 and len(ra['name']) > 30
        assert np.isfinite(ra['boxes_lidar']).all() and (ra['boxes_lidar'][:, 3:6] > 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize('tower', ['complement', 'all'])
def test_cu_masked_streams_change_no_output(cuda, tower):
    """device.cu_reserve (vilgod_amd/streams.py): the frames' front stages on streams restricted to 2 CUs of every XCD, the ViT passes on a
    second stream per worker (the other CUs / all CUs).  Where a workgroup runs cannot change a number: states, scores and boxes are
    bit-identical to the unrestricted run, and the masks reach the runtime (the stream objects carry them)."""
    from vilgod_amd.pipeline import PseudoLabelPipeline
    from vilgod_amd.streams import MaskedStream
    poses = synthetic.make_poses(6)
    frames = [synthetic.make_frame(70 + f, 60_000, n_objects=30) for f in range(4)]
    ref = PseudoLabelPipeline(device=cuda, vit_dtype='f16', max_points=61_000, clip_model_path='/nonexistent')
    msk = PseudoLabelPipeline(device=cuda, vit_dtype='f16', max_points=61_000, clip_model_path='/nonexistent', clip=ref.clip,
                              cu_reserve=2, cu_tower=tower)
    ref.new_sequence(); msk.new_sequence()
    a = ref.process_frames([ref.upload(f) for f in frames], poses[1:5], poses[0], n_workers=3)
    b = msk.process_frames([msk.upload(f) for f in frames], poses[1:5], poses[0], n_workers=3)
    w = msk._workers[0]
    assert isinstance(w.stream, MaskedStream) and int(w.stream.cu_mask_words[0]) == 0xFFFF
    assert isinstance(w.vit_stream, MaskedStream) == (tower == 'complement')
    n_valid = 0
    for (fa, ra, pa), (fb, rb, pb) in zip(a, b):
        assert np.array_equal(np.sort(fa.ground_point_indices), np.sort(fb.ground_point_indices))
        assert np.array_equal(fa.index, fb.index) and np.array_equal(fa.seg_off, fb.seg_off) and np.array_equal(fa.valid, fb.valid)
        assert np.array_equal(pa.cpu().numpy(), pb.cpu().numpy())
        assert np.array_equal(ra['name'], rb['name']) and np.array_equal(ra['boxes_lidar'], rb['boxes_lidar'])
        n_valid += int(fa.valid.sum())
    assert n_valid > 20


@pytest.mark.gpu
def test_device_hierarchy_changes_no_output(cuda):
    """hierarchy='device' (csrc/hdbscan_device.hip, the default) against hierarchy='host' (csrc/hdbscan_tree.cpp; zero_shot_detector.py:248):
    labels and probabilities are bit-identical, so every state, score and box of frames in flight is."""
    from vilgod_amd.pipeline import PseudoLabelPipeline
    poses = synthetic.make_poses(6)
    frames = [synthetic.make_frame(70 + f, 60_000, n_objects=30) for f in range(4)]
    ref = PseudoLabelPipeline(device=cuda, vit_dtype='f16', max_points=61_000, clip_model_path='/nonexistent', hierarchy='host')
    dev = PseudoLabelPipeline(device=cuda, vit_dtype='f16', max_points=61_000, clip_model_path='/nonexistent', clip=ref.clip, hierarchy='device')
    assert ref.hierarchy == 'host' and dev.hierarchy == 'device' and dev.cluster_model.hierarchy == 'device'
    ref.new_sequence(); dev.new_sequence()
    a = ref.process_frames([ref.upload(f) for f in frames], poses[1:5], poses[0], n_workers=3)
    b = dev.process_frames([dev.upload(f) for f in frames], poses[1:5], poses[0], n_workers=3)
    n_valid = 0
    for (fa, ra, pa), (fb, rb, pb) in zip(a, b):
        assert np.array_equal(fa.index, fb.index) and np.array_equal(fa.seg_off, fb.seg_off) and np.array_equal(fa.valid, fb.valid)
        assert np.array_equal(pa.cpu().numpy(), pb.cpu().numpy())
        assert np.array_equal(ra['name'], rb['name']) and np.array_equal(ra['boxes_lidar'], rb['boxes_lidar'])
        n_valid += int(fa.valid.sum())
    assert n_valid > 20
    with pytest.raises(ValueError):
        PseudoLabelPipeline(device=cuda, max_points=1000, clip_model_path='/nonexistent', clip=ref.clip, hierarchy='gpu')


@pytest.mark.gpu
def test_round_robin_frames_with_replicated_ground_equal_one_rank(cuda):
    """SURVEY 8e, bench.py --ground-handoff replicate: two "ranks" (two pipeline objects here) take the frames of one sequence
    round-robin; each runs the stateful ground pass over ALL frames itself (process_frames(own=...)) and its own frames in full.
    Merged, the results equal one pipeline processing the whole sequence: ground sets, clusters, validity, boxes, names, scores --
    Patchwork++'s adaptive state reached every frame through the same passes in the same order on every rank."""
    from vilgod_amd.pipeline import PseudoLabelPipeline
    poses = synthetic.make_poses(9)
    frames = [synthetic.make_frame(70 + f, 40_000, n_objects=16) for f in range(7)]
    one = PseudoLabelPipeline(device=cuda, vit_dtype='f16', max_points=41_000, clip_model_path='/nonexistent')
    one.new_sequence()
    want = one.process_frames(frames, poses[1:8], poses[0], n_workers=3)
    merged = {}
    for r in range(2):
        p = PseudoLabelPipeline(device=cuda, vit_dtype='f16', max_points=41_000, clip_model_path='/nonexistent')
        p.new_sequence()
        mine = [g for g in range(7) if g % 2 == r]
        got = p.process_frames(frames, poses[1:8], poses[0], n_workers=3, own=mine)
        assert len(got) == len(mine)
        merged.update(dict(zip(mine, got)))
    assert sorted(merged) == list(range(7))
    for g in range(7):
        (fa, ra, pa), (fb, rb, pb) = want[g], merged[g]
        assert fa.fnr == fb.fnr == g
        assert np.array_equal(np.sort(fa.ground_point_indices), np.sort(fb.ground_point_indices))
        assert np.array_equal(fa.index, fb.index) and np.array_equal(fa.seg_off, fb.seg_off) and np.array_equal(fa.valid, fb.valid)
        assert np.array_equal(pa.cpu().numpy(), pb.cpu().numpy())
        assert np.array_equal(ra['name'], rb['name']) and np.array_equal(ra['boxes_lidar'], rb['boxes_lidar'])
    assert sum(len(r['name']) for _, r, _ in want) > 0


@pytest.mark.gpu
def test_round_robin_frames_with_relayed_ground_state_equal_one_rank(cuda):
    """bench.py --ground-handoff relay (round 6): two "ranks" (two pipeline objects on two threads, a pair of queues standing in for the
    point-to-point link) take the frames of one sequence round-robin; each runs ONLY its own ground passes, taking the Patchwork++ state
    behind frame g - 1 from that frame's owner before its pass of frame g and handing its own on behind it (process_frames(own=...,
    relay=(recv, send)); vilgod_amd/dist.py relay_recv_state / relay_send_state do the same over torch.distributed).  Merged, the results
    equal one pipeline processing the whole sequence -- ground sets, clusters, validity, boxes, names, scores."""
    import queue
    import threading
    from vilgod_amd.pipeline import PseudoLabelPipeline
    n = 7
    poses = synthetic.make_poses(n + 2)
    frames = [synthetic.make_frame(90 + f, 40_000, n_objects=16) for f in range(n)]
    one = PseudoLabelPipeline(device=cuda, vit_dtype='f16', max_points=41_000, clip_model_path='/nonexistent')
    one.new_sequence()
    want = one.process_frames(frames, poses[1:n + 1], poses[0], n_workers=3)
    link = [queue.Queue(), queue.Queue()]                 # link[r]: states addressed to rank r
    merged, errors = {}, []

    def rank(r):
        try:
            torch.cuda.set_device(cuda)
            p = PseudoLabelPipeline(device=cuda, vit_dtype='f16', max_points=41_000, clip_model_path='/nonexistent')
            p.new_sequence()
            mine = [g for g in range(n) if g % 2 == r]

            def recv(g):
                if g > 0:
                    p.ground_model.set_state(link[r].get(timeout=120))

            def send(g):
                if g < n - 1:
                    link[(g + 1) % 2].put(p.ground_model.export_state())
            got = p.process_frames(frames, poses[1:n + 1], poses[0], n_workers=2, own=mine, relay=(recv, send))
            assert len(got) == len(mine)
            merged.update(dict(zip(mine, got)))
        except BaseException as e:     # noqa: BLE001
            errors.append(e)
    th = [threading.Thread(target=rank, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    assert sorted(merged) == list(range(n))
    for g in range(n):
        (fa, ra, pa), (fb, rb, pb) = want[g], merged[g]
        assert fa.fnr == fb.fnr == g
        assert np.array_equal(np.sort(fa.ground_point_indices), np.sort(fb.ground_point_indices)), g
        assert np.array_equal(fa.index, fb.index) and np.array_equal(fa.seg_off, fb.seg_off) and np.array_equal(fa.valid, fb.valid)
        assert np.array_equal(pa.cpu().numpy(), pb.cpu().numpy())
        assert np.array_equal(ra['name'], rb['name']) and np.array_equal(ra['boxes_lidar'], rb['boxes_lidar'])


@pytest.mark.gpu
def test_f16_pipeline_agrees_with_f32_pipeline_on_150k_frames(cuda):
    """The benchmarked fp16 pipeline against the fp32 (oracle-pinned, test_pipeline_matches_oracle_20k) pipeline on three synthetic
    150k-point frames: everything before the ViT is the same code -> ground set, clusters, valid flags and boxes EQUAL; per-crop
    probabilities within 1.5e-3 (measured 1.35e-3; north_star's 1e-3 is the fp32 mode's bound, asserted in
    test_pipeline_matches_oracle_20k and test_integration.py); a view's top-1 class may flip only where the fp32 margin between its two best classes is inside
    that error bound, and a cluster's final name only through such a view.  Flips are counted and printed."""
    from vilgod_amd.pipeline import PseudoLabelPipeline
    from vilgod_amd.clip_wrapper import ClipWrapper
    p16 = PseudoLabelPipeline(device=cuda, vit_dtype='f16', max_points=160_000, clip_model_path='/nonexistent')
    p32 = PseudoLabelPipeline(device=cuda, vit_dtype='f32', max_points=160_000, clip_model_path='/nonexistent')
    poses = synthetic.make_poses(4)
    n_views = n_view_flips = n_clusters = n_name_flips = 0
    worst = 0.0
    p16.new_sequence(); p32.new_sequence()
    for f in range(3):
        pts = synthetic.make_frame(20 + f, 150_000)
        fa, ra = p16.process_frame(pts, poses[f + 1], poses[0], fnr=f)
        pa = p16.last_probs.cpu().numpy()
        fb, rb = p32.process_frame(pts, poses[f + 1], poses[0], fnr=f)
        pb = p32.last_probs.cpu().numpy()
        assert np.array_equal(fa.ground_point_indices, fb.ground_point_indices)
        assert np.array_equal(fa.index, fb.index) and np.array_equal(fa.seg_off, fb.seg_off) and np.array_equal(fa.valid, fb.valid)
        assert np.array_equal(fa.boxes, fb.boxes, equal_nan=True)
        assert pa.shape == pb.shape and pa.shape[0] == 4 * int(fa.valid.sum())
        err = np.abs(pa - pb).max()
        worst = max(worst, float(err))
        assert err <= 1.5e-3, err
        srt = np.sort(pb, axis=1)
        margin = srt[:, -1] - srt[:, -2]
        flip = pa.argmax(1) != pb.argmax(1)
        assert (margin[flip] <= 2 * 1.5e-3).all(), margin[flip]
        ea, eb = fa.cls[p16.cls_key], fb.cls[p32.cls_key]
        rows = np.flatnonzero(fa.valid)
        names_differ = np.array([str(ea['name'][r]) != str(eb['name'][r]) for r in rows])
        has_flip = flip.reshape(len(rows), 4).any(1)
        # the vote is a function of the per-view classes and scores: a name changes through a flipped view, or -- without one --
        # through a tie-break between equal vote counts whose mean scores lie within the error bound (at most one tolerated)
        n_views += len(flip); n_view_flips += int(flip.sum())
        n_clusters += len(rows); n_name_flips += int(names_differ.sum())
        assert names_differ.sum() <= max(1, int(has_flip.sum())) + 1
        # clusters without a flipped view: same name, final score (mean probability of the winning views) within the bound
        quiet = ~has_flip & ~names_differ
        fa_, fb_ = np.asarray(ea['final'])[rows].astype(np.float64), np.asarray(eb['final'])[rows].astype(np.float64)
        assert np.abs(fa_[quiet] - fb_[quiet]).max() <= 1.5e-3
        if not names_differ.any():
            assert np.array_equal(ra['name'], rb['name']) and np.array_equal(ra['boxes_lidar'], rb['boxes_lidar'])
    print(f'f16 vs f32 pipeline, 3 x 150k frames: max |dp| {worst:.2e}; top-1 flips {n_view_flips}/{n_views} views '
          f'(all inside the fp32 margin bound), final-name flips {n_name_flips}/{n_clusters} clusters')
