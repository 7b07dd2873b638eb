"""Degenerate inputs through the GPU path: empty / tiny / all-ground / no-cluster frames, capacity errors, tiny sequences.
The reference has no tests for these; its behaviour is: no detections -> empty result arrays (zero_shot_detector.py:846-857)."""
import numpy as np
import pytest

from vilgod_amd import synthetic


@pytest.fixture(scope='module')
def pipe(cuda):
    from vilgod_amd.pipeline import PseudoLabelPipeline, default_preprocessor_cfg
    return PseudoLabelPipeline(default_preprocessor_cfg(), device=cuda, vit_dtype='f16', max_points=30_000,
                               clip_model_path='/nonexistent')


def _empty_result(res):
    return res['boxes_lidar'].shape == (0, 7) and len(res['name']) == 0 and len(res['score']) == 0 and len(res['moving']) == 0


@pytest.mark.gpu
@pytest.mark.parametrize('n', [0, 1, 2, 14, 16, 40])
def test_tiny_frames(pipe, n):
    rng = np.random.default_rng(n)
    pts = np.zeros((n, 5), np.float32)
    pts[:, :2] = rng.uniform(-5, 5, size=(n, 2))
    pts[:, 2] = rng.uniform(0.5, 2.0, size=n)
    poses = synthetic.make_poses(2)
    pipe.new_sequence()
    fs, res = pipe.process_frame(pts, poses[1], poses[0], fnr=1)
    assert fs.n_points == n and _empty_result(res)
    assert fs.n_detections == 0 or (~fs.valid).all() or n >= 15
    ser = fs.serialize
    assert isinstance(ser['_detections'], list)


@pytest.mark.gpu
def test_all_ground_and_no_cluster_frames(pipe):
    rng = np.random.default_rng(1)
    n = 8000
    flat = np.zeros((n, 5), np.float32)                     # a perfectly flat ground disc: everything is ground
    r, th = rng.uniform(3, 40, n), rng.uniform(0, 2 * np.pi, n)
    flat[:, 0], flat[:, 1], flat[:, 2] = r * np.cos(th), r * np.sin(th), rng.normal(0, 0.01, n)
    poses = synthetic.make_poses(2)
    pipe.new_sequence()
    fs, res = pipe.process_frame(flat, poses[1], poses[0], fnr=1)
    assert len(fs.ground_point_indices) > 0.9 * n and _empty_result(res)
    sparse = np.zeros((600, 5), np.float32)                 # isolated points far apart (HDBSCAN is density-RELATIVE: may still cluster)
    sparse[:, :2] = rng.uniform(-60, 60, size=(600, 2))
    sparse[:, 2] = rng.uniform(0.6, 3.0, size=600)
    pipe.new_sequence()
    fs, res = pipe.process_frame(sparse, poses[1], poses[0], fnr=1)
    assert len(res['name']) == len(res['score']) == len(res['boxes_lidar']) <= fs.n_detections
    assert fs.boxes is None or np.isnan(fs.boxes[~fs.valid]).all()


@pytest.mark.gpu
def test_capacity_is_an_error_not_a_crash(pipe):
    from vilgod_amd._lib import VilgodHipError
    pts = synthetic.make_frame(0, 40_000, n_objects=8)      # pipeline was built for 30k points
    poses = synthetic.make_poses(2)
    pipe.new_sequence()
    with pytest.raises(VilgodHipError):
        pipe.process_frame(pts, poses[1], poses[0], fnr=1)
    pipe.new_sequence()
    fs, res = pipe.process_frame(synthetic.make_frame(0, 20_000, n_objects=8), poses[1], poses[0], fnr=1)   # still usable
    assert fs.n_detections > 0


@pytest.mark.gpu
def test_duplicate_points_and_exact_ties(cuda):
    """Many exactly coincident points and a regular lattice (all pair distances tie): the strict edge order keeps the MST
    unique, so GPU == oracle bit for bit even here."""
    import torch
    from oracle import hdbscan_oracle as ho
    from vilgod_amd.hdbscan import HDBSCAN
    g = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(4), indexing='ij'), -1).reshape(-1, 3).astype(np.float32) * 0.25
    X = np.concatenate([g, g[:40], g[:40], np.full((30, 3), 7.5, np.float32)])
    m = HDBSCAN(min_cluster_size=15, cluster_selection_epsilon=0.15, max_points=10_000)
    lo, hi, w2, core2 = m.mst(torch.from_numpy(X).to(cuda), want_core=True)
    wcore = ho.core_distances_sq(X)
    assert np.array_equal(core2.cpu().numpy(), wcore)
    edges, ww2 = ho.mst_prim(X, wcore)
    wlo, whi = np.minimum(edges[:, 0], edges[:, 1]), np.maximum(edges[:, 0], edges[:, 1])
    glo, ghi, gw2 = lo.cpu().numpy(), hi.cpu().numpy(), w2.cpu().numpy()
    og, ow = np.lexsort((ghi, glo, gw2)), np.lexsort((whi, wlo, ww2))
    assert np.array_equal(gw2[og], ww2[ow]) and np.array_equal(glo[og], wlo[ow]) and np.array_equal(ghi[og], whi[ow])
    got = m.fit(X)
    wl, wp = ho.fit(X)
    assert np.array_equal(ho.canonical(got.labels_), ho.canonical(wl)) and np.array_equal(got.probabilities_, wp)


@pytest.mark.gpu
def test_two_frame_sequence_is_the_minimum(pipe):
    frames, poses = synthetic.make_sequence(seed=4, n_frames=2, n_points=8000, n_objects=6)
    out = pipe.process_sequence(frames, poses, poses[0], entropy_args=dict(n_neighbouring_frames=15, skip_frames=0), n_frames=2)
    assert len(out) == 2
    for fs, res in out:
        assert fs.entropy_scores is not None and set(res) == {'boxes_lidar', 'name', 'score', 'moving'}


@pytest.mark.gpu
def test_cu_masked_stream_lifecycle_through_the_abi(cuda):
    """vg_stream_create_cu_mask / vg_stream_destroy (include/vilgod_hip.h, "execution resources"): a stream restricted to one CU of every XCD
    runs a kernel of the library (same result as on the default stream) and can be destroyed when nothing of torch's allocator refers to it;
    an all-zero mask and a NULL argument are rejected."""
    import ctypes
    import numpy as np
    import torch
    from vilgod_amd._lib import lib, ptr
    from vilgod_amd.streams import cu_mask_words, device_cu_count
    n_cu = device_cu_count(cuda)
    assert n_cu >= 16
    words = cu_mask_words(n_cu, 1, 'front')
    h = ctypes.c_void_p()
    assert lib.vg_stream_create_cu_mask(ctypes.byref(h), words.ctypes.data_as(ctypes.c_void_p), len(words)) == 0 and h.value
    pts = torch.randn(5000, 5, device=cuda)
    T = torch.eye(4, dtype=torch.float64, device=cuda)
    T[0, 3] = 2.5
    a, b = torch.empty_like(pts), torch.empty_like(pts)
    torch.cuda.synchronize()
    assert lib.vg_ref_transform(ptr(pts), 5000, 5, ptr(T), ptr(a), h) == 0
    assert lib.vg_ref_transform(ptr(pts), 5000, 5, ptr(T), ptr(b), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(a, b) and float((a[:, 0] - pts[:, 0]).mean()) == pytest.approx(2.5, abs=1e-5)
    assert lib.vg_stream_destroy(h) == 0
    zero = np.zeros(len(words), np.uint32)
    assert lib.vg_stream_create_cu_mask(ctypes.byref(h), zero.ctypes.data_as(ctypes.c_void_p), len(zero)) == 1
    assert lib.vg_stream_destroy(None) == 1
