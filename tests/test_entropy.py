"""Entropy (PP) scores + two-frame clustering parity (the reference's default configuration).

CPU: the oracle (oracle/neighbors_oracle.py) against vectors produced by the REFERENCE's own numpy code
(tests/golden/entropy_golden.npz <- tests/golden/make_golden.py entropy: compute_ephe_score, filter_by_ephemeral_score,
and the sliding-window bookkeeping traced out of ZeroShotDetector.calculate_entropy_scores itself); the host mirror
(vilgod_amd/entropy.py, frame_state.static_from_entropy) against the same vectors.
GPU: the HIP kernels through the C ABI against the oracle -- neighbour counts, nearest-label transfer, sample keys,
5-D core distances / MST / labels: exact; entropy scores: 1e-12 (device log vs numpy log).
The neighbour searches themselves are parity-unpinned (CUDA-only third-party ops upstream, see the oracle's header).
"""
import os

import numpy as np
import pytest

from oracle import hdbscan_oracle as ho
from oracle import neighbors_oracle as no
from oracle import segment_oracle as so
from vilgod_amd import synthetic
from vilgod_amd.frame_state import static_from_entropy


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'entropy_golden.npz'))


def sequence(seed=3, n_frames=6, n_points=6000, n_objects=10):
    frames, poses = synthetic.make_sequence(seed=seed, n_frames=n_frames, n_points=n_points, n_objects=n_objects)
    X = []
    for f, p in zip(frames, poses):
        pr = so.apply_transform(f, np.linalg.inv(poses[0]) @ p)
        X.append(np.ascontiguousarray(pr[pr[:, 2] > 0.25]))
    return X


# ---------------------------------------------------------------------------------------------- CPU
def test_oracle_entropy_matches_reference(gold):
    for N in (2, 3, 5, 8, 15, 20):
        H = no.compute_ephe_score(gold[f'count_{N}'])
        assert np.array_equal(H, gold[f'H_{N}'], equal_nan=True), N


def test_oracle_and_mirror_static_flag_match_reference(gold):
    vals, seg = gold['eph_values'], gold['eph_seg']
    for (pct, mn), key in (((30, 0.5), 'eph_moving'), ((20, 0.7), 'eph_moving_20_07')):
        want = gold[key]
        got_o = np.array([no.filter_by_ephemeral_score(vals[seg[i]:seg[i + 1]], pct, mn) for i in range(len(seg) - 1)])
        assert np.array_equal(got_o, want)
        index = np.arange(len(vals), dtype=np.int32)               # identity packing
        static = static_from_entropy(vals, index, seg.astype(np.int32), pct, mn)
        assert np.array_equal(~static, want)


def test_window_matches_reference_bookkeeping(gold):
    from vilgod_amd.entropy import window
    for L, n in ((40, 15), (15, 15), (23, 7)):
        for fnr in range(L):
            for w in (window, no.window):
                frames, seek = w(fnr, L, n)
                assert frames[0] == gold[f'win_{L}_{n}_start'][fnr] and len(frames) == gold[f'win_{L}_{n}_len'][fnr]
                assert seek == gold[f'win_{L}_{n}_seek'][fnr] and frames[seek] == fnr


def test_oracle_ball_count_bruteforce():
    rng = np.random.default_rng(0)
    q = rng.uniform(-3, 3, size=(300, 3)).astype(np.float32)
    t = rng.uniform(-3, 3, size=(2000, 3)).astype(np.float32)
    r2 = np.float32(0.3) * np.float32(0.3)
    d2 = no.d2_f32(q[:, None, :], t[None, :, :])
    assert np.array_equal(no.ball_count(q, t, r2, 1000), (d2 < r2).sum(1))
    assert np.array_equal(no.ball_count(q, t, r2, 3), np.minimum((d2 < r2).sum(1), 3))
    idx, best = no.nearest(q, t, np.float32(0.2))
    dm = np.where(d2 <= np.float32(0.2), d2, np.inf)
    want = np.where(np.isinf(dm.min(1)), -1, dm.argmin(1))
    assert np.array_equal(idx, want)


def test_two_frame_input_structure():
    X = sequence()
    kept = no.entropy_scores_sequence(X, 4, 1)
    ent = [no.full_scores(len(x), s, i) for x, (s, i) in zip(X, kept)]
    seq, masks, rng = no.two_frame_input(X, ent, 2, 2, seed=0)
    assert rng == [2, 3] and seq.dtype == np.float32 and seq.shape[1] == 5
    assert np.all(seq[:masks[0].sum(), 4] == 0) and np.all(seq[masks[0].sum():, 4] == np.float32(0.1))
    assert no.two_frame_input(X, ent, 5, 2, seed=0)[2] == [4, 5]           # last frame pairs with its predecessor
    for m, f in zip(masks, rng):
        moving = ent[f] < 0.6
        assert m[~moving].sum() <= len(m) // 2                               # random half, minus the isolated points


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_ball_count_and_nearest_gpu(cuda):
    import torch
    from vilgod_amd.hdbscan import HDBSCAN
    X = sequence(seed=5, n_frames=3, n_points=20000, n_objects=14)
    m = HDBSCAN(min_cluster_size=15, cluster_selection_epsilon=0.15, max_points=50000)
    d = [torch.from_numpy(x).to(cuda) for x in X]
    m.grid(d[1])
    # (r2 = 0.5 and 1.5: 5^3 and 9^3 cells per query)
    for r2, cap in ((np.float32(0.3) * np.float32(0.3), 1000), (np.float32(0.2) * np.float32(0.2), 100), (np.float32(0.1), 4),
                    (np.float32(0.3) * np.float32(0.3), 5), (np.float32(0.5), 1000), (np.float32(1.5), 50)):
        for qi in (0, 1):
            got = m.ball_count(d[qi], r2, cap).cpu().numpy()
            assert np.array_equal(got, no.ball_count(X[qi], X[1], r2, cap)), (r2, cap, qi)
    gate = np.nextafter(np.float32(0.2), np.float32(0))
    idx, d2 = m.nearest(d[0], gate)
    widx, wd2 = no.nearest(X[0], X[1], gate)
    assert np.array_equal(idx.cpu().numpy(), widx)
    assert np.array_equal(d2.cpu().numpy(), wd2)
    # empty target / empty query
    m.grid(d[0][:0])
    assert int(m.ball_count(d[1], 0.09, 10).abs().sum()) == 0
    assert np.all(m.nearest(d[1], 0.2)[0].cpu().numpy() == -1)
    m.grid(d[0])
    assert m.ball_count(d[1][:0], 0.09, 10).numel() == 0


@pytest.mark.gpu
def test_entropy_scores_gpu(cuda, gold):
    import torch
    from vilgod_amd._lib import lib, ptr, stream_ptr, check
    for N in (2, 3, 5, 8, 15, 20):
        c = gold[f'count_{N}']
        dc = torch.from_numpy(np.ascontiguousarray(c.T.astype(np.int32))).to(cuda)
        H = torch.empty(c.shape[0], dtype=torch.float64, device=cuda)
        check(lib.vg_entropy_scores(ptr(dc), N, c.shape[0], -1, ptr(H), stream_ptr()))
        assert np.allclose(H.cpu().numpy(), gold[f'H_{N}'], rtol=0, atol=1e-12, equal_nan=True), N
    # the seek row: 1 is subtracted before the score
    c = gold['count_8'].copy()
    c[:, 2] += 1
    dc = torch.from_numpy(np.ascontiguousarray(c.T.astype(np.int32))).to(cuda)
    H = torch.empty(c.shape[0], dtype=torch.float64, device=cuda)
    check(lib.vg_entropy_scores(ptr(dc), 8, c.shape[0], 2, ptr(H), stream_ptr()))
    assert np.allclose(H.cpu().numpy(), gold['H_8'], rtol=0, atol=1e-12, equal_nan=True)


@pytest.mark.gpu
def test_subsample_keys_gpu(cuda):
    import torch
    from vilgod_amd._lib import lib, ptr, stream_ptr, check
    for seed, tag, n in ((0, 0, 1000), (7, 198, 4097), (123456789, 3, 1)):
        k = torch.empty(n, dtype=torch.int64, device=cuda)
        check(lib.vg_subsample_keys(seed, tag, n, ptr(k), stream_ptr()))
        assert np.array_equal(k.cpu().numpy(), no.subsample_keys(seed, tag, n))


@pytest.mark.gpu
@pytest.mark.parametrize('dim', [4, 5])
def test_mst_nd_gpu(cuda, dim):
    import torch
    from vilgod_amd.hdbscan import HDBSCAN
    X = sequence(seed=9, n_frames=4, n_points=3000, n_objects=6)
    kept = no.entropy_scores_sequence(X, 4, 1)
    ent = [no.full_scores(len(x), s, i) for x, (s, i) in zip(X, kept)]
    seq, _, _ = no.two_frame_input(X, ent, 1, 2, seed=0)
    seq = np.ascontiguousarray(seq[:1800, :dim])
    m = HDBSCAN(min_cluster_size=15, cluster_selection_epsilon=0.15, max_points=50000)
    lo, hi, w2, core2 = m.mst(torch.from_numpy(seq).to(cuda), want_core=True, dim=dim)
    wcore = ho.core_distances_sq(seq)
    assert np.array_equal(core2.cpu().numpy(), wcore)
    edges, ww2 = ho.mst_prim(seq, wcore)
    wlo, whi = np.minimum(edges[:, 0], edges[:, 1]), np.maximum(edges[:, 0], edges[:, 1])
    o_w = np.lexsort((whi, wlo, ww2))
    glo, ghi, gw2 = lo.cpu().numpy(), hi.cpu().numpy(), w2.cpu().numpy()
    o_g = np.lexsort((ghi, glo, gw2))
    assert np.array_equal(gw2[o_g], ww2[o_w]) and np.array_equal(glo[o_g], wlo[o_w]) and np.array_equal(ghi[o_g], whi[o_w])
    got = m.fit(seq)
    wl, wp = ho.fit(seq)
    assert np.array_equal(ho.canonical(got.labels_), ho.canonical(wl)) and np.array_equal(got.probabilities_, wp)


@pytest.mark.gpu
def test_sequence_entropy_and_two_frame_labels_gpu(cuda):
    import torch
    from vilgod_amd.hdbscan import HDBSCAN
    from vilgod_amd.entropy import EntropyScorer, TwoFrameClusterer, full_scores
    X = sequence(seed=3, n_frames=7, n_points=9000, n_objects=12)
    d = [torch.from_numpy(x).to(cuda) for x in X]
    m = HDBSCAN(min_cluster_size=15, cluster_selection_epsilon=0.15, max_points=50000)
    scorer = EntropyScorer(m, n_neighbouring_frames=5, skip_frames=1)
    H = scorer.score_sequence(d)
    want = no.entropy_scores_sequence(X, 5, 1)
    ent_g, ent_o = [], []
    for f in range(len(X)):
        vals, idx = scorer.reduce(H[f])
        wv, wi = want[f]
        # counts are exact; the score may differ in the last bits (log), so the < 0.9 cut may differ only AT the cut
        full_o = no.compute_ephe_score(no.count_neighbors([X[j] for j in no.window(f, len(X), 5)[0]], no.window(f, len(X), 5)[1], 1))
        assert np.allclose(H[f].cpu().numpy(), full_o, rtol=0, atol=1e-12)
        sure = np.abs(full_o - 0.9) > 1e-9
        assert np.array_equal(np.isin(np.arange(len(full_o)), idx)[sure], (full_o < 0.9)[sure])
        ent_g.append(full_scores(len(X[f]), vals, idx, device=cuda))
        ent_o.append(no.full_scores(len(X[f]), wv, wi))
        assert np.array_equal(ent_g[f].cpu().numpy(), ent_o[f])      # float32 after the cut: identical here
    two = TwoFrameClusterer(m, n_frames=2, seed=0)
    # ... and with the hierarchy stage on the device (the tree and the label transfer stay on the GPU: the pipeline's default)
    two_dev = TwoFrameClusterer(HDBSCAN(min_cluster_size=15, cluster_selection_epsilon=0.15, max_points=50000, hierarchy='device'), n_frames=2, seed=0)
    for fnr in (0, 3, len(X) - 1):
        seq_g = two.cluster_input(fnr, d, ent_g).cpu().numpy()
        seq_o, _, _ = no.two_frame_input(X, ent_o, fnr, 2, seed=0)
        assert np.array_equal(seq_g, seq_o)
        lab_g, prob_g = two.labels(fnr, d, ent_g)
        ls, ps = ho.fit(seq_o)
        lab_o, prob_o = no.knn_labels(X[fnr], seq_o, ls, ps)
        assert np.array_equal(ho.canonical(lab_g), ho.canonical(lab_o))
        assert np.array_equal(prob_g, prob_o)
        assert (lab_g >= 0).sum() > 100
        lab_d, prob_d = two_dev.labels(fnr, d, ent_g)
        assert lab_d.dtype == lab_g.dtype and np.array_equal(lab_d, lab_g) and np.array_equal(prob_d, prob_g)
