"""Generates tests/golden/*.npz by running the REFERENCE's own Python (stub-imported from
/root/reference, see oracle/refstubs.py) on seeded inputs.  Run only in the build container:

    python tests/golden/make_golden.py [render] [vit] [detect] [box]

The fixtures are data (inputs + the reference's outputs); no reference source is stored.
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refstubs  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def render_cases():
    """Seeded clusters: sizes x azimuths, roughly object shaped, float32 in the ego frame."""
    rng = np.random.default_rng(20241220)
    cases = []
    for P in [10, 50, 500, 5000]:
        for az, rg in [(0.3, 8.0), (2.0, 25.0), (-2.5, 55.0)]:
            c = np.array([rg * np.cos(az), rg * np.sin(az), 0.8])
            ext = rng.uniform([0.3, 0.3, 0.5], [2.5, 1.2, 1.0])
            pts = (rng.normal(size=(P, 3)) * ext + c).astype(np.float32)
            cases.append(pts)
    return cases


def make_render():
    refstubs.install()
    from src.utils import mv_utils, pointcloud_utils
    proj = mv_utils.RealisticProjection(refstubs.projection_cfg())
    cases = render_cases()
    store = {}
    hashes = []
    for i, pts in enumerate(cases):
        origin = pointcloud_utils.transform_cluster_points_to_origin(pts)          # pointcloud_utils.py:390
        t = torch.from_numpy(origin).float().unsqueeze(0)                           # zero_shot_detector.py:394
        img = proj.get_img(t).detach()                                              # mv_utils.py:173
        assert torch.equal(img[:, 0], img[:, 1]) and torch.equal(img[:, 0], img[:, 2])
        big = torch.nn.functional.interpolate(img, size=(224, 224), mode='bilinear', align_corners=True)
        big = big.permute(0, 3, 2, 1).detach().cpu().numpy()                        # zero_shot_detector.py:405-408
        u8 = np.stack([np.uint8(b * 255) for b in big])                             # :409
        assert (u8[..., 0] == u8[..., 1]).all() and (u8[..., 0] == u8[..., 2]).all()
        hashes.append([sha(origin), sha(img[:, 0].numpy()), sha(u8[..., 0])])
        store[f'pts_{i}'] = pts
        store[f'originf32_{i}'] = origin.astype(np.float32)
        if i % 3 == 0:   # keep full images for 4 of the 12 cases, hashes for all
            store[f'origin_{i}'] = origin
            store[f'img_{i}'] = img[:, 0].numpy()
            store[f'u8_{i}'] = u8[..., 0]
    store['hashes'] = np.array(hashes)
    store['rot_mat'] = proj.rot_mat.numpy()
    np.savez_compressed(os.path.join(OUT, 'render_golden.npz'), **store)
    print('render_golden.npz', len(cases), 'cases')


def make_render_small():
    """Clusters below ~50 points (valid clusters start at 10, waymo.yaml:16-30).  torch-CPU multiplies such small inputs
    without FMA while the HIP renderer (like a GPU BLAS) uses an FMA chain, so the view-rotated points can differ in the last
    bit and a point may cross a ceil() boundary.  To pin everything DOWNSTREAM of that 3x3 product bit for bit, the reference's
    own `point_transform` output is frozen per view together with the images the reference makes of it
    -> tests/golden/render_small_golden.npz; the test renders the frozen view points through an identity view."""
    refstubs.install()
    from src.utils import mv_utils, pointcloud_utils
    proj = mv_utils.RealisticProjection(refstubs.projection_cfg())
    rng = np.random.default_rng(20250103)
    store, hashes, n = {}, [], 0
    for P in [10, 11, 13, 16, 23, 31, 40, 49]:
        for az, rg in [(0.9, 6.0), (-1.7, 38.0)]:
            c = np.array([rg * np.cos(az), rg * np.sin(az), 0.6])
            ext = rng.uniform([0.2, 0.2, 0.3], [1.8, 0.9, 0.9])
            pts = (rng.normal(size=(P, 3)) * ext + c).astype(np.float32)
            origin = pointcloud_utils.transform_cluster_points_to_origin(pts)
            t = torch.from_numpy(origin).float().unsqueeze(0)
            v = proj.translation.shape[0]
            view_pts = proj.point_transform(points=torch.repeat_interleave(t, v, dim=0), rot_mat=proj.rot_mat.repeat(1, 1, 1))
            img = proj.get_img(t).detach()
            big = torch.nn.functional.interpolate(img, size=(224, 224), mode='bilinear', align_corners=True)
            big = big.permute(0, 3, 2, 1).detach().cpu().numpy()
            u8 = np.stack([np.uint8(b * 255) for b in big])
            store[f'pts_{n}'] = pts
            store[f'viewpts_{n}'] = view_pts.numpy()
            hashes.append([sha(img[:, 0].numpy()), sha(u8[..., 0])])
            if n % 4 == 0:
                store[f'img_{n}'] = img[:, 0].numpy()
                store[f'u8_{n}'] = u8[..., 0]
            n += 1
    store['hashes'] = np.array(hashes)
    np.savez_compressed(os.path.join(OUT, 'render_small_golden.npz'), **store)
    print('render_small_golden.npz', n, 'cases')


def make_vit():
    from vilgod_amd import clip_weights as cw
    m = refstubs.load_clip_model_py()
    cfg = dict(width=128, layers=2, heads=2, patch=16, resolution=64, output_dim=32)
    wd = cw.synthetic_vit_weights(11, **cfg)
    ref = m.VisionTransformer(cfg['resolution'], cfg['patch'], cfg['width'], cfg['layers'], cfg['heads'],
                              cfg['output_dim'])
    ref.load_state_dict(wd)
    ref.eval()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 3, 64, 64, generator=g)
    with torch.no_grad():
        y = ref(x)
    store = dict(x=x.numpy(), y=y.numpy(), seed=np.array(11), cfg=np.array([cfg[k] for k in
                 ('width', 'layers', 'heads', 'patch', 'resolution', 'output_dim')]))
    # full ViT-B/16: weights are re-derived from the seed on both sides; store input seed + output
    wd = cw.synthetic_vit_weights(0, **cw.VIT_B16)
    ref = m.VisionTransformer(224, 16, 768, 12, 12, 512)
    ref.load_state_dict(wd)
    ref.eval()
    xb = torch.randn(3, 3, 224, 224, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        yb = ref(xb)
    store['xb_seed'] = np.array(6)
    store['yb'] = yb.numpy()
    np.savez_compressed(os.path.join(OUT, 'vit_golden.npz'), **store)
    print('vit_golden.npz')


def make_detect():
    """B3/C1/D10/E1/F1/T1: the reference's LidarFrame / Detection / cluster_utils / pointcloud_utils run
    unchanged (stub imports) on a small seeded frame -> tests/golden/detect_golden.pkl."""
    import logging
    import pickle
    from functools import partial
    refstubs.install()
    from src.vilgod.lidar_frame import LidarFrame
    from src.utils import cluster_utils, pointcloud_utils
    from vilgod_amd import synthetic
    from oracle import hdbscan_oracle as ho, patchworkpp as opw
    cfg = refstubs.AttrDict(preprocessor=dict(clustering=dict(propability_threshold=0.3,
                            entropy_score_filter=dict(percentile=30, min_percentile_pp_score=0.5))))
    pts = synthetic.make_frame(7, 12000, n_objects=10)
    poses = synthetic.make_poses(3, step=0.5, seed=1)
    pose, ref_pose = poses[2], poses[0]
    frame = LidarFrame('seq', 2, pts, dict(gt_names=[], moving=[]), pose, ref_pose, cfg, logging.getLogger('g'))
    params = opw.Parameters()
    params.min_range = 1.5
    gidx = opw.mask_ground_points(pts, opw.patchworkpp(params), 1.723)
    frame.update_ground_indices(gidx)
    X = frame.points_ref_wo_ground[..., :3]
    labels, probs = ho.fit(X)
    frame.generate_detections(labels.copy(), probs.copy(), assign_gt=False, entropy_scores=frame.entropy_scores)
    plane = np.array([0.01, -0.005, 1.0, 0.02])
    filters = []
    for name, args in [('filter_by_number_points', dict(logic='and', required=True, min_points=10)),
                       ('filter_by_height', dict(logic='and', required=True, min_height=0.3, max_height=6)),
                       ('filter_by_plane_distance', dict(logic='and', required=True, max_min_height=1.0, min_max_height=0.5))]:
        filters.append([partial(getattr(cluster_utils, name), **args), name, args.get('logic'), args.get('required', False)])
    boxes, stats = [], []
    for det in frame.detections:
        det.filter(filters, plane_model=plane)                                            # objects.py:158
        cp = det.cluster_points
        corners, rz, area = pointcloud_utils.minimum_bounding_rectangle(cp[:, :2])        # pointcloud_utils.py:309
        l = np.linalg.norm(corners[0] - corners[1])
        w = np.linalg.norm(corners[0] - corners[-1])
        c = (corners[0] + corners[2]) / 2
        if w > l:
            l, w = w, l
            rz += np.pi / 2
        height = cp[:, 2].max() - cp[:, 2].min()
        box = np.array([c[0], c[1], cp[:, 2].min() + height / 2, l, w, height + 0.3, rz])  # zero_shot_detector.py:452-460
        det.update_bounding_box(box)
        boxes.append(box)
    # voting with engineered ties (lidar_frame.py:260-291)
    rng = np.random.default_rng(3)
    n_det = len(frame.detections)
    fine = ['car', 'truck', 'pedestrian', 'cyclist', 'pole', 'tree']
    mapping = dict(car='Vehicle', truck='Vehicle', pedestrian='Pedestrian', cyclist='Cyclist', pole='Background', tree='Background')
    detailed = rng.choice(fine, size=(n_det, 4))
    detailed[0] = ['car', 'truck', 'pedestrian', 'pedestrian']          # 2-2 tie
    detailed[1] = ['car', 'pole', 'cyclist', 'pedestrian']              # 1-1-1-1 tie
    names = np.vectorize(mapping.get)(detailed)
    scores = rng.uniform(0.1, 0.9, size=(n_det, 4)).astype(np.float32)
    key = 'clip_a_point_representation_of_a'
    upd = [d.valid for d in frame.detections]
    nv = sum(upd)
    frame.update_object_classes(names[:nv], detailed[:nv], scores[:nv], upd, key=key, aggregation='voting')
    ser = frame.serialize                                                                  # lidar_frame.py:41-59
    ego_boxes = pointcloud_utils.apply_transform(np.array(boxes), frame.transform_to_ego, box=True)   # zero_shot_detector.py:847
    out = dict(points=pts, pose=pose, ref_pose=ref_pose, ground_idx=gidx, labels=labels, probs=probs, plane=plane,
               points_ref=frame.points_ref, transform_to_ref=frame.transform_to_ref, transform_to_ego=frame.transform_to_ego,
               det_ids=[d.cluster_id for d in frame.detections],
               det_index=[d.cluster_points_index for d in frame.detections],
               det_center=[d.cluster_center for d in frame.detections], det_median=[d.cluster_mass_center for d in frame.detections],
               det_height=[d.height for d in frame.detections], valid=[bool(d.valid) for d in frame.detections],
               boxes_ref=np.array(boxes), boxes_ego=ego_boxes, vote_names=names, vote_detailed=detailed, vote_scores=scores,
               vote_key=key, serialized=ser)
    with open(os.path.join(OUT, 'detect_golden.pkl'), 'wb') as f:
        pickle.dump(out, f)
    print('detect_golden.pkl', n_det, 'detections,', nv, 'valid')


def make_entropy():
    """N1: the reference's pure-numpy pieces run unchanged -- pointcloud_utils.compute_ephe_score on seeded count
    matrices (several column counts), cluster_utils.filter_by_ephemeral_score on seeded float32 score vectors, and the
    sliding-window bookkeeping of ZeroShotDetector.calculate_entropy_scores (which frames are in the buffer and where the
    query sits) traced by running the method itself with `pointcloud_utils.calculate_entropy_scores` replaced by a
    recorder -> tests/golden/entropy_golden.npz."""
    refstubs.install()
    import types
    import torch
    from src.utils import cluster_utils, pointcloud_utils
    rng = np.random.default_rng(11)
    out = {}
    for N in (2, 3, 5, 8, 15, 20):
        c = rng.integers(0, 1001, size=(400, N)).astype(np.int64)
        c[rng.uniform(size=c.shape) < 0.25] = 0
        c[:5] = 0                                            # all-zero rows (no neighbour anywhere)
        c[5:10] = 1000
        out[f'count_{N}'] = c
        out[f'H_{N}'] = pointcloud_utils.compute_ephe_score(c)
    sc = [rng.uniform(0.2, 1.0, size=int(n)).astype(np.float32) for n in rng.integers(10, 300, size=64)]
    for v in sc[:16]:
        v[rng.uniform(size=len(v)) < 0.5] = 1.0              # stored scores >= 0.9 read back as 1.0
    out['eph_values'] = np.concatenate(sc)
    out['eph_seg'] = np.r_[0, np.cumsum([len(v) for v in sc])].astype(np.int64)
    out['eph_moving'] = np.array([cluster_utils.filter_by_ephemeral_score(v, percentile=30, min_percentile_pp_score=0.5) for v in sc])
    out['eph_moving_20_07'] = np.array([cluster_utils.filter_by_ephemeral_score(v, percentile=20, min_percentile_pp_score=0.7) for v in sc])
    # ---- window bookkeeping: run the reference method with a recording scorer ----
    from src.vilgod import zero_shot_detector as zsd
    trace = {}
    for L, n in ((40, 15), (15, 15), (23, 7)):
        rec = []

        class LF:
            def __init__(self, i):
                self.fnr = i
                self._entropy_scores = None
                self.points_ref_wo_ground = np.full((3, 5), float(i), np.float32)
            entropy_scores = None

            def update_entropy_scores(self, s, i):
                pass

        def fake(frame_buffer, seek, **kw):
            rec.append(([int(t[0, 0].item()) for t in frame_buffer], int(seek)))
            return np.ones(3)

        self_ = types.SimpleNamespace(lidar_frame_list=[LF(i) for i in range(L)], lenght=L,
                                      reset_progress_bar=lambda *a, **k: None,
                                      progress_bar=types.SimpleNamespace(update=lambda *a, **k: None),
                                      sync_lidar_frames=lambda: None)
        orig, orig_cuda = pointcloud_utils.calculate_entropy_scores, torch.Tensor.cuda
        pointcloud_utils.calculate_entropy_scores = fake
        torch.Tensor.cuda = lambda self, *a, **k: self
        empty = torch.cuda.empty_cache
        torch.cuda.empty_cache = lambda: None
        try:
            zsd.ZeroShotDetector.calculate_entropy_scores(self_, n)
        finally:
            pointcloud_utils.calculate_entropy_scores, torch.Tensor.cuda, torch.cuda.empty_cache = orig, orig_cuda, empty
        out[f'win_{L}_{n}_start'] = np.array([r[0][0] for r in rec])
        out[f'win_{L}_{n}_len'] = np.array([len(r[0]) for r in rec])
        out[f'win_{L}_{n}_seek'] = np.array([r[1] for r in rec])
        assert all(r[0] == list(range(r[0][0], r[0][0] + len(r[0]))) for r in rec)
    np.savez_compressed(os.path.join(OUT, 'entropy_golden.npz'), **out)
    print('entropy_golden.npz', sorted(out)[:6], '...')


def make_text():
    """D8: the reference's tokenizer (third_party/CLIP/clip/simple_tokenizer.py, with `ftfy.fix_text` stubbed to identity: the
    prompts are ASCII) and the reference's CLIP.encode_text (clip/model.py) on a small seeded text tower
    (clip_weights.synthetic_text_weights) -> tests/golden/text_golden.npz (token ids + text features)."""
    import types
    import importlib.util
    import torch
    from vilgod_amd import clip_weights as cw
    from vilgod_amd.pipeline import default_preprocessor_cfg
    sys.modules.setdefault('ftfy', types.SimpleNamespace(fix_text=lambda t: t))
    spec = importlib.util.spec_from_file_location('_ref_simple_tokenizer', f'{refstubs.REF}/third_party/CLIP/clip/simple_tokenizer.py')
    st = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(st)
    tk = st.SimpleTokenizer()
    cfg = default_preprocessor_cfg()['clip']
    prompts = [cfg['prompt_template'].format(c) for c in cfg['class_list']]
    extra = ["A photo of a   Dog's toy, 3 cats & 2.5 birds!", 'hello-world_foo BAR', "it's we're they'll"]
    texts = prompts + extra
    tokens = np.zeros((len(texts), 77), np.int64)
    for i, t in enumerate(texts):
        ids = [tk.encoder['<|startoftext|>']] + tk.encode(t) + [tk.encoder['<|endoftext|>']]
        tokens[i, :len(ids)] = ids
    m = refstubs.load_clip_model_py()
    width, layers, embed = 64, 2, 32
    model = m.CLIP(embed_dim=embed, image_resolution=32, vision_layers=1, vision_width=64, vision_patch_size=16, context_length=77,
                   vocab_size=49408, transformer_width=width, transformer_heads=1, transformer_layers=layers)
    sd = cw.synthetic_text_weights(0, width=width, layers=layers, embed=embed)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith('visual.') or k == 'logit_scale' for k in missing), (missing, unexpected)
    model = model.float().eval()
    with torch.no_grad():
        feat = model.encode_text(torch.from_numpy(tokens)).numpy()
    np.savez_compressed(os.path.join(OUT, 'text_golden.npz'), texts=np.array(texts), tokens=tokens, features=feat,
                        width=width, layers=layers, embed=embed)
    print('text_golden.npz', tokens.shape, feat.shape, 'longest prompt', int((tokens > 0).sum(1).max()), 'tokens')


class _RefKalman:
    """Stand-in for the absent `filterpy.kalman.KalmanFilter` so that the reference's Track class can run here: filterpy's
    published predict/update (kalman_filter.py: x = Fx, P = FPF' + Q; y = z - Hx, S = HPH' + R, K = PH'S^-1, x += Ky,
    P = (I-KH)P(I-KH)' + KRK').  Written independently of vilgod_amd/tracking.py."""

    def __init__(self, dim_x, dim_z):
        self.x = np.zeros((dim_x, 1)); self.P = np.eye(dim_x); self.Q = np.eye(dim_x); self.F = np.eye(dim_x)
        self.H = np.zeros((dim_z, dim_x)); self.R = np.eye(dim_z); self._I = np.eye(dim_x)

    def predict(self):
        self.x = np.dot(self.F, self.x)
        self.P = np.dot(np.dot(self.F, self.P), self.F.T) + self.Q

    def update(self, z):
        z = np.asarray(z)
        y = z - np.dot(self.H, self.x)
        PHT = np.dot(self.P, self.H.T)
        S = np.dot(self.H, PHT) + self.R
        K = np.dot(PHT, np.linalg.inv(S))
        self.x = self.x + np.dot(K, y)
        I_KH = self._I - np.dot(K, self.H)
        self.P = np.dot(np.dot(I_KH, self.P), I_KH.T) + np.dot(np.dot(K, self.R), K.T)


def _ref_q_discrete_white_noise(dim, dt=1., var=1., block_size=1, order_by_dim=True):
    assert dim == 4 and block_size == 1
    Q = [[(dt**6)/36, (dt**5)/12, (dt**4)/6, (dt**3)/6],
         [(dt**5)/12, (dt**4)/4, (dt**3)/2, (dt**2)/2],
         [(dt**4)/6, (dt**3)/2, dt**2, dt],
         [(dt**3)/6, (dt**2)/2, dt, 1.]]
    return np.array(Q) * var


def sat_iou3d(a, b):
    """Stand-in for pcdet's boxes_iou3d_gpu: only `iou > 0` is ever tested upstream (zero_shot_detector.py:739), so the
    values are 1.0 where the rotated BEV rectangles overlap (separating-axis test) and the z ranges overlap, else 0."""
    a, b = a.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64)

    def corners(bx):
        c, s = np.cos(bx[6]), np.sin(bx[6])
        dx, dy = bx[3] / 2, bx[4] / 2
        loc = np.array([[dx, dy], [-dx, dy], [-dx, -dy], [dx, -dy]])
        return loc @ np.array([[c, s], [-s, c]]) + bx[:2]

    def overlap(p, q):
        for poly in (p, q):
            for i in range(4):
                e = poly[(i + 1) % 4] - poly[i]
                ax = np.array([-e[1], e[0]])
                pp, qq = p @ ax, q @ ax
                if pp.max() <= qq.min() or qq.max() <= pp.min():
                    return False
        return True

    res = np.zeros((len(a), len(b)), np.float32)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            zo = min(x[2] + x[5] / 2, y[2] + y[5] / 2) - max(x[2] - x[5] / 2, y[2] - y[5] / 2)
            res[i, j] = 1.0 if zo > 0 and overlap(corners(x), corners(y)) else 0.0
    return torch.from_numpy(res)


def track_scene():
    """The seeded scene of the N2 goldens: a short coherent sequence, clustered per frame by the oracle."""
    from vilgod_amd import synthetic
    from oracle import hdbscan_oracle as ho, segment_oracle as so, neighbors_oracle as no
    frames, poses = synthetic.make_sequence(seed=11, n_frames=18, n_points=5000, n_objects=9, moving_frac=0.6)
    X, clusters = [], []
    for f, p in zip(frames, poses):
        pr = so.apply_transform(f, np.linalg.inv(poses[0]) @ p)
        X.append(np.ascontiguousarray(pr[pr[:, 2] > 0.25]))
    kept = no.entropy_scores_sequence(X, 5, 1)
    ent = [no.full_scores(len(x), s, i) for x, (s, i) in zip(X, kept)]
    for fnr, x in enumerate(X):
        labels, probs = ho.fit(x[:, :3])
        dets = so.generate_detections(labels, probs)
        # drop some detections so that tracks see misses, predictions, re-acquisitions and ends
        dets = [(c, i) for k, (c, i) in enumerate(dets) if ((fnr * 7 + k * 3) % 11 >= 2 or fnr < 2) and not (k == 1 and 6 <= fnr <= 11)]
        clusters.append(dets)
    return frames, poses, X, ent, clusters


def make_track():
    """N2: the reference's Tracker / Track / Detection classes (src/vilgod/tracker.py, src/dataclass/objects.py,
    src/utils/tracking_utils.py) run unchanged on the seeded scene, with `filterpy` replaced by the stand-in above
    -> tests/golden/track_golden.pkl (tracks as lists of (frame, prediction flag, source frame, source cluster id))."""
    import pickle
    refstubs.install()
    sys.modules['filterpy.kalman'].KalmanFilter = _RefKalman
    sys.modules['filterpy.common'].Q_discrete_white_noise = _ref_q_discrete_white_noise
    from src.dataclass.objects import Detection
    from src.vilgod.tracker import Tracker
    from oracle import neighbors_oracle as no
    frames, poses, X, ent, clusters = track_scene()
    cfg = refstubs.AttrDict(mode='cluster_center', assignment=refstubs.AttrDict(method='assign_detections_greedy', max_distance=1.0),
                            min_length=5, max_missed=3, min_distance_dynamic=2.0)
    tracker = Tracker('seq', cfg)
    dets_per_frame = []
    for fnr, (x, dets) in enumerate(zip(X, clusters)):
        objs = []
        for cid, idx in dets:
            d = Detection(cluster_id=cid, cluster_points=x[idx], cluster_points_index=idx)
            d.static = not no.filter_by_ephemeral_score(ent[fnr][idx])
            d._fnr, d._cid = fnr, cid
            objs.append(d)
        dets_per_frame.append(objs)
        tracker.next(objs, fnr)
    tracker.finish()
    tracks = []
    for t in tracker.tracks:
        tracks.append(dict(frames=list(t.frame_indices), entries=[(bool(d.track_prediction), d._fnr, d._cid) for d in t.detections],
                           kf_x=np.array(t.kf.x), kf_P=np.array(t.kf.P)))
    out = dict(clusters=[[(int(c), np.asarray(i)) for c, i in dets] for dets in clusters],
               static=[[bool(d.static) for d in objs] for objs in dets_per_frame], tracks=tracks)

    # ---- fit_bounding_boxes_simple, track branch (zero_shot_detector.py:422-684), run unbound on a stand-in `self` ----
    import types
    import torch
    from src.vilgod import zero_shot_detector as zsd
    T_ref = [np.linalg.inv(poses[0]) @ p for p in poses]
    lfl = [types.SimpleNamespace(fnr=f, detections=dets_per_frame[f], transform_to_ego=np.linalg.inv(poses[f]) @ poses[0])
           for f in range(len(X))]
    class_names = ['Vehicle', 'Pedestrian', 'Cyclist']
    self_ = types.SimpleNamespace(tracker=tracker, lidar_frame_list=lfl, dataset=types.SimpleNamespace(class_names=class_names),
                                  reset_progress_bar=lambda *a, **k: None, progress_bar=types.SimpleNamespace(update=lambda *a, **k: None),
                                  sync_lidar_frames=lambda: None)
    zsd.ZeroShotDetector.fit_bounding_boxes_simple(self_, {'name': 'minimum_bounding_rectangle', 'args': {}}, force=True,
                                                   valid_only=True, fg_only=False, classification_key='clip')
    for t, rec in zip(tracker.tracks, tracks):
        rec['boxes_fit'] = np.array([d.bounding_box for d in t.detections])
        rec['static_track_fit'] = [d.static_track for d in t.detections]
        rec['track_static_fit'] = bool(t.static)

    # ---- propagate_labels (zero_shot_detector.py:686-824) on seeded per-detection classification results ----
    rng = np.random.default_rng(5)
    names_all = ['Vehicle', 'Pedestrian', 'Cyclist', 'Background', 'Sign']
    cls_in = []
    for f, objs in enumerate(dets_per_frame):
        row = []
        for d in objs:
            h = float(d.cluster_points[:, 2].max() - d.cluster_points[:, 2].min())
            base = 0 if h < 1.9 and len(d.cluster_points) > 150 else (1 if h >= 1.5 else 3)
            name = names_all[base] if rng.uniform() < 0.7 else names_all[int(rng.integers(0, 5))]
            score = np.float32(rng.uniform(0.2, 0.95))
            d.add_object_entry('object_class', 'clip', name)
            d.add_object_entry('object_class_score', 'clip', score)
            row.append((name, float(score)))
        cls_in.append(row)
    out['cls_in'] = cls_in

    zsd.iou3d_nms_utils = types.SimpleNamespace(boxes_iou3d_gpu=sat_iou3d)
    zsd.ZeroShotDetector.propagate_labels(self_, classification_key='clip')
    for t, rec in zip(tracker.tracks, tracks):
        rec['track_static'] = bool(t.static); rec['track_valid'] = bool(t.valid); rec['class_label'] = t.class_label
        rec['corrected'] = bool(t.class_label_corrected); rec['corrected_by_size'] = bool(t.class_label_corrected_by_size)
        rec['boxes_final'] = np.array([d.bounding_box for d in t.detections])
        rec['static_track_final'] = [d.static_track for d in t.detections]
    out['final'] = [[dict(valid=bool(d.valid), name=d.object_class['clip'], score=float(d.object_class_score['clip']),
                          box=None if d.bounding_box is None else np.array(d.bounding_box), static_track=d.static_track)
                     for d in objs] for objs in dets_per_frame]
    with open(os.path.join(OUT, 'track_golden.pkl'), 'wb') as f:
        pickle.dump(out, f)
    print('track_golden.pkl', len(tracks), 'tracks, lengths', sorted(len(t['frames']) for t in tracks)[-8:],
          'predictions', sum(e[0] for t in tracks for e in t['entries']),
          '| moving tracks after fit', sum(not t['track_static_fit'] for t in tracks), 'after propagate', sum(not t['track_static'] for t in tracks),
          '| invalid tracks', sum(not t['track_valid'] for t in tracks), '| labels', [t['class_label'] for t in tracks])



# ---- N3 / N4: dataset adapters and the evaluation filters ---------------------------------------------------------------------
def _clip_polygon_area(pa, pb):
    """Stand-in for the area part of pcdet's CUDA IoU op: Sutherland-Hodgman clipping of convex polygon pa by pb (both CCW)."""
    out = [tuple(p) for p in pa]
    for i in range(len(pb)):
        a, b = pb[i], pb[(i + 1) % len(pb)]
        inp, out = out, []
        if not inp:
            break
        side = lambda q: (b[0] - a[0]) * (q[1] - a[1]) - (b[1] - a[1]) * (q[0] - a[0])
        for j in range(len(inp)):
            cur, prv = inp[j], inp[j - 1]
            sc, sp = side(cur), side(prv)
            if sc >= 0:
                if sp < 0:
                    t = sp / (sp - sc)
                    out.append((prv[0] + t * (cur[0] - prv[0]), prv[1] + t * (cur[1] - prv[1])))
                out.append(cur)
            elif sp >= 0:
                t = sp / (sp - sc)
                out.append((prv[0] + t * (cur[0] - prv[0]), prv[1] + t * (cur[1] - prv[1])))
    if len(out) < 3:
        return 0.0
    x, y = np.array([p[0] for p in out]), np.array([p[1] for p in out])
    return 0.5 * abs(float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))))


def _standin_iou3d(boxes_a, boxes_b):
    a, b = boxes_a.double().numpy(), boxes_b.double().numpy()
    out = np.zeros((len(a), len(b)))

    def poly(bx):
        c, s_ = np.cos(bx[6]), np.sin(bx[6])
        loc = np.array([[bx[3] / 2, bx[4] / 2], [-bx[3] / 2, bx[4] / 2], [-bx[3] / 2, -bx[4] / 2], [bx[3] / 2, -bx[4] / 2]])
        return loc @ np.array([[c, s_], [-s_, c]]) + bx[:2]
    for i in range(len(a)):
        for j in range(len(b)):
            zo = min(a[i, 2] + a[i, 5] / 2, b[j, 2] + b[j, 5] / 2) - max(a[i, 2] - a[i, 5] / 2, b[j, 2] - b[j, 5] / 2)
            if zo > 0:
                inter = _clip_polygon_area(poly(a[i]), poly(b[j])) * zo
                out[i, j] = inter / (np.prod(a[i, 3:6]) + np.prod(b[j, 3:6]) - inter)
    return torch.from_numpy(out)


def _install_pcdet_dataset_standins():
    """Stand-ins for what the reference's dataset classes inherit / call from OpenPCDet (un-vendored): file layout and helper
    semantics as published by OpenPCDet; the reference's own classes then run unchanged on top."""
    import pickle
    import types
    from pathlib import Path
    refstubs.install()
    from src.utils import pointcloud_utils as ref_pu

    class Template:
        def __init__(self, dataset_cfg=None, class_names=None, training=True, root_path=None, logger=None):
            self.dataset_cfg, self.training, self.class_names, self.logger = dataset_cfg, training, class_names, logger
            self.root_path = Path(root_path) if root_path is not None else Path(dataset_cfg.DATA_PATH)
            self.point_cloud_range = np.array(dataset_cfg.POINT_CLOUD_RANGE, dtype=np.float32)

        @property
        def mode(self):
            return 'train' if self.training else 'test'

    class WaymoBase(Template):
        def __init__(self, dataset_cfg, class_names, training=True, root_path=None, logger=None):
            super().__init__(dataset_cfg, class_names, training, root_path, logger)
            self.data_path = self.root_path / dataset_cfg.PROCESSED_DATA_TAG
            self._load(dataset_cfg.DATA_SPLIT[self.mode])

        def set_split(self, split):
            self._load(split)

        def _load(self, split):
            self.split = split
            lines = (self.root_path / 'ImageSets' / (split + '.txt')).read_text().splitlines()
            self.infos = []
            for line in lines:
                seq = os.path.splitext(line.strip())[0]
                f = self.data_path / seq / (seq + '.pkl')
                if f.exists():
                    self.infos.extend(pickle.load(open(f, 'rb')))
            k = self.dataset_cfg.SAMPLED_INTERVAL[self.mode]
            if k > 1:
                self.infos = self.infos[::k]

        def get_lidar(self, sequence_name, sample_idx):
            arr = np.load(self.data_path / sequence_name / ('%04d.npy' % sample_idx))
            pts, flag = arr[:, 0:5], arr[:, 5]
            if not self.dataset_cfg.get('DISABLE_NLZ_FLAG_ON_POINTS', False):
                pts = pts[flag == -1]
            pts[:, 3] = np.tanh(pts[:, 3])
            return pts

    class Argo2Base(Template):
        def __init__(self, dataset_cfg, class_names, training=True, root_path=None, logger=None):
            super().__init__(dataset_cfg, class_names, training, root_path, logger)
            self.split = dataset_cfg.DATA_SPLIT[self.mode]
            self.root_split_path = self.root_path / ('training' if self.split != 'test' else 'testing')
            self.argo2_infos = []
            self.include_argo2_data(self.mode)

        def set_split(self, split):
            self.split = split
            self.root_split_path = self.root_path / ('training' if self.split != 'test' else 'testing')

        def include_argo2_data(self, mode):
            for rel in self.dataset_cfg.INFO_PATH[mode]:
                f = self.root_path / rel
                if f.exists():
                    self.argo2_infos.extend(pickle.load(open(f, 'rb')))
            self.infos = self.argo2_infos          # the reference reads `self.infos` (argo2_dataset.py:43,59)

        def get_lidar(self, idx):
            return np.fromfile(str(self.root_split_path / 'velodyne' / ('%s.bin' % idx)), dtype=np.float32).reshape(-1, 4)

    def drop_info_with_name(info, name):
        keep = [i for i, x in enumerate(info['name']) if x != name]
        return {k: info[k][keep] for k in info.keys()}

    def keep_arrays_by_name(gt_names, used_classes):
        return np.array([i for i, x in enumerate(gt_names) if x in used_classes], dtype=np.int64)

    def boxes_to_corners_3d(b):
        b = np.asarray(b)
        t = np.array([[1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1], [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1]]) / 2
        out = np.zeros((len(b), 8, 3))
        for i in range(len(b)):
            c, s_ = np.cos(b[i, 6]), np.sin(b[i, 6])
            Rz = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]])
            out[i] = (t * b[i, 3:6]) @ Rz.T + b[i, :3]
        return out

    def fakelidar_to_lidar(b):
        b = b.copy()
        w, l, h, r = b[:, 3:4], b[:, 4:5], b[:, 5:6], b[:, 6:7]
        b[:, 2] += h[:, 0] / 2
        return np.concatenate([b[:, 0:3], l, w, h, -(r + np.pi / 2)], axis=-1)

    cu = types.ModuleType('pcdet.utils.common_utils')
    cu.drop_info_with_name, cu.keep_arrays_by_name, cu.apply_transform = drop_info_with_name, keep_arrays_by_name, ref_pu.apply_transform
    bu = types.ModuleType('pcdet.utils.box_utils')
    bu.boxes_to_corners_3d, bu.boxes3d_kitti_fakelidar_to_lidar = boxes_to_corners_3d, fakelidar_to_lidar
    sys.modules['pcdet.utils'].common_utils, sys.modules['pcdet.utils'].box_utils = cu, bu
    sys.modules['pcdet.utils.common_utils'], sys.modules['pcdet.utils.box_utils'] = cu, bu
    iou = types.ModuleType('pcdet.ops.iou3d_nms.iou3d_nms_utils')
    iou.boxes_iou3d_gpu = _standin_iou3d
    sys.modules['pcdet.ops.iou3d_nms'].iou3d_nms_utils = iou
    sys.modules['pcdet.ops.iou3d_nms.iou3d_nms_utils'] = iou
    for name, cls, attr in (('waymo', WaymoBase, 'WaymoDataset'), ('argo2', Argo2Base, 'Argo2Dataset')):
        pkg = types.ModuleType(f'pcdet.datasets.{name}')
        mod = types.ModuleType(f'pcdet.datasets.{name}.{name}_dataset')
        setattr(mod, attr, cls)
        sys.modules.setdefault('pcdet.datasets', types.ModuleType('pcdet.datasets'))
        sys.modules[f'pcdet.datasets.{name}'] = pkg
        sys.modules[f'pcdet.datasets.{name}.{name}_dataset'] = mod
    # TensorFlow / waymo_open_dataset: only imported, never run (the metric op itself cannot be pinned)
    tf = types.ModuleType('tensorflow')
    tf.get_logger = lambda: types.SimpleNamespace(setLevel=lambda *a: None)
    tf.test = types.SimpleNamespace(TestCase=object)
    tf.config = types.SimpleNamespace(list_physical_devices=lambda *a: [])
    sys.modules['tensorflow'] = tf
    for m in ('waymo_open_dataset', 'waymo_open_dataset.metrics', 'waymo_open_dataset.metrics.python', 'waymo_open_dataset.protos'):
        sys.modules[m] = types.ModuleType(m)
    sys.modules['waymo_open_dataset.metrics.python'].detection_metrics = None
    sys.modules['waymo_open_dataset.protos'].metrics_pb2 = None
    sys.modules['waymo_open_dataset.protos'].breakdown_pb2 = None
    sys.modules['waymo_open_dataset'].label_pb2 = None


def dataset_detections(ds, seed=0):
    """Detections for the evaluation filters: the sequence's own (filtered) boxes, jittered, a few dropped, a few invented."""
    rng = np.random.default_rng(seed)
    out = []
    for f in range(ds.sequence_length):
        a = ds.get_annos(f)
        b = np.asarray(a['gt_boxes'], np.float64)[:, :7].copy()
        keep = rng.uniform(size=len(b)) < 0.8
        b = b[keep] + rng.normal(0, 0.05, size=(int(keep.sum()), 7))
        names = np.asarray(a['gt_names'])[keep]
        extra = np.c_[rng.uniform(-45, 45, 3), rng.uniform(-18, 18, 3), rng.uniform(0.5, 1.5, 3), rng.uniform(0.5, 5, (3, 3)), rng.uniform(-3, 3, 3)]
        out.append({'boxes_lidar': np.concatenate([b, extra]), 'name': np.concatenate([names, np.array(['Vehicle', 'Pedestrian', 'Cyclist'])]),
                    'score': rng.uniform(0.05, 1.0, size=len(b) + 3), 'moving': rng.uniform(size=len(b) + 3) < 0.5})
    return out


EVAL_VARIANTS = {
    'plain': dict(class_agnostic=False, bev=False, moving=False, static=False, score_thresh=0.0, sampling_rate=1),
    'moving': dict(class_agnostic=False, bev=False, moving=True, static=False, score_thresh=0.0, sampling_rate=1),
    'static': dict(class_agnostic=False, bev=False, moving=False, static=True, score_thresh=0.3, sampling_rate=1),
    'agnostic_bev': dict(class_agnostic=True, bev=True, moving=False, static=False, score_thresh=0.0, sampling_rate=2),
}


def make_dataset():
    """N3/N4: the reference's WaymoDataset / Argo2Dataset (src/datasets/*.py) run unchanged over stand-in OpenPCDet base classes on
    the seeded fixture trees of vilgod_amd/fixture_data.py; `evaluation` runs up to (and including) the estimator's own
    generate_waymo_type_results / mask_by_distance -- the TF metric op after that cannot run.  -> dataset_golden.pkl"""
    import pickle
    import tempfile
    from vilgod_amd import fixture_data as fx
    _install_pcdet_dataset_standins()
    from src.datasets.waymo_dataset import WaymoDataset
    from src.datasets.argo2_dataset import Argo2Dataset
    from src.datasets import waymo_eval
    captured = {}

    def record(self, prediction_infos, gt_infos, class_name, distance_thresh=100, fake_gt_infos=True, cfg={}):
        pd_ = self.generate_waymo_type_results(prediction_infos, class_name, is_gt=False)
        gt_ = self.generate_waymo_type_results(gt_infos, class_name, is_gt=True, fake_gt_infos=fake_gt_infos)
        pd_m = self.mask_by_distance(distance_thresh, pd_[1], pd_[0], pd_[2], pd_[3], pd_[4])
        gt_m = self.mask_by_distance(distance_thresh, gt_[1], gt_[0], gt_[2], gt_[3], gt_[5])
        captured['last'] = dict(pd=[np.array(x) for x in pd_m], gt=[np.array(x) for x in gt_m], distance_thresh=distance_thresh,
                                fake=fake_gt_infos, cfg=dict(cfg))
        return {}
    waymo_eval.OpenPCDetWaymoDetectionMetricsEstimator.waymo_evaluation = record
    log = type('L', (), {'info': lambda self, *a, **k: None})()
    classes = ['Vehicle', 'Pedestrian', 'Cyclist']
    out = {}
    for kind, Cls, cfg, writer in (('waymo', WaymoDataset, fx.WAYMO_CFG, fx.write_waymo), ('argo2', Argo2Dataset, fx.ARGO2_CFG, fx.write_argo2)):
        root = tempfile.mkdtemp()
        writer(root, n_sequences=3, n_frames=6, n_points=3000, n_objects=10, seed=5)
        ds = Cls(refstubs.AttrDict(dict(cfg, DATA_PATH=root)), classes, training=True, root_path=None, logger=log, start_sequence=0, end_sequence=2)
        ds.training = False
        rec = dict(mapping=ds.sequence_mapping, start=ds.start_sequence, end=ds.end_sequence, names=list(ds.sequence_names), sequences=[])
        for name in ds.next_sequence():
            sq = dict(name=name, indices=ds.sequence_indices, moving_ids=sorted(ds._moving_track_ids),
                      annos=[ds.get_annos(f) for f in range(ds.sequence_length)],
                      annos_T=ds.get_annos(1, transformation=np.linalg.inv(ds.sequence_infos[0]['pose']) @ ds.sequence_infos[1]['pose']),
                      poses=[np.array(i['pose']) for i in ds.sequence_infos], points0=ds.get_lidar_points(0),
                      points1_T=ds.get_lidar_points(1, transformation=ds.sequence_infos[1]['pose'])[:50])
            dets = dataset_detections(ds, seed=11)
            sq['eval'] = {}
            for vname, kw in EVAL_VARIANTS.items():
                if kind == 'argo2' and vname == 'moving':
                    pass                                     # argo2: no IoU-based removal, still a valid variant
                ds.evaluation(dets, classes, indices=ds.sequence_indices, eval_cfg=dict(difficulties=[2], breakdown_range=False,
                              iou_thresholds=[0.4, 0.4, 0.4, 0.4]), eval_range=[-50., -20., 50., 20.], **kw)
                sq['eval'][vname] = captured.pop('last')
            ds.evaluation(dets, classes, sequence=True, eval_range=[-50., -20., 50., 20.], **EVAL_VARIANTS['plain'])
            sq['eval']['sequence'] = captured.pop('last')
            rec['sequences'].append(sq)
        out[kind] = rec
    with open(os.path.join(OUT, 'dataset_golden.pkl'), 'wb') as f:
        pickle.dump(out, f)
    for kind, rec in out.items():
        print(kind, rec['names'], [len(s['indices']) for s in rec['sequences']], 'moving', [len(s['moving_ids']) for s in rec['sequences']],
              'eval pd/gt', [(len(s['eval']['plain']['pd'][0]), len(s['eval']['plain']['gt'][0]), len(s['eval']['moving']['pd'][0]),
                              len(s['eval']['moving']['gt'][0])) for s in rec['sequences']])


if __name__ == '__main__':
    which = sys.argv[1:] or ['render', 'vit']
    for w in which:
        globals()['make_' + w]()
