"""Stand-ins that let the reference's OWN stage harness run in the build container (TEST INFRASTRUCTURE ONLY).

Used only by tests/golden/make_integration.py, here, where /root/reference exists: the reference's
`tools/preprocess_data.py::main` (sequence loop, :73-103) and `ZeroShotDetector.process()` with every stage method
(src/vilgod/zero_shot_detector.py:58-69, 105-123, 129-857), `LidarFrame` / `Detection` serialisation
(lidar_frame.py:41-59, objects.py:88-103) run UNCHANGED; only third-party packages that are absent from this image
are replaced, each at the level of the package's own call signature:

  hydra.main / omegaconf          decorator that hands the prepared config through; resolvers are no-ops
  pypatchworkpp                   oracle/patchworkpp.py (C++ restatement, same pybind surface)            [A1-A5]
  hdbscan.HDBSCAN                 oracle/hdbscan_oracle.fit behind .fit()/.labels_/.probabilities_        [B2]
  pyransac3d.Plane                oracle/segment_oracle.plane_ransac (hashed sample ids: seed, seed + 1 for the
                                  two Plane objects of one fit_plane call)                                [C2]
  pcdet ... pointnet2_utils.ball_query   the published ball_query_kernel_stack semantics (first `nsample` targets in index
                                  order with float32 fma d2 < r2, remaining slots = first hit)            [N1]
  pytorch3d.ops.knn.knn_points    K nearest by float32 fma squared distance, ascending, lowest index on ties [N1]
  numpy.random.choice             inside spatial_clustering's two-frame branch only: the counter-based half sample
                                  (oracle/neighbors_oracle.subsample_indices, keyed by frame) -- numpy's global Mersenne
                                  state after the reference's other draws cannot be reproduced on a GPU
  filterpy                        the published Kalman equations (tests/golden/make_golden.py::_RefKalman)  [N2]
  pcdet iou3d_nms_utils.boxes_iou3d_gpu   overlap sign by separating axes (make_golden.py::sat_iou3d)       [N2]
  clip                            `clip.load` -> the reference's own clip/model.py VisionTransformer in fp32 with the seeded
                                  synthetic weights (no checkpoint in the image) + seeded unit text features; `_transform`
                                  restated without torchvision (ToTensor + Normalize; Resize/CenterCrop are identities at 224)
  torch `.cuda()` / `.to('cuda')` identity (CPU container)
"""
import importlib.util
import sys
import types

import numpy as np
import torch

from . import refstubs
from . import hdbscan_oracle as ho
from . import neighbors_oracle as no
from . import segment_oracle as so


# ---- hdbscan --------------------------------------------------------------------------------------------------------------------
class HDBSCAN:
    """hdbscan.HDBSCAN(...) as the reference instantiates it (cluster_utils.py:11-12, waymo.yaml:10-15)."""

    def __init__(self, min_cluster_size=15, cluster_selection_epsilon=0.0, metric='euclidean', core_dist_n_jobs=-1, **kw):
        assert metric == 'euclidean'
        self.min_cluster_size, self.eps = int(min_cluster_size), float(cluster_selection_epsilon)

    def fit(self, X):
        X = np.asarray(X)
        if len(X) >= 2:
            self.labels_, self.probabilities_ = ho.fit(X, self.min_cluster_size, self.eps)
        else:
            self.labels_, self.probabilities_ = np.zeros(len(X), np.int64) - 1, np.zeros(len(X))
        return self


# ---- pyransac3d -----------------------------------------------------------------------------------------------------------------
class Plane:
    """pyransac3d.Plane: fit(pts, thresh, maxIteration) -> (equation, inlier indices).  pointcloud_utils.py:375-380 builds
    two objects per fit_plane call; the first draws with `seed`, the second with `seed + 1` (oracle/segment_oracle.fit_plane)."""
    seed = 666
    _made = 0

    def __init__(self):
        self._seed = Plane.seed + (Plane._made % 2)
        Plane._made += 1

    def fit(self, pts, thresh=0.05, minPoints=100, maxIteration=1000):
        eq, inl = so.plane_ransac(pts, thresh, maxIteration, self._seed)
        return list(eq), inl


# ---- pcdet ball_query / pytorch3d knn_points --------------------------------------------------------------------------------------
def ball_query(radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt):
    """pointnet2_stack ball_query: idx [M, nsample] int32 and the empty-ball mask (see module header)."""
    target = xyz.cpu().numpy().astype(np.float32)[:, :3]
    query = new_xyz.cpu().numpy().astype(np.float32)[:, :3]
    M = len(query)
    idx = np.zeros((M, nsample), np.int32)
    empty = np.ones(M, bool)
    if M and len(target):
        r2 = np.float32(radius) * np.float32(radius)
        lens, flat = no._candidates(query, target, float(radius))
        qi = np.repeat(np.arange(M), lens)
        hit = no.d2_f32(query[qi], target[flat]) < r2
        qi, flat = qi[hit], flat[hit]
        order = np.lexsort((flat, qi))
        qi, flat = qi[order], flat[order]
        start = np.searchsorted(qi, np.arange(M))
        cnt = np.searchsorted(qi, np.arange(M), side='right') - start
        for q in np.flatnonzero(cnt):
            k = min(int(cnt[q]), nsample)
            idx[q, :k] = flat[start[q]:start[q] + k]
            idx[q, k:] = flat[start[q]]
            empty[q] = False
    return torch.from_numpy(idx).unsqueeze(0), torch.from_numpy(empty)


def knn_points(p1, p2, K=1, **kw):
    """pytorch3d.ops.knn_points(p1 [1,N,3], p2 [1,M,3], K) -> .dists [1,N,K] squared float32 ascending, .idx [1,N,K]."""
    from scipy.spatial import cKDTree
    q = p1[0].cpu().numpy().astype(np.float32)[:, :3]
    t = p2[0].cpu().numpy().astype(np.float32)[:, :3]
    N, M = len(q), len(t)
    dists = np.zeros((N, K), np.float32)
    idx = np.zeros((N, K), np.int64)
    if N and M:
        kk = min(M, K + 8)
        _, cand = cKDTree(t.astype(np.float64)).query(q.astype(np.float64), k=kk)
        cand = cand.reshape(N, kk)
        d2 = no.d2_f32(q[:, None, :], t[cand])
        order = np.lexsort((cand, d2), axis=1)[:, :K]
        kq = min(K, kk)
        dists[:, :kq] = np.take_along_axis(d2, order, 1)[:, :kq]
        idx[:, :kq] = np.take_along_axis(cand, order, 1)[:, :kq]
    return types.SimpleNamespace(dists=torch.from_numpy(dists).unsqueeze(0), idx=torch.from_numpy(idx).unsqueeze(0))


# ---- numpy.random.choice inside the two-frame clustering ---------------------------------------------------------------------------
class HalfSample:
    """Replaces np.random.choice while ZeroShotDetector.spatial_clustering runs: the calls arrive in a known order
    (zero_shot_detector.py:211-228: for every frame fnr, for every f_idx of its window) so the frame each draw belongs to is
    known; the draw itself is the counter-based sample keyed by (seed, f_idx)."""

    def __init__(self, length, n_frames, seed):
        self.frames = [f for fnr in range(length) for f in range(min(fnr, length - n_frames), min(fnr + n_frames, length))]
        self.seed, self.k = seed, 0

    def __call__(self, n, size=None, replace=True, p=None):
        assert replace is False and p is None
        f = self.frames[self.k]
        self.k += 1
        return no.subsample_indices(self.seed, f, int(n), int(size))


# ---- clip -----------------------------------------------------------------------------------------------------------------------
MEAN = (0.48145466, 0.4578275, 0.40821073)
STD = (0.26862954, 0.26130258, 0.27577711)


def _preprocess(pil_image):
    """clip.py:79-86 at 224 x 224: Resize / CenterCrop return the input, convert('RGB'), ToTensor, Normalize."""
    assert pil_image.size == (224, 224)
    a = np.asarray(pil_image.convert('RGB'))
    t = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    mean = torch.tensor(MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(STD, dtype=torch.float32).view(3, 1, 1)
    return t.sub_(mean).div_(std)


class _SyntheticClip:
    def __init__(self, seed, n_classes):
        from vilgod_amd import clip_weights as cw
        m = refstubs.load_clip_model_py()
        self.visual = m.VisionTransformer(224, 16, 768, 12, 12, 512)
        self.visual.load_state_dict(cw.synthetic_vit_weights(seed, **cw.VIT_B16))
        self.visual.eval()
        self._text = cw.synthetic_text_features(seed, n_classes, 512)

    def encode_image(self, image):
        return self.visual(image.float())

    def encode_text(self, tokens):
        return self._text.clone()


def install(cfg, plane_seed=666, subsample_seed=0, clip_seed=0):
    """Everything the reference harness imports.  Returns the imported reference `tools/preprocess_data.py` module."""
    refstubs.install()
    from . import patchworkpp as opw
    sys.modules['pypatchworkpp'] = opw
    sys.modules['hdbscan'].HDBSCAN = HDBSCAN
    Plane.seed, Plane._made = int(plane_seed), 0
    sys.modules['pyransac3d'].Plane = Plane
    sys.modules['pcdet.ops.pointnet2.pointnet2_stack'].pointnet2_utils = types.SimpleNamespace(ball_query=ball_query)
    sys.modules['pytorch3d.ops.knn'].knn_points = knn_points
    # tests/golden/make_golden.py holds the filterpy / iou3d stand-ins of the N2 golden
    import os
    gdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
    if gdir not in sys.path:
        sys.path.insert(0, gdir)
    import make_golden as mg
    sys.modules['filterpy.kalman'].KalmanFilter = mg._RefKalman
    sys.modules['filterpy.common'].Q_discrete_white_noise = mg._ref_q_discrete_white_noise
    sys.modules['pcdet.ops.iou3d_nms'].iou3d_nms_utils = types.SimpleNamespace(boxes_iou3d_gpu=mg.sat_iou3d)

    n_classes = len(cfg.preprocessor.clip.class_list)
    model = _SyntheticClip(clip_seed, n_classes)
    clip = types.ModuleType('clip')
    clip.load = lambda name, device=None, jit=False, download_root=None: (model, _preprocess)
    clip.tokenize = lambda texts, context_length=77, truncate=False: torch.zeros(len(texts), context_length, dtype=torch.int64)
    sys.modules['clip'] = clip

    # CPU container: device moves are identities
    orig_to = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple('cpu' if (isinstance(x, str) and x.startswith('cuda')) else x for x in a)
        if isinstance(k.get('device'), str) and k['device'].startswith('cuda'):
            k['device'] = 'cpu'
        return orig_to(self, *a, **k)
    torch.Tensor.to = to
    torch.cuda.empty_cache = lambda: None
    torch.cuda.manual_seed = lambda s: None

    # hydra / omegaconf: the config is prepared by the caller (the repo's YAML tree, which mirrors the reference's keys)
    hydra = sys.modules['hydra']
    hydra.main = lambda **kw: (lambda fn: fn)
    oc = types.ModuleType('omegaconf')
    oc.DictConfig = dict
    oc.OmegaConf = types.SimpleNamespace(register_new_resolver=lambda *a, **k: None)
    sys.modules['omegaconf'] = oc

    spec = importlib.util.spec_from_file_location('_ref_preprocess_data', f'{refstubs.REF}/tools/preprocess_data.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # the reference's evaluate-all-sequences tail prints a Waymo AP table: not part of the pickles
    mod.eval_utils.print_eval_log = lambda ap_dict, logger: None

    # the half sample of the two-frame branch
    from src.vilgod import zero_shot_detector as zsd
    orig_sc = zsd.ZeroShotDetector.spatial_clustering

    def spatial_clustering(self, **kwargs):
        n_frames = kwargs.get('n_frames', 1)
        if n_frames <= 1:
            return orig_sc(self, **kwargs)
        keep = np.random.choice
        np.random.choice = HalfSample(len(self.lidar_frame_list), n_frames, subsample_seed)
        try:
            return orig_sc(self, **kwargs)
        finally:
            np.random.choice = keep
    zsd.ZeroShotDetector.spatial_clustering = spatial_clustering
    return mod
