"""Host side of the multi-view depth renderer (mirror of the reference's
`src/utils/mv_utils.py::RealisticProjection`, same constructor argument and `get_img` meaning).

The reference renders one cluster per call (zero_shot_detector.py:389-399) and post-processes on the
host (:405-409, clip.py:79-86).  Here `render_frame` renders every cluster x view of a frame with the
HIP kernels of csrc/render.hip and returns the ViT-ready crops without leaving the GPU; `get_img`
keeps the reference's per-call interface for parity tests.
"""
import contextlib

import numpy as np
import torch
from scipy.spatial.transform import Rotation

from . import _lib
from ._lib import lib, ptr, stream_ptr, check

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # third_party/CLIP/clip/clip.py:85
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

OUT_U8, OUT_F32, OUT_F16, OUT_RAW110, OUT_PATCH16, OUT_PATCH16_1CH = 0, 1, 2, 3, 4, 5

# the 4 views hard-coded at mv_utils.py:134-141 plus the two commented-out ones (:139-140) that
# make up the 2x3 grid of waymo.yaml:97-102 (BASELINE config "6-view render")
VIEWS_4 = [(0.0, 0.0, 0.0), (-np.pi / 10, 0.0, 0.0), (0.0, np.pi / 30, 0.0), (0.0, -np.pi / 30, 0.0)]
VIEWS_6 = VIEWS_4 + [(-np.pi / 10, np.pi / 30, 0.0), (-np.pi / 10, -np.pi / 30, 0.0)]


def _euler_to_rotmat(angles):
    """[V,3] (x,y,z) float32 -> [V,3,3] = Rx @ Ry @ Rz, evaluated in float32 with torch like
    mv_utils.py:40-88 so the matrices carry the reference's bits."""
    a = torch.tensor(np.asarray(angles)).float()
    x, y, z = a[:, 0], a[:, 1], a[:, 2]
    o, l = torch.zeros_like(z), torch.ones_like(z)
    rz = torch.stack([z.cos(), -z.sin(), o, z.sin(), z.cos(), o, o, o, l], 1).reshape(-1, 3, 3)
    ry = torch.stack([y.cos(), o, y.sin(), o, l, o, -y.sin(), o, y.cos()], 1).reshape(-1, 3, 3)
    rx = torch.stack([l, o, o, o, x.cos(), -x.sin(), o, x.sin(), x.cos()], 1).reshape(-1, 3, 3)
    return rx @ ry @ rz


def _gaussian_taps(sigma):
    """Distinct taps of the normalised 3x3 Gaussian (mv_utils.py:204-220 with ksize 3, depth 1)."""
    xs = np.arange(3, dtype=np.float32) - 1
    k1 = np.exp(-(xs ** 2) / (2 * sigma ** 2))
    k2 = torch.from_numpy(k1[:, None] @ k1[None, :])
    k2 = k2 / k2.sum()
    k2 = torch.Tensor((k2[None] * np.ones((1, 1, 1), np.float32)) / torch.sum(k2[None]))[0]
    return float(k2[0, 0]), float(k2[0, 1]), float(k2[1, 1])


class RealisticProjection:
    def __init__(self, lidar_image_projection_cfg, device='cuda', views=None, angle_mode='device'):
        cfg = lidar_image_projection_cfg
        get = (lambda k, d=None: cfg.get(k, d)) if hasattr(cfg, 'get') else (lambda k, d=None: getattr(cfg, k, d))
        self.resolution = get('resolution', 112)
        self.depth = get('depth', 8)
        self.obj_ratio = get('obj_ratio', 0.8)
        self.depth_bias = get('depth_bias', 0.2)
        if (self.resolution, self.depth, self.obj_ratio, self.depth_bias) != (112, 8, 0.8, 0.2):
            raise NotImplementedError('csrc/render.hip is specialised for resolution 112, depth 8, obj_ratio 0.8, '
                                      'depth_bias 0.2 (tools/configs/preprocessor/*.yaml)')
        gk = get('gaussian_kernel', {'sigma': 3, 'zsigma': 1})
        sigma = gk['sigma'] if isinstance(gk, dict) else gk.sigma
        views = VIEWS_4 if views is None else views
        # view direction angle of a cluster (pointcloud_utils.py:397, float32 np.arctan2 of the median): 'device' = correctly
        # rounded on the GPU (no host round trip); 'reference' = this host's numpy evaluates it, like the reference would here
        # (numpy's float32 arctan2 is a <= 1 ulp routine that differs between hosts -- DESIGN.md section 4)
        if angle_mode not in ('device', 'reference'):
            raise ValueError(f'angle_mode {angle_mode!r}: device | reference')
        self.angle_mode = angle_mode
        self.num_views = len(views)
        self.device = torch.device(device)
        self.rot_mat = _euler_to_rotmat(views).transpose(1, 2).contiguous()          # mv_utils.py:165-166
        lut = torch.empty(3 * 256 + 3, dtype=torch.float32)
        lv = torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255)          # ToTensor
        for c in range(3):
            lut[c * 256:(c + 1) * 256] = lv.sub(torch.tensor(CLIP_MEAN[c])).div(torch.tensor(CLIP_STD[c]))
        lut[768], lut[769], lut[770] = _gaussian_taps(sigma)
        # Rx(pi) @ Rz(pi/2) exactly as scipy builds it (pointcloud_utils.py:392-393, 408-409)
        t_img = Rotation.from_euler('x', np.pi).as_matrix() @ Rotation.from_euler('z', np.pi / 2.).as_matrix()
        self._d_rot = self.rot_mat.reshape(-1, 9).to(self.device)
        self._d_lut = lut.to(self.device)
        self._d_timg = torch.from_numpy(np.ascontiguousarray(t_img)).to(self.device)

    # -- frame-level fused path -----------------------------------------------------------------
    def render_frame(self, points, index, seg_off, transform_to_ego, out='f16', stream=None, out_buf=None):
        """points: [N,>=3] float32 CUDA tensor (ref frame, `points_ref_wo_ground`);
        index: [Ptot] int32 packed cluster point indices (cluster after cluster) or None;
        seg_off: [C+1] int32 CUDA; transform_to_ego: 4x4 float64 (numpy or tensor).
        Returns crops for all C*V (cluster-major, like torch.cat of get_img results)."""
        kind = {'u8': OUT_U8, 'f32': OUT_F32, 'f16': OUT_F16, 'raw110': OUT_RAW110, 'patch16': OUT_PATCH16, 'patch16c1': OUT_PATCH16_1CH}[out]
        dev = points.device
        n_clusters = seg_off.numel() - 1
        ptot = int(index.numel()) if index is not None else int(points.shape[0])
        V = self.num_views
        n = n_clusters * V
        if kind == OUT_U8:
            result = torch.empty((n, 224, 224, 3), dtype=torch.uint8, device=dev)
        elif kind == OUT_F32:
            result = torch.empty((n, 3, 224, 224), dtype=torch.float32, device=dev)
        elif kind == OUT_F16:
            result = torch.empty((n, 3, 224, 224), dtype=torch.float16, device=dev)
        elif kind in (OUT_PATCH16, OUT_PATCH16_1CH):
            # ViT-B/16 patch rows; rows padded to the GEMM's 256-row tile (padding rows are never written: zeros, or -- in a
            # caller-owned persistent buffer `out_buf` -- finite rows of an earlier frame; GEMM rows are independent).
            # 'patch16c1': ONE channel per row (256 columns, level / 256): the crop's three channels are the same image
            # (mv_utils.py:36); the tower folds the per-channel normalisation into its patch embedding (vg_vit_encode input_kind 3)
            rows = (n * 196 + 255) // 256 * 256
            width = 768 if kind == OUT_PATCH16 else 256
            if out_buf is not None:
                assert out_buf.dtype == torch.float16 and out_buf.shape[1] == width and out_buf.shape[0] >= rows
                result = out_buf
            else:
                # the renderer writes every one of the n * 196 patch rows: only the <= 255 rows of tile padding need zeros
                # (a zero-fill of the whole buffer was 96 MB per frame)
                result = torch.empty((rows, width), dtype=torch.float16, device=dev)
                result[n * 196:].zero_()
        else:
            result = torch.empty((n, 110, 110), dtype=torch.float32, device=dev)
        if n_clusters == 0 or ptot == 0:
            return result
        sp = stream_ptr(stream)
        T = torch.as_tensor(np.ascontiguousarray(np.asarray(transform_to_ego, dtype=np.float64))).to(dev)
        ego = torch.empty((ptot, 3), dtype=torch.float32, device=dev)
        check(lib.vg_gather_ego(ptr(points), points.stride(0), ptr(index), ptot, ptr(T), ptr(ego), sp), 'vg_gather_ego')
        med = torch.empty((n_clusters, 3), dtype=torch.float32, device=dev)
        rot = torch.empty((n_clusters, 6), dtype=torch.float64, device=dev)
        check(lib.vg_cluster_median(ptr(ego), ptr(seg_off), n_clusters, ptr(med), ptr(rot), sp), 'vg_cluster_median')
        if self.angle_mode == 'reference':
            with torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext():
                m = med.cpu().numpy()                                     # [C,3] float32: one small read-back per frame
                ang = torch.from_numpy(np.arctan2(m[:, 1], m[:, 0])).to(dev)
            check(lib.vg_cluster_rot(ptr(ang), n_clusters, ptr(rot), sp), 'vg_cluster_rot')
        # per-point cluster id from the offsets
        ar = torch.arange(ptot, device=dev, dtype=torch.int32)
        pt_cluster = (torch.searchsorted(seg_off[1:].contiguous(), ar, right=True)).to(torch.int32)
        origin = torch.empty((ptot, 3), dtype=torch.float32, device=dev)
        check(lib.vg_to_origin(ptr(ego), ptr(pt_cluster), ptot, ptr(med), ptr(rot), ptr(self._d_timg), ptr(origin), sp),
              'vg_to_origin')
        check(lib.vg_render_crops(ptr(origin), ptr(seg_off), n_clusters, ptr(self._d_rot), V, ptr(self._d_lut),
                                  ptr(result), kind, sp), 'vg_render_crops')
        self._last = dict(ego=ego, median=med, origin=origin, rot=rot)
        return result

    def render_origin(self, origin, seg_off, out='f16', stream=None):
        """Render from already origin-transformed points (D1 output), [Ptot,3] float32 CUDA."""
        kind = {'u8': OUT_U8, 'f32': OUT_F32, 'f16': OUT_F16, 'raw110': OUT_RAW110}[out]
        n = (seg_off.numel() - 1) * self.num_views
        shape, dt = {OUT_U8: ((n, 224, 224, 3), torch.uint8), OUT_F32: ((n, 3, 224, 224), torch.float32),
                     OUT_F16: ((n, 3, 224, 224), torch.float16), OUT_RAW110: ((n, 110, 110), torch.float32)}[kind]
        result = torch.empty(shape, dtype=dt, device=origin.device)
        if n:
            check(lib.vg_render_crops(ptr(origin), ptr(seg_off), seg_off.numel() - 1, ptr(self._d_rot),
                                      self.num_views, ptr(self._d_lut), ptr(result), kind, stream_ptr(stream)),
                  'vg_render_crops')
        return result

    # -- reference-shaped per-call interface (mv_utils.py:173-187) ---------------------------------
    def get_img(self, points):
        """points: [b,P,3] float32 CUDA tensor of origin-transformed cluster points ->
        [b*V,3,110,110] float32 (three identical channels, as mv_utils.py:36)."""
        b, P, _ = points.shape
        seg = torch.arange(0, (b + 1) * P, P, dtype=torch.int32, device=points.device)
        raw = self.render_origin(points.reshape(b * P, 3).contiguous().float(), seg, out='raw110')
        return raw[:, None, :, :].repeat(1, 3, 1, 1)
