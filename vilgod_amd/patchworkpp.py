"""Host side of GPU ground segmentation: a mirror of the reference's pybind11 module `pypatchworkpp`
(third_party/patchwork-plusplus/python_wrapper/pybinding.cpp:9-55) on top of csrc/ground.hip.

    params = Parameters(); params.min_range = 1.5            # zero_shot_detector.py:137-139
    pp = patchworkpp(params)                                   # one per sequence (stateful), :140
    pp.estimateGround(points)                                  # numpy [N,>=4] or CUDA float32 tensor
    idx = pp.getGround()[:, -1].astype(int)                    # pointcloud_utils.py:54-56

`estimate_mask` is the zero-copy form the fused pipeline uses (CUDA in, CUDA uint8 mask out).
"""
import ctypes
import time

import numpy as np
import torch

from ._lib import lib, ptr, stream_ptr, check


class Parameters(ctypes.Structure):
    """patchwork::Params (patchworkpp.h:38-108), same attribute names as the pybind11 class."""
    _fields_ = [
        ('enable_RNR', ctypes.c_int), ('enable_RVPF', ctypes.c_int), ('enable_TGR', ctypes.c_int),
        ('num_iter', ctypes.c_int), ('num_lpr', ctypes.c_int), ('num_min_pts', ctypes.c_int),
        ('num_zones', ctypes.c_int), ('num_rings_of_interest', ctypes.c_int),
        ('RNR_ver_angle_thr', ctypes.c_double), ('RNR_intensity_thr', ctypes.c_double),
        ('sensor_height', ctypes.c_double), ('th_seeds', ctypes.c_double), ('th_dist', ctypes.c_double),
        ('th_seeds_v', ctypes.c_double), ('th_dist_v', ctypes.c_double), ('max_range', ctypes.c_double),
        ('min_range', ctypes.c_double), ('uprightness_thr', ctypes.c_double),
        ('adaptive_seed_selection_margin', ctypes.c_double),
        ('num_sectors_each_zone', ctypes.c_int * 4), ('num_rings_each_zone', ctypes.c_int * 4),
        ('max_flatness_storage', ctypes.c_int), ('max_elevation_storage', ctypes.c_int),
        ('elevation_thr', ctypes.c_double * 4), ('flatness_thr', ctypes.c_double * 4),
    ]

    def __init__(self):
        super().__init__()
        lib.vg_ground_default_params(ctypes.byref(self))
        self.verbose = False          # accepted for drop-in compatibility, unused
        self.intensity_thr = 0.0


class patchworkpp:
    def __init__(self, params, max_points=400_000, device='cuda'):
        self.device = torch.device(device)
        self.max_points = int(max_points)
        self._params = params
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.vg_ground_create(ctypes.byref(h), ctypes.byref(params), self.max_points), 'vg_ground_create')
        self._h = h
        self._pts = None
        self._mask = None

    def __del__(self):
        h = getattr(self, '_h', None)
        if h is not None and lib is not None:
            lib.vg_ground_destroy(h)
            self._h = None

    def reset(self, params=None):
        """New sequence: fresh adaptive state (the reference constructs a new object)."""
        if params is not None:
            self._params = params
        check(lib.vg_ground_reset(self._h, ctypes.byref(self._params) if params is not None else None),
              'vg_ground_reset')

    # -- zero-copy path -----------------------------------------------------------------------------
    def estimate_mask(self, points, z_offset=0.0, out=None, stream=None):
        """points: CUDA float32 [N,>=4] (x,y,z,intensity,...) -> CUDA uint8 [N], 1 = ground."""
        n = points.shape[0]
        assert points.is_cuda and points.dtype == torch.float32 and (n == 0 or points.stride(1) == 1)
        mask = out if out is not None else torch.empty(n, dtype=torch.uint8, device=points.device)
        if n == 0:                      # an empty scan: no ground, and the adaptive state is left untouched
            return mask
        check(lib.vg_ground_estimate(self._h, ptr(points), n, points.stride(0), float(z_offset), ptr(mask),
                                     stream_ptr(stream)), 'vg_ground_estimate')
        return mask

    # -- pypatchworkpp-shaped interface ------------------------------------------------------------------
    def estimateGround(self, points):
        """points: [N, >=4] numpy (any float dtype; converted to float32 like Eigen::MatrixXf) or CUDA tensor."""
        if isinstance(points, np.ndarray):
            pts = torch.from_numpy(np.ascontiguousarray(points[:, :4], dtype=np.float32)).to(self.device)
        else:
            pts = points[:, :4].contiguous().float()
        self._pts = pts
        t0 = time.perf_counter()
        self._mask = self.estimate_mask(pts, 0.0)
        torch.cuda.synchronize(self.device)
        self._time_taken_us = (time.perf_counter() - t0) * 1e6

    def _cloud(self, sel):
        idx = torch.nonzero(sel).squeeze(1)
        out = torch.cat([self._pts[idx, :3], idx[:, None].float()], dim=1)
        return out.cpu().numpy()

    def getGround(self):
        return self._cloud(self._mask != 0)

    def getNonground(self):
        return self._cloud(self._mask == 0)

    def state(self):
        out = (ctypes.c_double * 17)()
        check(lib.vg_ground_get_state(self._h, out, stream_ptr()), 'vg_ground_get_state')
        o = np.array(out[:])
        return dict(sensor_height=o[0], elevation_thr=o[1:5].copy(), flatness_thr=o[5:9].copy(),
                    n_elevation=o[9:13].astype(int), n_flatness=o[13:17].astype(int))

    def export_state(self):
        """The complete frame-to-frame state as bytes (vg_ground_export_state): hand it to `set_state` of another object
        (another stream / GPU / rank) and the sequence continues there exactly (frame-sharded sequences)."""
        buf = ctypes.create_string_buffer(int(lib.vg_ground_state_bytes()))
        check(lib.vg_ground_export_state(self._h, buf, stream_ptr()), 'vg_ground_export_state')
        return buf.raw

    def set_state(self, blob):
        if len(blob) != int(lib.vg_ground_state_bytes()):
            raise ValueError(f'ground state blob of {len(blob)} bytes, expected {int(lib.vg_ground_state_bytes())}')
        check(lib.vg_ground_set_state(self._h, ctypes.c_char_p(bytes(blob)), stream_ptr()), 'vg_ground_set_state')

    def getHeight(self):
        return self.state()['sensor_height']

    def getTimeTaken(self):
        """microseconds spent in the last estimateGround (patchworkpp.cpp:322, patchworkpp.h:151)."""
        return float(getattr(self, '_time_taken_us', 0.0))

    def patch_info(self):
        n = lib.vg_ground_num_patches(self._h)
        out = np.zeros((n, 12), dtype=np.float32)
        check(lib.vg_ground_get_patch_info(self._h, out.ctypes.data_as(ctypes.c_void_p), stream_ptr()),
              'vg_ground_get_patch_info')
        return out

    def getCenters(self):
        info = self.patch_info()
        return info[info[:, 0] >= self._params.num_min_pts][:, 5:8]

    def getNormals(self):
        info = self.patch_info()
        return info[info[:, 0] >= self._params.num_min_pts][:, 2:5]


def mask_ground_points_patchwork_pp(points, patchwork_pp, z_offset=0.0):
    """pointcloud_utils.py:49-56, same signature; returns ground point indices (numpy int)."""
    if isinstance(points, np.ndarray):
        pts = torch.from_numpy(np.ascontiguousarray(points[..., :4], dtype=np.float32)).to(patchwork_pp.device)
    else:
        pts = points
    mask = patchwork_pp.estimate_mask(pts, z_offset)
    return torch.nonzero(mask).squeeze(1).cpu().numpy().astype(int)
