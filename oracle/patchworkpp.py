"""ctypes front-end of the C++ ground-segmentation oracle (oracle/patchworkpp_oracle.cpp), shaped like
the reference's pybind11 module `pypatchworkpp` (python_wrapper/pybinding.cpp:14-53): `Parameters()`,
`patchworkpp(params)`, `.estimateGround(points)`, `.getGround()` ...

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see the .cpp header).
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, '_build', 'liboracle.so')


class Parameters(ctypes.Structure):
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
        load().pw_default_params(ctypes.byref(self))
        self.verbose = False


_lib = None


def build():
    subprocess.run(['make', '-C', HERE], check=True, capture_output=True)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = ctypes.CDLL(LIB)
        _lib.pw_create.restype = ctypes.c_void_p
        _lib.pw_create.argtypes = [ctypes.c_void_p]
        _lib.pw_destroy.argtypes = [ctypes.c_void_p]
        _lib.pw_estimate.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _lib.pw_get_state.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _lib.pw_num_patches.argtypes = [ctypes.c_void_p]
        _lib.pw_get_patch_info.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _lib.pw_eig3.argtypes = [ctypes.c_void_p] * 3
        _lib.pw_default_params.argtypes = [ctypes.c_void_p]
        _lib.pw_set_numeric_model.argtypes = [ctypes.c_void_p, ctypes.c_int]
    return _lib


class patchworkpp:
    def __init__(self, params, numeric_model=0):
        """numeric_model: 0 = float64 fixed-order sums (default; what the HIP kernels reproduce bit for bit), 1 / 2 = the float32
        two-pass arithmetic of patchworkpp.cpp:55-62 with scalar / 8-lane summation order (sensitivity study, see the .cpp header)."""
        self._lib = load()
        self._h = ctypes.c_void_p(self._lib.pw_create(ctypes.byref(params)))
        if self._lib.pw_set_numeric_model(self._h, int(numeric_model)):
            raise ValueError(f'numeric_model {numeric_model}')
        self._pts = None
        self._mask = None

    def __del__(self):
        if getattr(self, '_h', None):
            self._lib.pw_destroy(self._h)
            self._h = None

    def estimateGround(self, points):
        """points: [N, >=4] (x, y, z, intensity[, idx]); converted to float32 like Eigen::MatrixXf."""
        pts = np.ascontiguousarray(np.asarray(points)[:, :4], dtype=np.float32)
        mask = np.zeros(len(pts), dtype=np.uint8)
        self._lib.pw_estimate(self._h, pts.ctypes.data, len(pts), 4, mask.ctypes.data)
        self._pts, self._mask = pts, mask
        return mask

    def ground_mask(self):
        return self._mask.astype(bool)

    def getGround(self):
        idx = np.nonzero(self._mask)[0]
        return np.concatenate([self._pts[idx, :3], idx[:, None].astype(np.float32)], axis=1)

    def getNonground(self):
        idx = np.nonzero(self._mask == 0)[0]
        return np.concatenate([self._pts[idx, :3], idx[:, None].astype(np.float32)], axis=1)

    def state(self):
        out = np.zeros(17)
        self._lib.pw_get_state(self._h, out.ctypes.data)
        return dict(sensor_height=out[0], elevation_thr=out[1:5].copy(), flatness_thr=out[5:9].copy(),
                    n_elevation=out[9:13].astype(int), n_flatness=out[13:17].astype(int))

    def getHeight(self):
        return self.state()['sensor_height']

    def patch_info(self):
        n = self._lib.pw_num_patches(self._h)
        out = np.zeros((n, 12), dtype=np.float32)
        self._lib.pw_get_patch_info(self._h, out.ctypes.data)
        return out


def eig3(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    w = np.zeros(3)
    V = np.zeros((3, 3))
    load().pw_eig3(A.ctypes.data, w.ctypes.data, V.ctypes.data)
    return w, V


def mask_ground_points(points, pp, z_offset=0.0):
    """pointcloud_utils.py:49-56: returns ground point INDICES."""
    pts = np.asarray(points)[..., :4].astype(np.float64).copy()
    pts[..., 2] -= z_offset
    pp.estimateGround(pts)
    return pp.getGround()[..., -1].astype(int)
