"""Stub-import harness for the read-only reference checkout (TEST INFRASTRUCTURE ONLY).

Used ONLY in the build container (where /root/reference exists) by
tests/golden/make_golden.py to run the reference's own Python on seeded inputs
and freeze the outputs as fixtures.  Nothing here is imported by the product
package, by `-m gpu` tests, by smoke() or by bench.py: /root/reference does not
exist on the GPU box.

The reference imports a dozen packages that are absent from this image
(torch_scatter, hydra, numba, pyransac3d, pytorch3d, pcdet, easydict, filterpy,
kornia, cv2, hdbscan).  We inject minimal stand-ins into sys.modules so that the
reference modules on the hot path import UNCHANGED:

  torch_scatter.scatter(src, index, dim, out, reduce="max")
        -> out.scatter_reduce_(dim, index, src, "amax", include_self=True)
           (mv_utils.py:124 is the only call site)
  hydra.utils.instantiate(cfg)   -> import cfg["_target_"] and call it
  numba.jit                      -> identity decorator
  torch.Tensor.cuda / nn.Module.cuda -> identity (CPU container)
"""
import importlib
import importlib.util
import sys
import types

REF = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class AttrDict(dict):
    """Minimal OmegaConf/EasyDict stand-in: attribute access on nested dicts."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v

    def __setattr__(self, k, v):
        self[k] = v

    def get(self, k, default=None):
        v = dict.get(self, k, default)
        return AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v


def _instantiate(cfg, *args, **kwargs):
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    modname, attr = target.rsplit(".", 1)
    fn = getattr(importlib.import_module(modname), attr)
    cfg = {k: (tuple(v) if isinstance(v, list) else v) for k, v in cfg.items()}
    cfg.update(kwargs)
    return fn(*args, **cfg)


def install():
    import torch

    if "torch_scatter" in sys.modules and getattr(sys.modules["torch_scatter"], "_vg_stub", False):
        return

    def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
        assert reduce == "max" and out is not None
        out.scatter_reduce_(dim, index, src, "amax", include_self=True)
        return out

    _mod("torch_scatter", scatter=scatter, _vg_stub=True)
    hydra = _mod("hydra")
    hydra.utils = _mod("hydra.utils", instantiate=_instantiate)

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    _mod("numba", jit=jit)
    _mod("pyransac3d")
    p3 = _mod("pytorch3d")
    p3.ops = _mod("pytorch3d.ops")
    p3.ops.knn = _mod("pytorch3d.ops.knn", knn_gather=None, knn_points=None)
    pc = _mod("pcdet")
    pc.__path__ = []
    pc.ops = _mod("pcdet.ops")
    pc.utils = _mod("pcdet.utils", common_utils=None)
    pc.ops.roiaware_pool3d = _mod("pcdet.ops.roiaware_pool3d", roiaware_pool3d_utils=None)
    pc.ops.pointnet2 = _mod("pcdet.ops.pointnet2")
    pc.ops.pointnet2.pointnet2_stack = _mod("pcdet.ops.pointnet2.pointnet2_stack", pointnet2_utils=None)
    pc.ops.iou3d_nms = _mod("pcdet.ops.iou3d_nms", iou3d_nms_utils=None)
    _mod("easydict", EasyDict=AttrDict)
    fp = _mod("filterpy")
    fp.kalman = _mod("filterpy.kalman", KalmanFilter=None)
    fp.common = _mod("filterpy.common", Q_discrete_white_noise=None)
    _mod("kornia")
    _mod("cv2")
    import sklearn.cluster

    _mod("hdbscan", HDBSCAN=sklearn.cluster.HDBSCAN)

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    if REF not in sys.path:
        sys.path.insert(0, REF)


def load_clip_model_py():
    """third_party/CLIP/clip/model.py standalone (``import clip`` needs torchvision/ftfy)."""
    spec = importlib.util.spec_from_file_location("_ref_clip_model", f"{REF}/third_party/CLIP/clip/model.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def projection_cfg():
    """tools/configs/preprocessor/waymo.yaml:75-96 after Hydra resolution."""
    return AttrDict(
        depth_bias=0.2, obj_ratio=0.8, bg_clr=0.0, resolution=112, depth=8,
        maxpool=dict(_target_="torch.nn.MaxPool3d", kernel_size=(1, 5, 5), stride=1, padding=(0, 1, 1)),
        conv3d=dict(_target_="torch.nn.Conv3d", in_channels=1, out_channels=1, kernel_size=(1, 3, 3),
                    stride=1, padding=(0, 1, 1), bias=True),
        gaussian_kernel=dict(sigma=3, zsigma=1),
    )
