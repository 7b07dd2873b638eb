"""ctypes binding of the DEVELOPMENT build of the library (libvilgod_hip_dev.so = the product sources compiled with -DVG_DEV:
ablation variants, cycle-stamp traces, superseded kernels).  Used by tools/bench_gemm_*.py / bench_attention.py only.

    python -m vilgod_amd.build --dev
"""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from vilgod_amd import _lib, build  # noqa: E402

DEV_LIB = os.path.join(ROOT, 'vilgod_amd', 'libvilgod_hip_dev.so')


def load():
    if not os.path.exists(DEV_LIB):
        build.build(dev=True)
    lib = ctypes.CDLL(DEV_LIB)
    protos = dict(_lib.parse_header())
    protos.update(_lib.parse_header(os.path.join(HERE, 'vilgod_hip_dev.h')))
    for name, (restype, sig) in protos.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = [t for t, _ in sig]
    return lib


lib = load()
ptr, stream_ptr, check = _lib.ptr, _lib.stream_ptr, _lib.check
