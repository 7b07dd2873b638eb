"""ctypes binding of libvilgod_hip.so.

The prototypes are parsed from include/vilgod_hip.h, so the header is the single source of truth
for the ABI.  There is NO fallback: if the library is missing or a symbol cannot be resolved the
import fails loudly (the hot path has no CPU implementation in this package).
"""
import ctypes
import os
import re

# torch ships its own libamdhip64; it must be in the process BEFORE our library is loaded so that both resolve to
# ONE HIP runtime (otherwise our hipMalloc lands in a second, uninitialised runtime: "no ROCm-capable device").
import torch  # noqa: F401  (plumbing: device memory, streams, torch.distributed)

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), 'include', 'vilgod_hip.h')
LIB_PATH = os.environ.get('VILGOD_HIP_LIB') or os.path.join(HERE, 'libvilgod_hip.so')       # override: A/B runs of two builds on one box

_SCALARS = {
    'int': ctypes.c_int, 'int32_t': ctypes.c_int32, 'int64_t': ctypes.c_int64, 'uint32_t': ctypes.c_uint32,
    'uint64_t': ctypes.c_uint64, 'uint8_t': ctypes.c_uint8, 'size_t': ctypes.c_size_t, 'float': ctypes.c_float, 'double': ctypes.c_double,
    'void': None,
}


class VilgodHipError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """-> {name: (restype, [(ctype, argname), ...])} for every `vg_*` prototype."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    src = re.sub(r'//[^\n]*', ' ', src)
    protos = {}
    for m in re.finditer(r'\b(int|void|int64_t|size_t|double|const\s+char\s*\*)\s+(vg_\w+)\s*\(([^)]*)\)\s*;', src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        sig = []
        if args and args != 'void':
            for a in args.split(','):
                a = ' '.join(a.split())
                if '*' in a:
                    sig.append((ctypes.c_void_p, a.split('*')[-1].strip()))
                else:
                    toks = a.replace('const ', '').split()
                    sig.append((_SCALARS[toks[0]], toks[-1]))
        if 'char' in ret:
            restype = ctypes.c_char_p
        else:
            restype = _SCALARS[ret]
        protos[name] = (restype, sig)
    return protos


def _load():
    if not os.path.exists(LIB_PATH):
        raise VilgodHipError(
            f'{LIB_PATH} not found. Build it first:  python -m vilgod_amd.build  '
            '(there is no CPU fallback for the hot path).')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, sig) in parse_header().items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise VilgodHipError(f'{LIB_PATH} does not export {name} declared in {HEADER}; rebuild') from e
        fn.restype = restype
        fn.argtypes = [t for t, _ in sig]
    return lib


lib = _load()


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr(stream=None):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(s.cuda_stream)


def check(status, what=''):
    if status != 0:
        names = {1: 'bad argument', 2: 'HIP runtime error', 3: 'capacity exceeded'}
        raise VilgodHipError(f'{what} failed: status {status} ({names.get(status, "?")})')
