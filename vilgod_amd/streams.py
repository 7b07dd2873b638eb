"""HIP streams restricted to a set of compute units (`vg_stream_create_cu_mask`, include/vilgod_hip.h).

Why: a projection-GEMM workgroup of the ViT tower (csrc/vit.hip k_gemm_f16_pp64) needs an EMPTY compute unit -- all 160 KB of LDS and
every vector register -- so every small workgroup of another frame's ground / clustering / render kernels that lands on a CU between
two tiles holds that CU's matrix pipe idle for its lifetime (tools/exp_interference.py: one frame's MST costs the GEMM stream 1.39 ms).
With `device.cu_reserve = r` the front-stage streams may only use r CUs of every XCD and (with `cu_tower = 'complement'`) the ViT
streams only the others.  Numerics cannot depend on where a workgroup runs; the sweep is recorded in LAB_NOTES.md.

Mask layout (amdkfd, gfx9.4.3+ in SPX mode): bit i of the mask is CU slot i // 8 of XCD i % 8, and consecutive slots of one XCD walk
its shader engines -- the low 8 r bits are r CUs of every XCD spread over its engines (tools/micro/cu_mask_probe.hip prints the
placement a mask really produces)."""
import ctypes
import threading

import numpy as np
import torch

from ._lib import lib, check

N_XCD = 8


def device_cu_count(device=None):
    n = ctypes.c_int32(0)
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        check(lib.vg_device_cu_count(ctypes.byref(n)), 'vg_device_cu_count')
    return int(n.value)


def cu_mask_words(n_cu, reserve_per_xcd, role):
    """-> uint32 words of the CU mask.  role 'front': the reserved CUs (the low 8 r bits); 'tower': every other CU."""
    r = int(reserve_per_xcd)
    if n_cu % N_XCD or n_cu < 8 * N_XCD:
        raise ValueError(f'CU masks assume {N_XCD} XCDs in SPX mode (MI300X / MI355X: 256 or 304 CUs); this device reports {n_cu} CUs -- '
                         'on another partition mode bit i of the mask is not CU slot i // 8 of XCD i % 8 (ADVICE r5)')
    if not 0 < r * N_XCD < n_cu:
        raise ValueError(f'cu_reserve must leave both sides at least one CU per XCD (got {r} of {n_cu // N_XCD})')
    bits = np.zeros((n_cu + 31) // 32 * 32, dtype=bool)
    if role == 'front':
        bits[:r * N_XCD] = True
    elif role == 'tower':
        bits[r * N_XCD:n_cu] = True
    else:
        raise ValueError(role)
    return np.packbits(bits.reshape(-1, 32)[:, ::-1], axis=1).view('>u4').astype(np.uint32).ravel()


class MaskedStream(torch.cuda.ExternalStream):
    """A torch view of a CU-masked HIP stream.  The HIP stream is created once and lives until the process ends: torch's caching
    allocator keeps events on every stream a tensor was `record_stream`ed on and records them when the block is freed, so destroying
    the stream under it crashes the process (seen: a segmentation fault when a test's pipeline went out of scope)."""

    def __new__(cls, words, device):
        device = torch.device(device)
        words = np.ascontiguousarray(words, dtype=np.uint32)
        h = ctypes.c_void_p()
        with torch.cuda.device(device):
            check(lib.vg_stream_create_cu_mask(ctypes.byref(h), words.ctypes.data_as(ctypes.c_void_p), len(words)),
                  'vg_stream_create_cu_mask')
        self = super().__new__(cls, h.value, device=device)
        self.cu_mask_words = words.copy()
        return self


# Every masked stream is its own hardware queue, and queues beyond GPU_MAX_HW_QUEUES are time-sliced (seven pipeline objects with masked
# streams in one process: everything ran 4x slower).  So the streams are pooled per (device, mask) and handed out by POSITION: worker k of
# every pipeline object of a process gets the same k-th stream -- two pipeline objects used at the same time would serialise on it, which
# the tools that build several avoid by running one variant per process.
_POOL = {}
_POOL_LOCK = threading.Lock()


def pooled_stream(device, words, position):
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), tuple(int(w) for w in words))
    with _POOL_LOCK:
        pool = _POOL.setdefault(key, [])
        while len(pool) <= position:
            pool.append(MaskedStream(words, device))
        return pool[position]


def make_streams(device, reserve_per_xcd, tower='complement'):
    """-> (front_stream(k), tower_stream(k)) for worker position k of `device`; tower: 'complement' | 'all' (unmasked torch stream)."""
    n_cu = device_cu_count(device)
    front_words = cu_mask_words(n_cu, reserve_per_xcd, 'front')
    tower_words = cu_mask_words(n_cu, reserve_per_xcd, 'tower')

    def front(k):
        return pooled_stream(device, front_words, k)

    def tower_stream(k):
        if tower == 'all':
            return torch.cuda.Stream(device=device)
        return pooled_stream(device, tower_words, k)
    return front, tower_stream
