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
    """A torch view of a CU-masked HIP stream; the HIP stream lives as long as this object."""

    def __new__(cls, words, device):
        device = torch.device(device)
        words = np.ascontiguousarray(words, dtype=np.uint32)
        h = ctypes.c_void_p()
        with torch.cuda.device(device):
            check(lib.vg_stream_create_cu_mask(ctypes.byref(h), words.ctypes.data_as(ctypes.c_void_p), len(words)),
                  'vg_stream_create_cu_mask')
        self = super().__new__(cls, h.value, device=device)
        self._vg_handle = h
        self.cu_mask_words = words.copy()
        return self

    def __del__(self):
        h = getattr(self, '_vg_handle', None)
        if h is not None and lib is not None:
            self._vg_handle = None
            try:
                lib.vg_stream_destroy(h)
            except Exception:       # noqa: BLE001  (interpreter shutdown)
                pass


def make_streams(device, reserve_per_xcd, tower='complement'):
    """-> (new_front_stream, new_tower_stream) factories for `device`; tower: 'complement' | 'all' (unmasked torch stream)."""
    n_cu = device_cu_count(device)
    front_words = cu_mask_words(n_cu, reserve_per_xcd, 'front')
    tower_words = cu_mask_words(n_cu, reserve_per_xcd, 'tower')

    def front():
        return MaskedStream(front_words, device)

    def tower_stream():
        if tower == 'all':
            return torch.cuda.Stream(device=device)
        return MaskedStream(tower_words, device)
    return front, tower_stream
