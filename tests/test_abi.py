"""The C-ABI library loads on a CPU-only host and exports every symbol include/*.h declares.
No compute is called here (no GPU)."""
import ctypes
import glob
import os
import re

from conftest import ROOT


def test_every_declared_symbol_is_exported():
    from vilgod_amd import _lib
    declared = set()
    for h in glob.glob(os.path.join(ROOT, 'include', '*.h')):
        declared |= set(_lib.parse_header(h).keys())
    assert len(declared) >= 5
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    assert _lib.lib.vg_abi_version() >= 1


def test_header_has_no_torch_types():
    for h in glob.glob(os.path.join(ROOT, 'include', '*.h')):
        src = open(h).read()
        assert 'torch' not in re.sub(r'/\*.*?\*/', '', src, flags=re.S)
        assert 'at::' not in src and 'c10::' not in src


def test_header_cites_reference_lines():
    src = open(os.path.join(ROOT, 'include', 'vilgod_hip.h')).read()
    assert len(re.findall(r'\.(py|cpp|h):\d+', src)) >= 8
