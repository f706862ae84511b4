"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol the header declares,
and argument validation answers without touching a GPU."""
import ctypes
import os
import re

import pytest

from scpair_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "scpair_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(scp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "libscpair_hip.so not built (run __graft_entry__.build())"
    h = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(h, s), f"{s} declared in include/scpair_hip.h but not exported"
    for s in _lib.EXPORTS:
        assert s in syms, f"{s} bound in _lib.py but not declared in the header"
    for s in syms:
        assert s in _lib.EXPORTS, f"{s} declared in the header but not bound in _lib.py"


def test_argument_validation_without_gpu():
    L = _lib.lib()
    assert L.scp_version() >= 1
    assert L.scp_gemm_nt(0, 4, 4, None, 1, 1, None, 1, 1, None, 1, None, 0, 1, 3, None) == -1
    assert b"positive" in L.scp_last_error()
    assert L.scp_tiles_build(None, None, None, None, 4, 4, None, None, None, None, None, None, None, None) == -1
    assert L.scp_dec_kp(40) == 48 and L.scp_dec_kp(9) == 16 and L.scp_dec_kp(50) == 64
    assert L.scp_dec_chunks(100000) == (100000 + 127) // 128      # chunks of DEC_NPT = 2 feature tiles of 64
    assert L.scp_tiles_ptr_len(130, 1000) == 3 * 1000 + 2
    assert L.scp_gemm_split_count(900, 4) >= 1
    with pytest.raises(ValueError):
        _lib.check(-1, "x")
