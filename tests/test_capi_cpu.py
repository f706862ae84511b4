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


def test_wide_gemm_argument_validation_without_gpu():
    """scp_gemm_bf16 / scp_gemm_bf16_adamw / scp_raw_rows_feat_*: every refusal is decided on the host before any HIP
    call (sizes, null pointers, leading dimensions, alignment, missing workspace), with the reason in scp_last_error();
    the K-range count and the workspace size are pure functions of the shape."""
    L = _lib.lib()
    a = 4096                                         # any 16-byte aligned, never dereferenced address
    assert L.scp_gemm_bf16(0, 0, 0, 8, 8, a, 8, a, 8, a, 8, None, None, 0, None, None) == -1
    assert b"positive" in L.scp_last_error()
    assert L.scp_gemm_bf16(0, 0, 8, 8, 8, None, 8, a, 8, a, 8, None, None, 0, None, None) == -1
    assert L.scp_gemm_bf16(2, 0, 8, 8, 8, a, 8, a, 8, a, 8, None, None, 0, None, None) == -1
    assert L.scp_gemm_bf16(0, 0, 8, 8, 20, a, 16, a, 24, a, 8, None, None, 0, None, None) == -1     # lda < K rounded up to 8
    assert b"leading dimension" in L.scp_last_error()
    assert L.scp_gemm_bf16(0, 0, 8, 8, 16, a, 20, a, 16, a, 8, None, None, 0, None, None) == -1     # lda % 8 != 0
    assert L.scp_gemm_bf16(0, 0, 8, 8, 16, a + 2, 16, a, 16, a, 8, None, None, 0, None, None) == -1  # misaligned A
    assert L.scp_gemm_bf16(0, 0, 8, 8, 16, a, 16, a, 16, a, 4, None, None, 0, None, None) == -1     # ldc < N
    assert L.scp_gemm_bf16(0, 0, 70000, 8, 20000, a, 20000, a, 20000, a, 8, None, None, 0, None, None) == -2   # operand >= 2 GiB
    # 1024 x 20 000 x 20 000 runs as several K-ranges and refuses to start without its workspace
    ns = L.scp_gemm_bf16_nsplit(1024, 20000, 20000)
    assert 1 < ns <= 16 and ns == L.scp_gemm_bf16_nsplit(1024, 20000, 20000)
    assert L.scp_gemm_bf16_ws_floats(1024, 20000, 20000, 20000) == ns * 1024 * 20000
    assert L.scp_gemm_bf16_ws_floats(20000, 20000, 1024, 20000) == 0 and L.scp_gemm_bf16_nsplit(20000, 20000, 1024) == 1
    assert L.scp_gemm_bf16(0, 0, 1024, 20000, 20000, a, 20000, a, 20000, a, 20000, None, None, 0, None, None) == -1
    assert b"workspace" in L.scp_last_error()
    # AdamW in the weight-gradient epilogue: float4 epilogue needs N % 4 == 0 (SCP_ERR_SHAPE: use the two-kernel path)
    hp = (1e-3, 0.0, 0.9, 0.999, 1e-8, 0.1, 0.001, 1.0)
    assert L.scp_gemm_bf16_adamw(64, 66, 32, a, 64, a, 72, a, 66, a, a, a, 72, *hp, None, None) == -2
    assert L.scp_gemm_bf16_adamw(64, 64, 32, a, 64, a, 64, a, 64, a, a, None, 64, *hp, None, None) == -1
    assert L.scp_gemm_bf16_adamw(64, 64, 32, a, 64, a, 64, a, 64, a, a, a, 64, 1e-3, 0.0, 0.9, 0.999, 1e-8, 0.0, 0.001, 1.0,
                                 None, None) == -1
    assert b"bias corrections" in L.scp_last_error()
    # per-feature ZINB head
    assert L.scp_raw_rows_feat_ws_floats(1024, 20000) == 2 * 64 * 20000 and L.scp_raw_rows_feat_ws_floats(17, 5) == 2 * 2 * 5
    assert L.scp_raw_rows_feat_fwd(None, 8, None, 8, a, 8, a, None, None, 4, 8, 0, a, a, a, a, None) == -1
    assert L.scp_raw_rows_feat_bwd(a, 8, None, 8, a, 8, a, None, None, a, a, None, 4, 8, 0, a, 8, None, 8, None, a, None, a,
                                   None) == -1      # dfs asked for without fs
    assert b"absent parameter" in L.scp_last_error()
    assert L.scp_raw_rows_feat_bwd(a, 8, None, 8, a, 8, a, a, None, a, a, None, 4, 8, 0, a, 8, None, 8, None, a, None, None,
                                   None) == -1      # parameter gradient without the workspace
