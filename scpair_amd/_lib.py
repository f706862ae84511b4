"""ctypes binding of libscpair_hip.so (C ABI declared in include/scpair_hip.h).

There is NO fallback: if the library is missing, importing any compute entry point raises.  The
binding passes raw device pointers (tensor.data_ptr()) and torch's current HIP stream; torch is used
only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libscpair_hip.so")

_lib = None

P = C.c_void_p
I = C.c_int
L = C.c_long
F = C.c_float
U64 = C.c_uint64


class AdamTensor(C.Structure):
    _fields_ = [("p", P), ("g", P), ("m", P), ("v", P), ("n", L),
                ("lr", F), ("weight_decay", F), ("bias_corr1", F), ("bias_corr2", F), ("grad_scale", F),
                ("p_bf16", P), ("cols", L), ("ld_bf16", L), ("g_bf16", L), ("bc_dev", P), ("ld_p", L)]


class EncTail(C.Structure):
    _fields_ = [("B", I), ("H1", I), ("H2", I), ("N3", I),
                ("b1", P), ("g1", P), ("be1", P), ("W2", P), ("b2", P), ("g2", P), ("be2", P), ("W3", P), ("b3", P),
                ("eps1", F), ("eps2", F), ("slope", F), ("p1", F), ("p2", F),
                ("seed1", U64), ("seed2", U64), ("seed_dev", P),
                ("xhat1", P), ("rstd1", P), ("xhat2", P), ("rstd2", P),
                ("slabs", P), ("slab_stride", L), ("ld_slab", L), ("nslab", I), ("out", P), ("ldo", L),
                ("dOut", P), ("lddo", L), ("dY1", P), ("ldy", L), ("dYT", P), ("ldt", L), ("ws", P),
                ("dg1", P), ("dbe1", P), ("db1", P), ("dW2", P), ("dg2", P), ("dbe2", P), ("db2", P), ("dW3", P),
                ("db3", P),
                ("cov", P), ("ldcov", L), ("C", I), ("W1c", P), ("ldw1c", L), ("dW1c", P), ("lddw1c", L),
                ("adam", I), ("adam_m", P), ("adam_v", P), ("adam_sh", P), ("adam_ldsh", L),
                ("lr", F), ("wd", F), ("beta1", F), ("beta2", F), ("adam_eps", F), ("bc1", F), ("bc2", F), ("gscale", F),
                ("bc_dev", P),
                ("bn", I), ("bn_train", I),
                ("bn1_w", P), ("bn1_b", P), ("bn1_rm", P), ("bn1_rv", P),
                ("bn2_w", P), ("bn2_b", P), ("bn2_rm", P), ("bn2_rv", P),
                ("bn_eps1", F), ("bn_eps2", F), ("bn_mom1", F), ("bn_mom2", F),
                ("bn_ws", P), ("bn_dy1", P),
                ("dbn1_w", P), ("dbn1_b", P), ("dbn2_w", P), ("dbn2_b", P)]


_SIGS = {
    "scp_last_error": (C.c_char_p, []),
    "scp_version": (I, []),
    "scp_tiles_ptr_len": (L, [I, I]),
    "scp_tiles_ws_len": (L, [I, I]),
    "scp_tiles_to_dense": (I, [I, I, P, P, P, P, L, P]),
    "scp_tiles_transform": (I, [I, I, P, P, P, P, P, I, F, P]),
    "scp_tiles_build": (I, [P, P, P, P, I, I, P, P, P, P, P, P, P, P]),
    "scp_gemm_split_count": (I, [I, I]),
    "scp_spmm_fwd": (I, [I, I, I, P, P, P, P, L, P, L, P, I, I, P]),
    "scp_spmm_dw": (I, [I, I, I, P, P, P, P, L, P, L, I, P]),
    "scp_cast_bf16": (I, [P, L, L, L, P, L, P]),
    "scp_relu_cast_bf16": (I, [P, L, L, L, P, L, P]),
    "scp_transpose_bf16": (I, [P, I, I, L, P, L, P]),
    "scp_split3_bf16": (I, [P, L, L, L, P, L, L, P]),
    "scp_split3_transpose_bf16": (I, [P, I, I, L, P, L, P]),
    "scp_spmm_fwd_bf16_min_split": (I, [I]),
    "scp_spmm_fwd_bf16": (I, [I, I, I, P, P, P, P, P, L, P, L, I, P]),
    "scp_spmm_dw_bf16": (I, [I, I, I, P, P, P, P, P, L, P, L, P]),
    "scp_spmm_dw_bf16_g16": (I, [I, I, I, P, P, P, P, P, L, P, L, P]),
    "scp_spmm_dw_adamw_bf16": (I, [I, I, I, P, P, P, P, P, L, P, L, P, P, P, L, F, F, F, F, F, F, F, F, P, P]),
    "scp_gemm_nt": (I, [I, I, I, P, L, L, P, L, L, P, L, P, I, I, I, P]),
    "scp_gemm_bf16_nsplit": (I, [I, I, I]),
    "scp_gemm_bf16_ws_floats": (L, [I, I, I, L]),
    "scp_gemm_bf16": (I, [I, I, I, I, I, P, L, P, L, P, L, P, P, L, P, P]),
    "scp_gemm_bf16_adamw": (I, [I, I, I, P, L, P, L, P, L, P, P, P, L, F, F, F, F, F, F, F, F, P, P]),
    "scp_slab_reduce": (I, [P, L, I, I, I, L, P, P, L, P]),
    "scp_colsum": (I, [P, L, I, I, P, P, P]),
    "scp_colsum_noalloc_max_cols": (I, []),
    "scp_colsum_noalloc": (I, [P, L, I, I, P, P]),
    "scp_bn_fwd": (I, [P, L, I, I, P, P, P, P, I, F, F, F, U64, P, P, P, P, L, P]),
    "scp_bn_bwd": (I, [P, L, P, L, I, I, P, P, P, I, F, U64, P, P, L, P, P, P]),
    "scp_ln_act_fwd": (I, [P, L, L, I, P, P, P, I, I, F, F, F, U64, P, P, L, P, P, L, P]),
    "scp_ln_act_bwd_grid": (I, [I]),
    "scp_ln_act_bwd": (I, [P, L, P, L, P, P, P, I, I, F, F, U64, P, P, L, P, P, P, P, P]),
    "scp_enc_tail_fwd": (I, [C.POINTER(EncTail), I, P]),
    "scp_enc_tail_ws_floats": (L, [I, I, I, I, I]),
    "scp_enc_tail_bn_ws_floats": (L, [I, I]),
    "scp_enc_tail_bwd": (I, [C.POINTER(EncTail), I, P]),
    "scp_dec_chunks": (I, [I]),
    "scp_dec_kp": (I, [I]),
    "scp_dec_ws_floats": (L, [I, I, I]),
    "scp_dec_ws_doubles": (L, [I, I]),
    "scp_ber_dec_bce": (I, [I, I, I, I, P, L, P, L, P, P, P, I, P, P, P, P, P, P, P, L, P, P, P, L, P, P, P, I, P]),
    "scp_nb_dec_ll": (I, [I, I, I, I, P, L, P, L, P, P, P, P, P, P, P, P, P, P, P, L, P, P, P, P, L, P, P, P, P, I, P]),
    "scp_mse_masked_fwd": (I, [P, L, P, L, I, I, P, P, P]),
    "scp_mse_masked_bwd": (I, [P, L, P, L, I, I, P, P, P, L, P]),
    "scp_bce_masked_fwd": (I, [P, P, L, P, P, P]),
    "scp_bce_masked_bwd": (I, [P, P, L, P, P, P, P]),
    "scp_nbll_fwd": (I, [P, P, P, I, I, P, L, P, P]),
    "scp_nbll_bwd": (I, [P, P, P, I, I, P, P, L, P, P, P, P]),
    "scp_reparam_kl_fwd": (I, [P, P, P, L, P, P, P, P, P]),
    "scp_reparam_kl_bwd": (I, [P, P, P, P, P, P, P, L, P, P, P]),
    "scp_ber_prob": (I, [P, L, I, I, P, I, P, P, L, I, P]),
    "scp_relu_softmax_rows": (I, [P, L, I, I, P, I, P, L, P, L, I, P]),
    "scp_raw_rows_fwd": (I, [P, L, P, L, P, L, P, L, P, I, I, I, P, P, P, P, P]),
    "scp_raw_rows_bwd": (I, [P, L, P, L, P, L, P, L, P, P, P, P, I, I, I, P, L, P, L, P, L, P, P]),
    "scp_raw_rows_feat_ws_floats": (L, [I, I]),
    "scp_raw_rows_feat_fwd": (I, [P, L, P, L, P, L, P, P, P, I, I, I, P, P, P, P, P]),
    "scp_raw_rows_feat_bwd": (I, [P, L, P, L, P, L, P, P, P, P, P, P, I, I, I, P, L, P, L, P, P, P, P, P]),
    "scp_fos_counts": (I, [P, L, P, L, I, I, P, P, P, P, P]),
    "scp_adamw_multi": (I, [C.POINTER(AdamTensor), I, F, F, F, P]),
    "scp_step_state_set": (I, [P, U64, F, F, P]),
}

EXPORTS = tuple(_SIGS)


def lib():
    """The loaded shared library (raises if it has not been built: `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with scpair_amd/csrc/build.sh "
                               "(there is no CPU/eager fallback for the HIP kernels)")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


class ScpError(RuntimeError):
    pass


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().scp_last_error().decode()
        if rc in (-1,):
            raise ValueError(f"{what}: {msg}")
        raise ScpError(f"{what}: error {rc}: {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """Handle of torch's current HIP stream on the current device (every kernel is launched on it)."""
    if _raw_stream is not None:
        # same value as torch.cuda.current_stream().cuda_stream without building the Python Stream object (called
        # ~50 times per training step)
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("scpair_amd HIP kernels need tensors in GPU memory (got a CPU tensor); "
                               "there is no CPU fallback")
