"""torch.autograd.Function wrappers over the C-ABI HIP kernels (include/scpair_hip.h).

Every op here runs on the GPU through libscpair_hip.so; nothing falls back to eager torch math.
torch supplies device memory, the current stream and the autograd tape only.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib
from .sparse import BatchTiles

import os as _os

_cfg = {"ns": 3, "seed_ctr": 0, "fused_timer": None, "decoder_timer": None, "fwd_wgs": int(_os.environ.get("SCPAIR_FWD_WGS", "256"))}


def set_precision(mode: str):
    """'fp32'  -> every fp32 MFMA operand is split into 3 bf16 terms (fp32-equivalent products)
       'bf16'  -> operands rounded to bf16 once (fp32 accumulate, fp32 master weights)"""
    if mode not in ("fp32", "bf16"):
        raise ValueError("precision must be 'fp32' or 'bf16'")
    _cfg["ns"] = 3 if mode == "fp32" else 1


def get_precision() -> str:
    return "fp32" if _cfg["ns"] == 3 else "bf16"


def _ns():
    return _cfg["ns"]


def set_fused_timer(sink):
    """Diagnostic: a list that receives (start_event, end_event, n_params, operand_bytes, kind) for every fused
    weight-gradient + AdamW launch (bench.py's roofline figure), or None to switch it off.  kind "spmm_dw_adamw": the
    sparse first layers (operand_bytes = the tile-list entries and the bf16 dY^T the launch reads); "wgemm_adamw": a wide
    plain layer (operand_bytes = the bf16 dy and x operands)."""
    _cfg["fused_timer"] = sink


def set_decoder_timer(sink):
    """Diagnostic: a list that receives (start_event, end_event, kind, B, F, K, train) for every call of a fused decoder
    entry point (kind "ber": scp_ber_dec_bce, "nb": scp_nb_dec_ll; each call = its passes + finishing launches), or None."""
    _cfg["decoder_timer"] = sink


def set_seed_rank(rank: int):
    """Data parallelism: mixes the rank into every dropout seed drawn from now on (scpair_amd.distributed.GradSync)."""
    _cfg["seed_rank"] = int(rank)


def next_seed() -> int:
    _cfg["seed_ctr"] += 1
    return ((torch.initial_seed() & 0xFFFFFFFF) << 24) + _cfg["seed_ctr"] + (_cfg.get("seed_rank", 0) << 56)


class StepState:
    """16 bytes of device memory {uint64 seed offset; float bias_corr1; float bias_corr2} behind a training step
    captured in a HIP graph (scpair_amd.graph.CapturedStep): the graph bakes every by-value kernel argument, so the
    dropout kernels add `seed_ptr[0]` to their seed and the AdamW kernels read their bias corrections from `bc_ptr`;
    `set()` rewrites the words before each replay (scp_step_state_set, by-value arguments of a tiny launch)."""

    def __init__(self, device):
        self.buf = torch.zeros(2, dtype=torch.int64, device=device)
        self.seed_ptr = self.buf.data_ptr()
        self.bc_ptr = self.buf.data_ptr() + 8

    def set(self, seed_offset: int, bc1: float, bc2: float):
        _lib.check(_lib.lib().scp_step_state_set(self.buf.data_ptr(), int(seed_offset) & 0xFFFFFFFFFFFFFFFF, float(bc1),
                                                 float(bc2), _lib.stream()), "scp_step_state_set")


_cfg["step_state"] = None


def set_step_state(state):
    """The StepState the dropout kernels launched from now on read their seed offset from (None: by-value seeds only).
    Set by CapturedStep around the capture of a training step."""
    _cfg["step_state"] = state


def _seed_dev():
    st = _cfg["step_state"]
    return None if st is None else st.seed_ptr


# ---- bf16 shadow copies of first-layer weights (precision 'bf16') ------------------------------------------
import weakref

_shadows = {}          # id(W) -> (weakref(W), shadow, version); tensors compare element-wise, so no WeakKeyDictionary


def row_ld(W):
    """Row stride (elements) of a 2-D weight whose rows are contiguous -- W itself contiguous, or a row-padded view
    made by home_first_layers -- else None."""
    if W.ndim != 2 or W.stride(1) != 1 or W.stride(0) < W.shape[1]:
        return None
    return W.stride(0)


def shadow_of(W, create=True):
    """bf16 copy [H, ld] (ld = cols rounded up to 64: rows start on 128-byte boundaries; zero padded) of a 2-D fp32
    weight, for the bf16 first-layer kernels.  It is refreshed here whenever torch has modified W (version counter) and
    rewritten by FusedAdamW in the same kernel that updates W, so it never lags the master weights."""
    ent = _shadows.get(id(W))
    if ent is not None and ent[0]() is not W:          # id reused by another tensor
        ent = None
    else:
        ent = None if ent is None else (ent[1], ent[2], ent[3])
    if ent is not None and ent[1] == W._version and ent[2] == W.data_ptr() and ent[0].device == W.device:
        return ent[0]
    if not create:
        return None
    if len(_shadows) > 64:
        for k in [k for k, v in _shadows.items() if v[0]() is None]:
            del _shadows[k]
    L = _lib.lib()
    H, cols = W.shape
    ld = (cols + 63) // 64 * 64
    pk, off = _pack_of(W)
    if pk is not None and pk[1].device == W.device:   # homed weights: adjacent row blocks of one shadow
        if pk[2]["shadow"] is None or pk[2]["shadow"].shape != (pk[1].shape[0], ld) or pk[2]["shadow"].device != W.device:
            pk[2]["shadow"] = torch.empty((pk[1].shape[0], ld), dtype=torch.bfloat16, device=W.device)
        sh = pk[2]["shadow"][off:off + H]
    elif ent is not None and ent[0].shape == (H, ld) and ent[0].device == W.device:
        sh = ent[0]
    else:
        sh = torch.empty((H, ld), dtype=torch.bfloat16, device=W.device)
    Wc = W.detach()
    ldw = row_ld(Wc)
    if ldw is None:
        raise RuntimeError("first-layer weights must have contiguous rows")
    _lib.check(L.scp_cast_bf16(Wc.data_ptr(), H, cols, ldw, sh.data_ptr(), ld, _lib.stream()), "scp_cast_bf16")
    _shadows[id(W)] = (weakref.ref(W), sh, W._version, W.data_ptr())
    return sh


def _f32c(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _empty(shape, like, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=like.device)


# ------------------------------------------------------------------------------------------------
# raw launchers
# ------------------------------------------------------------------------------------------------
def gemm_nt(M, N, K, A, a_sr, a_sk, B, b_sr, b_sk, Cout, ldc, bias=None, accumulate=False, nsplit=1, ns=None,
            a_off=0, b_off=0, c_off=0):
    L = _lib.lib()
    _lib.check(L.scp_gemm_nt(M, N, K, A.data_ptr() + 4 * a_off, a_sr, a_sk, B.data_ptr() + 4 * b_off, b_sr, b_sk,
                             Cout.data_ptr() + 4 * c_off, ldc, _lib.ptr(bias), int(accumulate), nsplit,
                             ns or _ns(), _lib.stream()), "scp_gemm_nt")


def slab_reduce(slabs, nslab, rows, cols, out, bias=None):
    L = _lib.lib()
    _lib.check(L.scp_slab_reduce(slabs.data_ptr(), rows * cols, nslab, rows, cols, cols, _lib.ptr(bias), out.data_ptr(),
                                 cols, _lib.stream()), "scp_slab_reduce")


def gemm_nt_auto(M, N, K, A, a_sr, a_sk, B, b_sr, b_sk, Cout, bias=None, ns=None):
    """C[M,N] (contiguous) = A B^T + bias; few-tile / deep-K shapes (the 900->40 layers and their weight gradients)
    are split along K into slabs so they do not run on a handful of workgroups, then reduced in a fixed order."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    nsplit = 1
    if tiles < 64 and K >= 512:
        nsplit = max(1, min(K // 128, 256 // tiles, 32))
    if nsplit > 1:
        nsplit = _lib.lib().scp_gemm_split_count(K, nsplit)
    if nsplit <= 1:
        gemm_nt(M, N, K, A, a_sr, a_sk, B, b_sr, b_sk, Cout, N, bias=bias, ns=ns)
        return
    slabs = torch.empty((nsplit, M, N), dtype=torch.float32, device=Cout.device)
    gemm_nt(M, N, K, A, a_sr, a_sk, B, b_sr, b_sk, slabs, N, nsplit=nsplit, ns=ns)
    slab_reduce(slabs, nsplit, M, N, Cout, bias)


def colsum(X, rows, cols, ld=None):
    L = _lib.lib()
    out = _empty((cols,), X)
    ws = _empty((32 * cols,), X)
    _lib.check(L.scp_colsum(X.data_ptr(), ld or cols, rows, cols, out.data_ptr(), ws.data_ptr(), _lib.stream()),
               "scp_colsum")
    return out


def _pick_split(M, N, K):
    """Split-K factor so that a tall-K GEMM still fills ~2 waves of 256 CUs."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    want = max(1, 512 // max(tiles, 1))
    ksteps = (K + 63) // 64
    return max(1, min(want, ksteps // 4 if ksteps >= 8 else 1, 32))


# ------------------------------------------------------------------------------------------------
# Linear (dense):  y = x W^T + b
# ------------------------------------------------------------------------------------------------
_LIB_GEMM_MIN = 1 << 29      # M*N*K from which a plain nn.Linear goes to the library bf16 GEMM (precision 'bf16' only)


def cast_bf16(x, rows, cols, relu=False):
    """bf16 copy [rows, ld] (ld = cols rounded up to 8, zero padded) of a contiguous fp32 [rows, cols] tensor
    (relu: of relu(x), in the same pass)."""
    ld = (cols + 7) // 8 * 8
    out = torch.empty((rows, ld), dtype=torch.bfloat16, device=x.device)
    L = _lib.lib()
    fn, name = (L.scp_relu_cast_bf16, "scp_relu_cast_bf16") if relu else (L.scp_cast_bf16, "scp_cast_bf16")
    _lib.check(fn(x.data_ptr(), rows, cols, cols, out.data_ptr(), ld, _lib.stream()), name)
    return out


def gemm_bf16(a_col, b_col, M, N, K, A, lda, B, ldb, C, ldc, bias=None, mask=None, ldm=0):
    """C[m][n] = sum_k A(m,k) B(n,k) (+ bias[n]) (zero where mask <= 0) on the hand-written wide MFMA kernel
    (csrc/wgemm.hip); A / B bf16, stored [rows][ld] (col flag 0) or [K][ld] (col flag 1); C fp32 [M][ldc]."""
    L = _lib.lib()
    nws = L.scp_gemm_bf16_ws_floats(M, N, K, ldc)
    ws = torch.empty((nws,), dtype=torch.float32, device=C.device) if nws else None
    _lib.check(L.scp_gemm_bf16(a_col, b_col, M, N, K, A.data_ptr(), lda, B.data_ptr(), ldb, C.data_ptr(), ldc,
                               _lib.ptr(bias), _lib.ptr(mask), ldm, _lib.ptr(ws), _lib.stream()), "scp_gemm_bf16")


class LinearFn(torch.autograd.Function):
    """y = [relu](x) W^T + b.  Wide layers in precision 'bf16' (the ZINB G x G zero-inflation layer model.py:228,
    the VAE 900 <-> G layers pairedVAE.py:111-119) run forward, input gradient and weight gradient on the hand-written
    wide MFMA kernel (scp_gemm_bf16): bf16 weight shadow (kept current by FusedAdamW) x bf16 activations, bias in the
    forward epilogue, the relu's backward as a mask in the input-gradient epilogue.  relu_in fuses the torch.relu the
    reference applies to the layer's input (model.py:240)."""

    @staticmethod
    def forward(ctx, x, W, b, relu_in=False, fused=None):
        _lib.require_cuda(x, W, b)
        x = _f32c(x)
        Wc = _f32c(W)
        M, K = x.shape
        N = Wc.shape[0]
        y = _empty((M, N), x)
        ctx.has_bias = b is not None
        ctx.relu_in = bool(relu_in)
        ctx.lib = _ns() == 1 and M * N * K >= _LIB_GEMM_MIN and W.is_contiguous() and W.dtype == torch.float32
        # AdamW step inside the weight-gradient kernel (FusedAdamW.fuse_weight_update): wide path only, float4 epilogue
        ctx.fused = fused if (ctx.lib and K % 4 == 0) else None
        if ctx.lib:
            sh = shadow_of(W)
            xb = cast_bf16(x, M, K, relu=ctx.relu_in)
            gemm_bf16(0, 0, M, N, K, xb, xb.shape[1], sh, sh.shape[1], y, N, bias=_f32c(b))
            ctx.save_for_backward(xb, W)
            ctx.dims = (M, N, K)
            return y
        if ctx.relu_in:
            x = torch.relu(x)
        gemm_nt_auto(M, N, K, x, K, 1, Wc, K, 1, y, bias=_f32c(b))
        ctx.save_for_backward(x, Wc)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _f32c(dy)
        dx = dW = db = None
        if ctx.lib:
            xb, W = ctx.saved_tensors
            M, N, K = ctx.dims
            dyb = cast_bf16(dy, M, N)
            if ctx.needs_input_grad[0]:
                sh = shadow_of(W)
                dx = _empty((M, K), dy)
                # dx[m][j] = sum_n dy[m][n] W[n][j]: W is the k-major ("col") operand; relu'(x) = [relu(x) > 0]
                gemm_bf16(0, 1, M, K, N, dyb, dyb.shape[1], sh, sh.shape[1], dx, K,
                          mask=xb if ctx.relu_in else None, ldm=xb.shape[1])
            if ctx.needs_input_grad[1] and ctx.fused is not None:
                # weight gradient + AdamW step of W in one kernel (after dx, which reads the shadow this rewrites):
                # dW never reaches HBM, autograd gets no gradient for W
                m_, v_, sh_, hp = ctx.fused.begin_step(W)
                tm = _cfg["fused_timer"]
                if tm is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                _lib.check(_lib.lib().scp_gemm_bf16_adamw(N, K, M, dyb.data_ptr(), dyb.shape[1], xb.data_ptr(), xb.shape[1],
                                                          W.data_ptr(), K, m_.data_ptr(), v_.data_ptr(), sh_.data_ptr(),
                                                          sh_.shape[1], *hp, _lib.stream()), "scp_gemm_bf16_adamw")
                if tm is not None:
                    e1.record()
                    tm.append((e0, e1, N * K, 2 * M * (dyb.shape[1] + xb.shape[1]), "wgemm_adamw"))
            elif ctx.needs_input_grad[1]:
                dW = _empty((N, K), dy)
                # dW[n][j] = sum_m dy[m][n] x[m][j]: both operands k-major
                gemm_bf16(1, 1, N, K, M, dyb, dyb.shape[1], xb, xb.shape[1], dW, K)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = colsum(dy, M, N)
            return dx, dW, db, None, None
        x, W = ctx.saved_tensors
        M, K = x.shape
        N = W.shape[0]
        if ctx.needs_input_grad[0]:
            dx = _empty((M, K), x)
            # dx[m][j] = sum_n dy[m][n] W[n][j]
            gemm_nt_auto(M, K, N, dy, N, 1, W, 1, K, dx)
            if ctx.relu_in:
                dx = dx * (x > 0)
        if ctx.needs_input_grad[1]:
            dW = _empty((N, K), x)
            # dW[n][j] = sum_m dy[m][n] x[m][j]
            gemm_nt_auto(N, K, M, dy, 1, N, x, 1, K, dW)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(dy, M, N)
        return dx, dW, db, None, None


def linear(x, W, b=None, relu_input=False):
    fused = None
    if _fused and torch.is_grad_enabled() and W.requires_grad:
        fused = fused_handle(W)
    return LinearFn.apply(x, W, b, relu_input, fused)


# ------------------------------------------------------------------------------------------------
# FCNN layer: Linear -> [LayerNorm] -> LeakyReLU(slope) -> Dropout(p), dense or sparse input
# (utils.py:54-64; BatchNorm, when enabled, is applied by the caller between act and dropout)
# ------------------------------------------------------------------------------------------------
class FcnnLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cov, W, b, gamma, beta, slope, p_drop, seed, eps, shadow=None, fused=None, sink=None):
        """x: dense [B, F] tensor or BatchTiles; cov: dense [B, C] or None; W: [H, F + C];
        shadow: bf16 copy of W for the bf16 sparse kernels (or None); fused: FusedFirstLayer handle when the AdamW
        step of W runs inside the weight-gradient kernel (or None)."""
        L = _lib.lib()
        sparse = isinstance(x, BatchTiles)
        Wc = W if (W.dtype == torch.float32 and row_ld(W) is not None) else _f32c(W)   # contiguous or row-padded (homed)
        _lib.require_cuda(Wc, cov, None if sparse else x)
        H, Ftot = Wc.shape
        ldw = row_ld(Wc)
        Bn, F = x.shape
        C_ = 0 if cov is None else cov.shape[1]
        if F + C_ != Ftot:
            raise ValueError(f"FCNN layer: weight expects {Ftot} inputs, got {F} (+{C_} covariates)")
        ns = _ns()
        dev = Wc.device
        use_bf = sparse and ns == 1 and shadow is not None
        if use_bf:
            tiles_ = ((Bn + 255) // 256) * ((H + 127) // 128)
            nsplit = max(1, min(512 // max(tiles_, 1), ((F + 63) // 64) // 4, 16), L.scp_spmm_fwd_bf16_min_split(F))
        else:
            nsplit = _pick_split(Bn, H, F)
        nsplit = L.scp_gemm_split_count(F, nsplit)
        nslab = nsplit + (1 if C_ else 0)
        slabs = torch.empty((nslab, Bn, H), dtype=torch.float32, device=dev)
        if use_bf:
            _lib.check(L.scp_spmm_fwd_bf16(Bn, H, F, _lib.ptr(x.ptr), _lib.ptr(x.key), _lib.ptr(x.val),
                                           _lib.ptr(x.pk), shadow.data_ptr(), shadow.shape[1], slabs.data_ptr(), H, nsplit,
                                           _lib.stream()), "scp_spmm_fwd_bf16")
            xs = None
        elif sparse:
            _lib.check(L.scp_spmm_fwd(Bn, H, F, _lib.ptr(x.ptr), _lib.ptr(x.key), _lib.ptr(x.val), Wc.data_ptr(),
                                      ldw, slabs.data_ptr(), H, None, nsplit, ns, _lib.stream()), "scp_spmm_fwd")
            xs = None
        else:
            xs = _f32c(x)
            gemm_nt(Bn, H, F, xs, F, 1, Wc, ldw, 1, slabs, H, nsplit=nsplit, ns=ns)
        covc = None
        if C_:
            covc = _f32c(cov)
            gemm_nt(Bn, H, C_, covc, C_, 1, Wc, ldw, 1, slabs, H, ns=ns, b_off=F, c_off=nsplit * Bn * H)
        out = torch.empty((Bn, H), dtype=torch.float32, device=dev)
        xhat = torch.empty((Bn, H), dtype=torch.float32, device=dev)
        rstd = torch.empty((Bn,), dtype=torch.float32, device=dev)
        g_, b_ = _f32c(gamma), _f32c(beta)
        seed_dev = _seed_dev() if p_drop > 0 else None
        _lib.check(L.scp_ln_act_fwd(slabs.data_ptr(), H, Bn * H, nslab, _lib.ptr(_f32c(b)), _lib.ptr(g_), _lib.ptr(b_),
                                    Bn, H, eps, slope, p_drop, seed, seed_dev, xhat.data_ptr(), H, rstd.data_ptr(),
                                    out.data_ptr(), H, _lib.stream()), "scp_ln_act_fwd")
        ctx.sparse = sparse
        ctx.tiles = x if sparse else None
        ctx.save_for_backward(xs, covc, Wc, g_, b_, xhat, rstd)
        ctx.cfg = (slope, p_drop, seed, Bn, H, F, C_, ns, b is not None)
        ctx.seed_dev = seed_dev
        ctx.use_bf = use_bf
        ctx.fused = fused if (use_bf and C_ == 0 and ldw % 4 == 0) else None
        ctx.sink = sink if (use_bf and C_ == 0 and ctx.fused is None) else None
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        xs, cov, W, gamma, beta, xhat, rstd = ctx.saved_tensors
        slope, p_drop, seed, Bn, H, F, C_, ns, has_bias = ctx.cfg
        dout = _f32c(dout)
        dev = W.device
        Ftot = F + C_
        dY = torch.empty((Bn, H), dtype=torch.float32, device=dev)
        dgamma = torch.empty(H, dtype=torch.float32, device=dev) if gamma is not None else None
        dbeta = torch.empty(H, dtype=torch.float32, device=dev) if gamma is not None else None
        dbias = torch.empty(H, dtype=torch.float32, device=dev) if has_bias else None
        ws = torch.empty(L.scp_ln_act_bwd_grid(Bn) * 3 * H, dtype=torch.float32, device=dev)
        _lib.check(L.scp_ln_act_bwd(dout.data_ptr(), H, xhat.data_ptr(), H, rstd.data_ptr(), _lib.ptr(gamma),
                                    _lib.ptr(beta), Bn, H, slope, p_drop, seed, ctx.seed_dev, dY.data_ptr(), H, _lib.ptr(dgamma),
                                    _lib.ptr(dbeta), _lib.ptr(dbias), ws.data_ptr(), _lib.stream()), "scp_ln_act_bwd")
        dx = dW = None
        if ctx.needs_input_grad[2]:
            if ctx.fused is None and ctx.sink is None:
                dW = torch.empty((H, Ftot), dtype=torch.float32, device=dev)
            if ctx.sparse and ctx.use_bf:
                t = ctx.tiles
                ldt = (Bn + 63) // 64 * 64
                dYT = torch.empty((H, ldt), dtype=torch.bfloat16, device=dev)
                _lib.check(L.scp_transpose_bf16(dY.data_ptr(), Bn, H, H, dYT.data_ptr(), ldt, _lib.stream()),
                           "scp_transpose_bf16")
                if ctx.fused is not None:
                    # weight gradient + AdamW step of W in one kernel: dW never reaches HBM, autograd gets no gradient
                    m_, v_, sh_, hp = ctx.fused.begin_step(W)
                    dW = None
                    if row_ld(m_) != row_ld(W) or row_ld(v_) != row_ld(W):
                        raise RuntimeError("fused first layer: AdamW moments and weights differ in layout")
                    tm = _cfg["fused_timer"]
                    if tm is not None:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                    _lib.check(L.scp_spmm_dw_adamw_bf16(Bn, H, F, _lib.ptr(t.ptr), _lib.ptr(t.key), _lib.ptr(t.val), _lib.ptr(t.pk),
                                                        dYT.data_ptr(), ldt, W.data_ptr(), row_ld(W), m_.data_ptr(),
                                                        v_.data_ptr(), sh_.data_ptr(), sh_.shape[1], *hp,
                                                        _lib.stream()), "scp_spmm_dw_adamw_bf16")
                    if tm is not None:
                        e1.record()
                        tm.append((e0, e1, H * Ftot, int(t.nnz) * (8 if t.val is not None else 4) + 2 * H * ldt, "spmm_dw_adamw"))
                elif ctx.sink is not None:
                    # data parallel: dW goes to the exchange buffer as bf16; the optimiser reads it from there
                    dW = None
                    _lib.check(L.scp_spmm_dw_bf16_g16(Bn, H, F, _lib.ptr(t.ptr), _lib.ptr(t.key), _lib.ptr(t.val), _lib.ptr(t.pk),
                                                      dYT.data_ptr(), ldt, ctx.sink.view.data_ptr(), Ftot,
                                                      _lib.stream()), "scp_spmm_dw_bf16_g16")
                    ctx.sink.written = True
                else:
                    _lib.check(L.scp_spmm_dw_bf16(Bn, H, F, _lib.ptr(t.ptr), _lib.ptr(t.key), _lib.ptr(t.val), _lib.ptr(t.pk),
                                                  dYT.data_ptr(), ldt, dW.data_ptr(), Ftot, _lib.stream()),
                               "scp_spmm_dw_bf16")
            elif ctx.sparse:
                t = ctx.tiles
                _lib.check(L.scp_spmm_dw(Bn, H, F, _lib.ptr(t.ptr), _lib.ptr(t.key), _lib.ptr(t.val), dY.data_ptr(), H,
                                         dW.data_ptr(), Ftot, ns, _lib.stream()), "scp_spmm_dw")
            else:
                if C_ == 0:
                    gemm_nt_auto(H, F, Bn, dY, 1, H, xs, 1, F, dW, ns=ns)
                else:
                    gemm_nt(H, F, Bn, dY, 1, H, xs, 1, F, dW, Ftot, ns=ns)
            if C_:
                gemm_nt(H, C_, Bn, dY, 1, H, cov, 1, C_, dW, Ftot, ns=ns, c_off=F)
        if (not ctx.sparse) and ctx.needs_input_grad[0]:
            dx = torch.empty((Bn, F), dtype=torch.float32, device=dev)
            gemm_nt_auto(Bn, F, H, dY, H, 1, W, 1, row_ld(W), dx, ns=ns)
        return dx, None, dW, dbias, dgamma, dbeta, None, None, None, None, None, None, None


def fcnn_layer(x, cov, W, b, gamma, beta, slope, p_drop=0.0, seed=0, eps=1e-5):
    shadow = fused = sink = None
    if isinstance(x, BatchTiles) and _ns() == 1 and W.is_cuda and W.dtype == torch.float32 and row_ld(W) is not None:
        shadow = shadow_of(W)
        if cov is None and torch.is_grad_enabled() and W.requires_grad:
            fused = fused_handle(W)
            sink = grad_sink_of(W) if fused is None else None
    return FcnnLayerFn.apply(x, cov, W, b, gamma, beta, float(slope), float(p_drop), int(seed), float(eps), shadow,
                             fused, sink)


# ------------------------------------------------------------------------------------------------
# Whole encoder on a sparse minibatch: first-layer product(s) on the tile list + the fused tail (enc_tail.hip)
#   x -> Linear(F, H1) -> LN -> act -> Dropout -> Linear(H1, H2) -> LN -> act -> Dropout [-> Linear(H2, N3)]
# for one or two branches (Input_Module.hidden_rep + extra_linear_rep, hidden_lib + its head: model.py:77-92, 124-143).
# Two branches whose first-layer weights are packed (pack_first_layers) run their first-layer product, its weight
# gradient and the AdamW step as ONE launch each over the [H1a + H1b, F] weight block.
# ------------------------------------------------------------------------------------------------
_packs = {}            # id(first weight) -> ([weakref(W), ...], block [sum H, ld], {"shadow": bf16 [sum H, lds] or None})


def _pack_of(W):
    """(home entry, row offset) of a first-layer weight that lives in a block made by home_first_layers, else (None, 0)."""
    for ent in _packs.values():
        off = 0
        ws = [r() for r in ent[0]]
        if any(w is None for w in ws):
            continue
        for w in ws:
            if w is W:
                return (ent, off) if _homed_ok(ent, ws) else (None, 0)
            off += w.shape[0]
    return None, 0


def _homed_ok(ent, ws):
    """Every weight of the entry still is the view of the block it was made (load_state_dict and optimiser steps write in
    place and keep it so; `.data = ...` assignments by the user do not)."""
    block = ent[1]
    off = 0
    for w in ws:
        if w.data_ptr() != block.data_ptr() + 4 * off * block.shape[1] or w.stride(0) != block.shape[1] or w.device != block.device:
            return False
        off += w.shape[0]
    return True


def _rows_adjacent(a, b, row_bytes):
    return b.data_ptr() == a.data_ptr() + a.shape[0] * row_bytes


def home_first_layers(*Ws):
    """Give one first-layer weight -- or the weights of the branches of one encoder, which see the same input (fp32,
    [H_i, F] each, on the GPU) -- the storage the bf16 first-layer kernels stream best: adjacent row blocks of ONE
    [sum H_i, ld] tensor whose rows start on 128-byte boundaries (ld = F rounded up to 32 floats, pad columns zero).  The
    parameters' .data become views [:, :F] of it: values, shapes, names and Parameter identity are unchanged, state_dict /
    load_state_dict / torch.save work as before; only .stride(0) may differ from F.  With adjacent blocks the kernels treat
    the branches as one layer (one product, one weight-gradient + AdamW launch); with aligned rows the AdamW epilogue
    touches whole cache lines (an odd F, or a covariate block behind the features, otherwise costs ~30 % of its rate).
    Idempotent; returns False when the weights cannot be homed.  Optimiser state and bf16 shadows created afterwards
    follow the same layout."""
    if not 1 <= len(Ws) <= ENC_MAX_BRANCHES or len(set(id(w) for w in Ws)) != len(Ws):
        return False
    W0 = Ws[0]
    for w in Ws:
        if w.ndim != 2 or w.shape[1] != W0.shape[1] or not w.is_cuda or w.device != W0.device or w.dtype != torch.float32 \
                or row_ld(w) is None:
            return False
    F = W0.shape[1]
    ld = (F + 31) // 32 * 32
    ent = _packs.get(id(W0))
    if ent is not None:
        ws = [r() for r in ent[0]]
        if len(ws) == len(Ws) and all(a is b for a, b in zip(ws, Ws)) and ent[1].shape[1] == ld and _homed_ok(ent, ws):
            return True
    for k in [k for k, v in _packs.items() if any(r() is None for r in v[0])]:
        del _packs[k]
    for w in Ws:                                       # a weight moves out of the home it had (e.g. a pair re-homed alone)
        for k in [k for k, v in _packs.items() if any(r() is w for r in v[0])]:
            del _packs[k]
    Hs = [w.shape[0] for w in Ws]
    with torch.no_grad():
        block = torch.empty((sum(Hs), ld), dtype=torch.float32, device=W0.device)
        if ld > F:
            block[:, F:].zero_()
        off = 0
        for w, h in zip(Ws, Hs):
            block[off:off + h, :F].copy_(w)
            w.data = block[off:off + h, :F]
            off += h
    for w in Ws:
        _shadows.pop(id(w), None)
    _packs[id(W0)] = ([weakref.ref(w) for w in Ws], block, {"shadow": None})
    return True


def ensure_homed(W):
    """home_first_layers(W) unless W already lives in a home (alone or with the other branches of its encoder)."""
    if _pack_of(W)[0] is not None:
        return True
    return home_first_layers(W)


def pack_first_layers(Wa, Wb):
    """home_first_layers for the two branches of one encoder (hidden_rep and hidden_lib, model.py:77-92)."""
    return home_first_layers(Wa, Wb)


def _new_adam_state(opt, p):
    """Fresh AdamW state of p; homed weights get moments of the same layout (one [sum H, ld] block each for the weights
    of one home, pad columns zero and never touched)."""
    ent, off = _pack_of(p)
    if ent is not None:
        ws = [r() for r in ent[0]]
        if all(len(opt.state[w]) == 0 for w in ws):
            block = ent[1]
            m = torch.zeros_like(block)
            v = torch.zeros_like(block)
            lo = 0
            for q in ws:
                st = opt.state[q]
                st["step"] = 0
                st["exp_avg"] = m[lo:lo + q.shape[0], :q.shape[1]]
                st["exp_avg_sq"] = v[lo:lo + q.shape[0], :q.shape[1]]
                lo += q.shape[0]
            return
    st = opt.state[p]
    st["step"] = 0
    if p.is_contiguous():
        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
    else:                                              # row-padded view whose home is gone: same strides
        ld = row_ld(p)
        if ld is None:
            raise RuntimeError("FusedAdamW needs parameters with contiguous rows")
        st["exp_avg"] = torch.zeros((p.shape[0], ld), dtype=p.dtype, device=p.device)[:, :p.shape[1]]
        st["exp_avg_sq"] = torch.zeros((p.shape[0], ld), dtype=p.dtype, device=p.device)[:, :p.shape[1]]


def _relayout_state(st, p):
    """Moments created before p was homed (or after its home was dropped): copy them into p's row layout."""
    ld = row_ld(p)
    if ld is None:
        raise RuntimeError("FusedAdamW needs parameters with contiguous rows")
    for k in ("exp_avg", "exp_avg_sq"):
        if row_ld(st[k]) != ld:
            t = torch.zeros((p.shape[0], ld), dtype=p.dtype, device=p.device)[:, :p.shape[1]]
            t.copy_(st[k])
            st[k] = t


ENC_MAX_BRANCHES = 4


def enc_branch_supported(x, cov, W1, W2, W3, ln1, ln2):
    """The fused path covers the reference's default encoder shape: sparse minibatch, optionally a covariate block of
    up to 64 columns (cov=['labels'] is the reference constructor's default), two FCNN layers with LayerNorm, hidden widths
    H1 % 4 == 0, H1 <= 1024, H2 <= 64, final Linear with N3 <= 64 outputs."""
    if not isinstance(x, BatchTiles) or not ln1 or not ln2:
        return False
    if cov is not None and not (torch.is_tensor(cov) and cov.is_cuda and cov.ndim == 2 and 0 < cov.shape[1] <= 64
                                and cov.shape[0] == x.shape[0]):
        return False
    if W1.shape[1] != x.shape[1] + (0 if cov is None else cov.shape[1]):
        return False
    H1, H2 = W1.shape[0], W2.shape[0]
    if H1 % 4 or H1 > 1024 or H2 > 64 or W2.shape[1] != H1 or (W3 is not None and (W3.shape[0] > 64 or W3.shape[1] != H2)):
        return False
    if not (W1.is_cuda and W1.dtype == torch.float32 and row_ld(W1) is not None and W1.data_ptr() % 16 == 0):
        return False                                   # W1: contiguous, or a row-padded view (home_first_layers)
    return all(t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0)
               for t in (W2, W3))


def _carve(sizes, dev):
    """One allocation cut into float32 pieces of the given sizes, each 16-byte aligned."""
    offs, o = [], 0
    for n_ in sizes:
        offs.append(o)
        o += (n_ + 3) // 4 * 4
    buf = torch.empty(o + 4, dtype=torch.float32, device=dev)
    return buf, [buf[a:a + n_] for a, n_ in zip(offs, sizes)]


class EncBranchesFn(torch.autograd.Function):
    """cfg: dict(slope, br=[dict(p1, p2, seed1, seed2, eps1, eps2, shadow, fused, sink)] per branch);
    tensors: W1, b1, g1, be1, W2, b2, g2, be2, W3, b3, bn1_w, bn1_b, bn2_w, bn2_b per branch (the last four: affine
    parameters of the BatchNorm1d layers of an FCNN built with batchnorm=True, None otherwise; cfg br[i]["bn"] then holds
    the running tensors, eps, momentum and the mode).  Returns one output per branch."""

    NT = 14

    @staticmethod
    def forward(ctx, x, cfg, *tensors):
        L = _lib.lib()
        nb = len(cfg["br"])
        P = [tensors[EncBranchesFn.NT * i:EncBranchesFn.NT * i + EncBranchesFn.NT] for i in range(nb)]
        Bg, F = x.shape                            # rows of the first-layer product
        # feature-sharded data parallelism (distributed.FirstLayerShard): x holds the GLOBAL batch (every rank's slice padded
        # to a common length) restricted to this rank's columns, W1 is this rank's column block; the ranks' partial products
        # are summed and the encoder goes on with this rank's own rows
        shd = getattr(x, "shard", None)
        Bn = Bg if shd is None else shd.n_local    # rows the tail works on
        dev = P[0][0].device
        ns = _ns()
        H1s = [p_[0].shape[0] for p_ in P]
        H2s = [p_[4].shape[0] for p_ in P]
        N3s = [0 if p_[8] is None else p_[8].shape[0] for p_ in P]
        cov = cfg.get("cov")                        # [B, C] covariate block concatenated to the input (model.py:128), or None
        cov = None if cov is None else _f32c(cov)
        Cn = 0 if cov is None else cov.shape[1]
        Ft = F + Cn                                 # columns of the first-layer weights
        if shd is not None and (Cn or Bn <= 0 or Bg != shd.world * shd.width):
            raise ValueError("feature-sharded first layer: no covariate block, a non-empty slice on every rank and a batch "
                             "of world x width rows are required")
        for p_ in P:
            if p_[0].shape[1] != Ft:
                raise ValueError(f"encoder: first layer expects {p_[0].shape[1]} inputs, the minibatch has {F} features"
                                 f" (+{Cn} covariates)")
        ldws = [row_ld(p_[0]) for p_ in P]             # row strides of the first-layer weights (Ft, or padded: homed)
        shadows = [b_["shadow"] for b_ in cfg["br"]]
        use_bf = ns == 1 and all(sh is not None for sh in shadows)
        # one first-layer launch for both branches when their weights (and bf16 shadows) are adjacent row blocks
        merged = (nb == 2 and use_bf and ldws[0] == ldws[1] and _rows_adjacent(P[0][0], P[1][0], 4 * ldws[0])
                  and shadows[0].shape[1] == shadows[1].shape[1]
                  and _rows_adjacent(shadows[0], shadows[1], 2 * shadows[0].shape[1]))
        groups = [list(range(nb))] if merged else [[i] for i in range(nb)]
        # fp32-equivalent mode with implicit-ones data (binarised peaks): the sparse operand is exact in bf16, so the
        # product needs only the three terms of the weights -- three launches of the one-term bf16 kernel (and one of its
        # weight-gradient kernel over the minibatch repeated three times) instead of the general 6-product kernel
        x3 = (ns == 3 and not use_bf and x.val is None and shd is None and 3 * Bg <= 32768 and Bg <= 16384
              and not _os.environ.get("SCPAIR_FP32_GENERAL"))
        train = any(ctx.needs_input_grad)
        bns = [b_.get("bn") for b_ in cfg["br"]]
        has_bn = bns[0] is not None
        if any((b_ is not None) != has_bn or (has_bn and b_["train"] != bns[0]["train"]) for b_ in bns):
            raise ValueError("encoder: the branches of one call share the BatchNorm mode")
        keep_saves = train or (has_bn and bns[0]["train"])      # batch statistics: the segments hand over through them
        nouts = [n3 if n3 > 0 else h2 for n3, h2 in zip(N3s, H2s)]
        # one allocation: slabs of every first-layer launch | outputs | saved-for-backward pieces
        nsplits, slab_sizes = [], []
        for g in groups:
            Hg = sum(H1s[i] for i in g)
            if use_bf or x3:
                tiles_ = ((Bg + 255) // 256) * ((Hg + 127) // 128)
                nsp = max(1, min(_cfg.get("fwd_wgs", 512) // max(tiles_, 1), ((F + 63) // 64) // 4, 16), L.scp_spmm_fwd_bf16_min_split(F))
            else:
                nsp = _pick_split(Bg, Hg, F)
            nsp = L.scp_gemm_split_count(F, nsp)
            nsplits.append(nsp)
            slab_sizes.append((3 if x3 else 1) * nsp * Bg * Hg)
        sizes = list(slab_sizes) + [Bn * n for n in nouts]
        if keep_saves:
            for i in range(nb):
                sizes += [Bn * H1s[i], Bn, Bn * H2s[i], Bn]
        n_before_bn = len(sizes)
        if has_bn:
            sizes += [L.scp_enc_tail_bn_ws_floats(H1s[i], H2s[i]) for i in range(nb)]
        buf, pieces = _carve(sizes, dev)
        slabs = pieces[:len(groups)]
        outs = [pieces[len(groups) + i].view(Bn, nouts[i]) for i in range(nb)]
        saves = [pieces[len(groups) + nb + 4 * i:len(groups) + nb + 4 * i + 4] if keep_saves else (None,) * 4 for i in range(nb)]
        bn_wss = pieces[n_before_bn:n_before_bn + nb] if has_bn else [None] * nb
        stream = _lib.stream()
        slab_of = {}
        shd_keep = []
        for gi, g in enumerate(groups):
            Hg = sum(H1s[i] for i in g)
            W1 = P[g[0]][0]
            if use_bf:
                sh = shadows[g[0]]
                _lib.check(L.scp_spmm_fwd_bf16(Bg, Hg, F, _lib.ptr(x.ptr), _lib.ptr(x.key), _lib.ptr(x.val), _lib.ptr(x.pk),
                                               sh.data_ptr(),
                                               sh.shape[1], slabs[gi].data_ptr(), Hg, nsplits[gi], stream),
                           "scp_spmm_fwd_bf16")
            elif x3:
                ldsh = (F + 63) // 64 * 64
                sh3 = torch.empty((3, Hg, ldsh), dtype=torch.bfloat16, device=dev)
                _lib.check(L.scp_split3_bf16(W1.data_ptr(), Hg, F, ldws[g[0]], sh3.data_ptr(), ldsh, Hg * ldsh, stream),
                           "scp_split3_bf16")
                for t3 in range(3):
                    _lib.check(L.scp_spmm_fwd_bf16(Bg, Hg, F, _lib.ptr(x.ptr), _lib.ptr(x.key), None, None,
                                                   sh3.data_ptr() + 2 * t3 * Hg * ldsh, ldsh,
                                                   slabs[gi].data_ptr() + 4 * t3 * nsplits[gi] * Bg * Hg, Hg, nsplits[gi],
                                                   stream), "scp_spmm_fwd_bf16")
                nsplits[gi] *= 3                  # the tail sums all slabs
            else:
                _lib.check(L.scp_spmm_fwd(Bg, Hg, F, _lib.ptr(x.ptr), _lib.ptr(x.key), _lib.ptr(x.val), W1.data_ptr(),
                                          ldws[g[0]], slabs[gi].data_ptr(), Hg, None, nsplits[gi], ns, stream), "scp_spmm_fwd")
            base, stride_, nsl = slabs[gi].data_ptr(), Bg * Hg, nsplits[gi]
            if shd is not None:
                # this rank's partial product of the global batch -> summed over the ranks; the tail reads its own rows
                if nsl > 1:
                    y = torch.empty((Bg, Hg), dtype=torch.float32, device=dev)
                    slab_reduce(slabs[gi], nsl, Bg, Hg, y)
                else:
                    y = slabs[gi][:Bg * Hg].view(Bg, Hg)
                y = shd.sync.reduce_scatter_rows(y)          # [width, Hg]: this rank's rows, summed over the ranks
                shd_keep.append(y)
                base, stride_, nsl = y.data_ptr(), 0, 1
            col = 0
            for i in g:
                slab_of[i] = (base + 4 * col, stride_, Hg, nsl)
                col += H1s[i]
        any_drop = any(b_["p1"] > 0 or b_["p2"] > 0 for b_ in cfg["br"])
        seed_dev = _seed_dev() if any_drop else None
        arr = (_lib.EncTail * nb)()
        for i in range(nb):
            W1, b1, g1, be1, W2, b2, g2, be2, W3, b3, bn1w, bn1b, bn2w, bn2b = P[i]
            c = cfg["br"][i]
            m = arr[i]
            m.B, m.H1, m.H2, m.N3 = Bn, H1s[i], H2s[i], N3s[i]
            if has_bn:
                bn = bns[i]
                m.bn, m.bn_train = 1, int(bn["train"])
                m.bn1_w, m.bn1_b, m.bn2_w, m.bn2_b = _lib.ptr(bn1w), _lib.ptr(bn1b), _lib.ptr(bn2w), _lib.ptr(bn2b)
                m.bn1_rm, m.bn1_rv, m.bn2_rm, m.bn2_rv = (_lib.ptr(bn["rm1"]), _lib.ptr(bn["rv1"]), _lib.ptr(bn["rm2"]),
                                                          _lib.ptr(bn["rv2"]))
                m.bn_eps1, m.bn_eps2, m.bn_mom1, m.bn_mom2 = bn["eps1"], bn["eps2"], bn["mom1"], bn["mom2"]
                m.bn_ws = bn_wss[i].data_ptr()
            m.b1, m.g1, m.be1 = _lib.ptr(b1), g1.data_ptr(), be1.data_ptr()
            m.W2, m.b2, m.g2, m.be2 = W2.data_ptr(), b2.data_ptr(), g2.data_ptr(), be2.data_ptr()
            m.W3, m.b3 = _lib.ptr(W3), _lib.ptr(b3)
            m.eps1, m.eps2, m.slope, m.p1, m.p2 = c["eps1"], c["eps2"], cfg["slope"], c["p1"], c["p2"]
            m.seed1, m.seed2, m.seed_dev = c["seed1"], c["seed2"], seed_dev
            xh1, r1, xh2, r2 = saves[i]
            m.xhat1, m.rstd1, m.xhat2, m.rstd2 = _lib.ptr(xh1), _lib.ptr(r1), _lib.ptr(xh2), _lib.ptr(r2)
            m.slabs, m.slab_stride, m.ld_slab, m.nslab = slab_of[i]
            m.out, m.ldo = outs[i].data_ptr(), nouts[i]
            if Cn:
                m.cov, m.ldcov, m.C = cov.data_ptr(), Cn, Cn
                m.W1c, m.ldw1c = W1.data_ptr() + 4 * F, ldws[i]
        _lib.check(L.scp_enc_tail_fwd(arr, nb, stream), "scp_enc_tail_fwd")
        if train:
            ctx.arr, ctx.tiles, ctx.buf, ctx.cfg = arr, x, buf, cfg
            ctx.cov = cov
            ctx.save_for_backward(*[t for t in tensors if t is not None])
            ctx.present = [t is not None for t in tensors]
            ctx.dims = (Bn, F, H1s, H2s, N3s, ns, use_bf, groups, Cn, ldws)
            ctx.shd, ctx.x3 = shd, x3
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        L = _lib.lib()
        it = iter(ctx.saved_tensors)
        tensors = [next(it) if pr else None for pr in ctx.present]
        Bn, F, H1s, H2s, N3s, ns, use_bf, groups, Cn, ldws = ctx.dims
        Ft = F + Cn
        nb = len(H1s)
        P = [tensors[EncBranchesFn.NT * i:EncBranchesFn.NT * i + EncBranchesFn.NT] for i in range(nb)]
        cfg, arr, t = ctx.cfg, ctx.arr, ctx.tiles
        dev = P[0][0].device
        stream = _lib.stream()
        shd = ctx.shd
        Bg = Bn if shd is None else shd.world * shd.width      # rows of the first-layer weight-gradient product
        ldt = (Bn + 63) // 64 * 64 if shd is None else shd.width
        # one allocation for every small gradient and the workspaces; the first-layer gradient operand separately
        has_bn = cfg["br"][0].get("bn") is not None
        sizes = []
        for i in range(nb):
            nws = L.scp_enc_tail_ws_floats(Bn, H1s[i], H2s[i], N3s[i], Cn)
            sizes += [H1s[i], H1s[i], H1s[i], H2s[i], H2s[i], H2s[i], N3s[i] * H2s[i], N3s[i], H2s[i] * H1s[i], nws,
                      0 if use_bf else Bn * H1s[i]]
            sizes += [H1s[i], H1s[i], H2s[i], H2s[i], Bn * H1s[i]] if has_bn else [0, 0, 0, 0, 0]
        gb, pieces = _carve(sizes, dev)
        need_w1 = [ctx.needs_input_grad[2 + EncBranchesFn.NT * i] for i in range(nb)]
        dYT = {}
        if use_bf:
            for g in groups:
                Hg = sum(H1s[i] for i in g)
                tbuf = (torch.empty if shd is None else torch.zeros)((Hg, ldt), dtype=torch.bfloat16, device=dev)
                row = 0
                for i in g:
                    dYT[i] = (tbuf, row)
                    row += H1s[i]
        # ---- how the first-layer weight gradient of every branch is produced: decided before the tail's backward runs,
        # because with a covariate block that kernel's reduction writes (or consumes) the gradient of the weight columns
        # F .. F+C-1.  A unit = the branches one weight-gradient launch covers; mode 'fused': AdamW step in the kernel's
        # epilogue, 'sink': bf16 into the exchange buffer, 'plain': fp32 gradient for autograd ----
        plan = []               # (branches, mode, payload)
        if not use_bf:
            for i in range(nb):
                if need_w1[i]:
                    plan.append(([i], "plain32", torch.empty((H1s[i], Ft), dtype=torch.float32, device=dev)))
        else:
            for g in groups:
                fuse_ok = ldws[g[0]] % 4 == 0          # rows of W / moments 16-byte aligned (any F: homed weights are)
                if not any(need_w1[i] for i in g):
                    continue
                units = [g]
                if len(g) == 2:
                    fz = [cfg["br"][i]["fused"] if fuse_ok else None for i in g]
                    sk = [cfg["br"][i]["sink"] for i in g]
                    if fz[0] is not None and fz[1] is not None:
                        if not _fused_pair_ok(fz[0], fz[1], P[g[0]][0], P[g[1]][0]):
                            units = [[g[0]], [g[1]]]
                    elif fz[0] is not None or fz[1] is not None or (sk[0] is None) != (sk[1] is None) \
                            or (sk[0] is not None and not _rows_adjacent(sk[0].view, sk[1].view, 2 * F)) \
                            or not all(need_w1[i] for i in g):
                        units = [[g[0]], [g[1]]]
                for u in units:
                    if not need_w1[u[0]]:
                        continue
                    fz = cfg["br"][u[0]]["fused"] if fuse_ok else None
                    sk = cfg["br"][u[0]]["sink"] if fz is None else None
                    if fz is not None:
                        st = fz.begin_step(P[u[0]][0])
                        for i in u[1:]:
                            cfg["br"][i]["fused"].begin_step(P[i][0])
                        plan.append((u, "fused", st))
                    elif sk is not None:
                        plan.append((u, "sink", sk))
                    else:
                        Hu = sum(H1s[i] for i in u)
                        plan.append((u, "plain", torch.empty((Hu, Ft), dtype=torch.float32, device=dev)))
        keep = []
        grads = []
        for i in range(nb):
            dg1, dbe1, db1, dg2, dbe2, db2, dW3, db3, dW2, ws, dY1, dbn1w, dbn1b, dbn2w, dbn2b, bn_dy1 = pieces[16 * i:16 * i + 16]
            W1, b1 = P[i][0], P[i][1]
            nout = N3s[i] if N3s[i] else H2s[i]
            do = douts[i]
            do = torch.zeros((Bn, nout), dtype=torch.float32, device=dev) if do is None else _f32c(do)
            keep.append(do)
            m = arr[i]
            m.dOut, m.lddo = do.data_ptr(), do.shape[1]
            if shd is not None and not use_bf:     # rows padded to the common slice length for the all-gather
                dY1 = torch.zeros((shd.width, H1s[i]), dtype=torch.float32, device=dev)
                keep.append(dY1)
            m.dY1, m.ldy = (None, 0) if use_bf else (dY1.data_ptr(), H1s[i])
            if use_bf:
                tbuf, row = dYT[i]
                m.dYT, m.ldt = tbuf.data_ptr() + 2 * row * ldt, ldt
            else:
                m.dYT, m.ldt = None, 0
            m.ws = ws.data_ptr()
            m.dg1, m.dbe1, m.db1 = dg1.data_ptr(), dbe1.data_ptr(), _lib.ptr(db1 if b1 is not None else None)
            m.dW2, m.dg2, m.dbe2, m.db2 = dW2.data_ptr(), dg2.data_ptr(), dbe2.data_ptr(), db2.data_ptr()
            m.dW3, m.db3 = _lib.ptr(dW3 if N3s[i] else None), _lib.ptr(db3 if N3s[i] else None)
            m.dW1c, m.lddw1c, m.adam = None, 0, 0
            gbn = [None] * 4
            if has_bn:
                m.bn_dy1 = bn_dy1.data_ptr()
                has_aff = [P[i][10 + k] is not None for k in range(4)]
                m.dbn1_w, m.dbn1_b = _lib.ptr(dbn1w if has_aff[0] else None), _lib.ptr(dbn1b if has_aff[1] else None)
                m.dbn2_w, m.dbn2_b = _lib.ptr(dbn2w if has_aff[2] else None), _lib.ptr(dbn2b if has_aff[3] else None)
                gbn = [t_ if ok_ else None for t_, ok_ in zip((dbn1w, dbn1b, dbn2w, dbn2b), has_aff)]
            grads.append([None, db1 if b1 is not None else None, dg1, dbe1, dW2.view(H2s[i], H1s[i]), db2, dg2, dbe2,
                          dW3.view(N3s[i], H2s[i]) if N3s[i] else None, db3 if N3s[i] else None] + gbn)
        if Cn:
            # the covariate columns of each branch's first-layer gradient (or their AdamW step) come out of the tail
            for u, mode, payload in plan:
                row_ = 0
                for i in u:
                    m = arr[i]
                    if mode in ("plain", "plain32"):
                        m.dW1c, m.lddw1c = payload.data_ptr() + 4 * (row_ * Ft + F), Ft
                    elif mode == "fused":
                        fz_i = cfg["br"][i]["fused"]
                        p_i = fz_i.param
                        st_i = fz_i.opt.state[p_i]
                        sh_i = shadow_of(p_i)
                        lr, wd, b1_, b2_, eps_, bc1, bc2, gs, bc_dev = payload[3]
                        if row_ld(st_i["exp_avg"]) != ldws[i] or row_ld(st_i["exp_avg_sq"]) != ldws[i]:
                            raise RuntimeError("fused first layer: AdamW moments and weights differ in layout")
                        m.adam = 1
                        m.adam_m, m.adam_v = st_i["exp_avg"].data_ptr() + 4 * F, st_i["exp_avg_sq"].data_ptr() + 4 * F
                        m.adam_sh, m.adam_ldsh = sh_i.data_ptr() + 2 * F, sh_i.shape[1]
                        m.lr, m.wd, m.beta1, m.beta2, m.adam_eps = lr, wd, b1_, b2_, eps_
                        m.bc1, m.bc2, m.gscale, m.bc_dev = bc1, bc2, gs, bc_dev
                    else:
                        raise RuntimeError("encoder with a covariate block: the bf16 gradient sink is not supported")
                    row_ += H1s[i]
        _lib.check(L.scp_enc_tail_bwd(arr, nb, stream), "scp_enc_tail_bwd")
        dy1_of = {}
        if shd is not None:
            # feature-sharded: every rank's rows of dY, weighted by its share of the step's label elements, are gathered;
            # the gradient of this rank's column block over the GLOBAL batch is then complete here (no gradient exchange)
            if use_bf:
                for g in groups:
                    tbuf = dYT[g[0]][0]
                    if shd.share is not None:
                        tbuf.mul_(shd.share.to(tbuf.dtype))
                    Hg_ = tbuf.shape[0]
                    full = shd.sync.gather_rows(tbuf.view(-1)).view(shd.world, Hg_, ldt).permute(1, 0, 2).reshape(Hg_, Bg)
                    row = 0
                    for i in g:
                        dYT[i] = (full, row)
                        row += H1s[i]
                ldt = Bg
            else:
                for i in range(nb):
                    d_ = keep[2 * i + 1]
                    if shd.share is not None:
                        d_.mul_(shd.share)
                    dy1_of[i] = shd.sync.gather_rows(d_.view(-1)).view(Bg, H1s[i])
        # ---- first-layer weight gradient (and, where registered, the AdamW step in its epilogue) -------------------
        tp, tk, tv, tpk = _lib.ptr(t.ptr), _lib.ptr(t.key), _lib.ptr(t.val), _lib.ptr(t.pk)
        tiles3 = None
        Bn = Bg
        for u, mode, payload in plan:
            Hu = sum(H1s[i] for i in u)
            W1 = P[u[0]][0]
            if mode == "plain32":
                i = u[0]
                dy1_ptr = dy1_of[i].data_ptr() if shd is not None else arr[i].dY1
                if ctx.x3:
                    # dW = dY^T X with dY in three bf16 terms and X exact: the one-term kernel over the minibatch three times
                    if tiles3 is None:
                        tiles3 = BatchTiles(t.csr, t.rows.repeat(3), want_lib=False, capacity=3 * int(t.nnz))
                    ldt3 = (3 * Bn + 63) // 64 * 64
                    dyt3 = torch.empty((H1s[i], ldt3), dtype=torch.bfloat16, device=dev)
                    keep.append(dyt3)
                    _lib.check(L.scp_split3_transpose_bf16(dy1_ptr, Bn, H1s[i], H1s[i], dyt3.data_ptr(), ldt3, stream),
                               "scp_split3_transpose_bf16")
                    _lib.check(L.scp_spmm_dw_bf16(3 * Bn, H1s[i], F, _lib.ptr(tiles3.ptr), _lib.ptr(tiles3.key), None, None,
                                                  dyt3.data_ptr(), ldt3, payload.data_ptr(), Ft, stream), "scp_spmm_dw_bf16")
                    grads[i][0] = payload
                    continue
                _lib.check(L.scp_spmm_dw(Bn, H1s[i], F, tp, tk, tv, dy1_ptr, H1s[i], payload.data_ptr(), Ft, ns, stream),
                           "scp_spmm_dw")
                grads[i][0] = payload
                continue
            tbuf, row = dYT[u[0]]
            dyt_ptr = tbuf.data_ptr() + 2 * row * ldt
            if mode == "fused":
                m_, v_, sh_, hp = payload
                tm = _cfg["fused_timer"]
                if tm is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                if row_ld(m_) != ldws[u[0]] or row_ld(v_) != ldws[u[0]]:
                    raise RuntimeError("fused first layer: AdamW moments and weights differ in layout")
                _lib.check(L.scp_spmm_dw_adamw_bf16(Bn, Hu, F, tp, tk, tv, tpk, dyt_ptr, ldt, W1.data_ptr(), ldws[u[0]],
                                                    m_.data_ptr(), v_.data_ptr(), sh_.data_ptr(), sh_.shape[1],
                                                    *hp, stream), "scp_spmm_dw_adamw_bf16")
                if tm is not None:
                    e1.record()
                    tm.append((e0, e1, Hu * F, int(t.nnz) * (8 if t.val is not None else 4) + 2 * Hu * ldt, "spmm_dw_adamw"))
            elif mode == "sink":
                _lib.check(L.scp_spmm_dw_bf16_g16(Bn, Hu, F, tp, tk, tv, tpk, dyt_ptr, ldt, payload.view.data_ptr(), F, stream),
                           "scp_spmm_dw_bf16_g16")
                for i in u:
                    cfg["br"][i]["sink"].written = True
            else:
                _lib.check(L.scp_spmm_dw_bf16(Bn, Hu, F, tp, tk, tv, tpk, dyt_ptr, ldt, payload.data_ptr(), Ft, stream),
                           "scp_spmm_dw_bf16")
                row_ = 0
                for i in u:
                    grads[i][0] = payload[row_:row_ + H1s[i]]
                    row_ += H1s[i]
        ctx.buf = ctx.tiles = ctx.arr = ctx.cov = ctx.shd = None
        del tiles3
        flat = [None, None]
        for i in range(nb):
            flat += grads[i]
        return tuple(flat)


def _fused_pair_ok(fa, fb, Wa, Wb):
    """Both weights of a packed pair are stepped by the same optimiser with the same hyper-parameters and step count,
    and their moments and bf16 shadows are adjacent row blocks: one fused weight-gradient + AdamW launch covers both."""
    pa, pb, opt = fa.param, fb.param, fa.opt
    if pa is None or pb is None or opt is None or fb.opt is not opt or fa.group is not fb.group:
        return False
    if len(opt.state[pa]) == 0 and len(opt.state[pb]) == 0:
        _new_adam_state(opt, pa)
    sa, sb = opt.state[pa], opt.state[pb]
    if len(sa) == 0 or len(sb) == 0 or sa["step"] != sb["step"]:
        return False
    ld = row_ld(Wa)
    if ld is None or any(row_ld(t) != ld for t in (Wb, sa["exp_avg"], sb["exp_avg"], sa["exp_avg_sq"], sb["exp_avg_sq"])):
        return False
    sha, shb = shadow_of(pa), shadow_of(pb)
    return (_rows_adjacent(sa["exp_avg"], sb["exp_avg"], 4 * ld) and _rows_adjacent(sa["exp_avg_sq"], sb["exp_avg_sq"], 4 * ld)
            and sha.shape[1] == shb.shape[1] and _rows_adjacent(sha, shb, 2 * sha.shape[1]))


def enc_branches(x, branches, slope, seeds=None, cov=None):
    """branches: list of dicts with W1, b1, g1, be1, W2, b2, g2, be2, W3, b3 (W3 / b3 / b1 may be None) and optional
    p1, p2, eps1, eps2; for an FCNN built with batchnorm=True additionally bn1_w, bn1_b, bn2_w, bn2_b (affine parameters or
    None) and bn = dict(rm1, rv1, rm2, rv2: running tensors or None, eps1, eps2, mom1, mom2, train) (see bn_spec()).  seeds: optional list of (seed1, seed2) per branch (tests); drawn from next_seed() otherwise.
    cov: optional [B, C] covariate block the reference concatenates to the input (model.py:128); W1 is then [H1, F + C].
    Returns the list of branch outputs."""
    if not 1 <= len(branches) <= ENC_MAX_BRANCHES:
        raise ValueError("enc_branches: 1..%d branches" % ENC_MAX_BRANCHES)
    cfg = {"slope": float(slope), "br": [], "cov": cov}
    tensors = []
    grad_on = torch.is_grad_enabled()
    for i, b in enumerate(branches):
        W1 = b["W1"]
        p1, p2 = float(b.get("p1", 0.0)), float(b.get("p2", 0.0))
        shadow = fused = sink = None
        if _ns() == 1:
            shadow = shadow_of(W1)
            if grad_on and W1.requires_grad:
                fused = fused_handle(W1)
                sink = grad_sink_of(W1) if (fused is None and cov is None) else None
        if seeds is not None:
            s1, s2 = seeds[i]
        else:
            s1, s2 = (next_seed() if p1 > 0 else 0), (next_seed() if p2 > 0 else 0)
        cfg["br"].append(dict(p1=p1, p2=p2, seed1=int(s1), seed2=int(s2), eps1=float(b.get("eps1", 1e-5)),
                              eps2=float(b.get("eps2", 1e-5)), shadow=shadow, fused=fused, sink=sink, bn=b.get("bn")))
        tensors += [W1, b.get("b1"), b["g1"], b["be1"], b["W2"], b["b2"], b["g2"], b["be2"], b.get("W3"), b.get("b3"),
                    b.get("bn1_w"), b.get("bn1_b"), b.get("bn2_w"), b.get("bn2_b")]
    return list(EncBranchesFn.apply(x, cfg, *tensors))


def bn_spec(bn1, bn2, training):
    """The `bn` entry of an enc_branches() branch for two nn.BatchNorm1d modules, with nn.BatchNorm1d's bookkeeping:
    num_batches_tracked advances in training mode, momentum None means the cumulative moving average."""
    out = {"train": bool(training) or bn1.running_mean is None or bn2.running_mean is None}
    for k, bn in (("1", bn1), ("2", bn2)):
        if training and bn.track_running_stats and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
        mom = bn.momentum
        if mom is None:
            mom = 1.0 / float(bn.num_batches_tracked) if (training and bn.num_batches_tracked is not None) else 0.0
        track = bn.track_running_stats and bn.running_mean is not None
        out["rm" + k] = bn.running_mean if track else None
        out["rv" + k] = bn.running_var if track else None
        out["eps" + k], out["mom" + k] = float(bn.eps), float(mom)
    return out


def enc_branch(x, W1, b1, g1, be1, W2, b2, g2, be2, W3, b3, slope, p1=0.0, p2=0.0, eps1=1e-5, eps2=1e-5, seeds=None):
    return enc_branches(x, [dict(W1=W1, b1=b1, g1=g1, be1=be1, W2=W2, b2=b2, g2=g2, be2=be2, W3=W3, b3=b3, p1=p1, p2=p2,
                                 eps1=eps1, eps2=eps2)], slope, None if seeds is None else [seeds])[0]


# ------------------------------------------------------------------------------------------------
# BatchNorm1d (+ the layer's dropout), utils.py:61-63
# ------------------------------------------------------------------------------------------------
class BatchNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, training, momentum, eps, p_drop, seed):
        L = _lib.lib()
        x = _f32c(x)
        _lib.require_cuda(x)
        B, H = x.shape
        y = torch.empty_like(x)
        mean = _empty((H,), x)
        rstd = _empty((H,), x)
        seed_dev = _seed_dev() if p_drop > 0 else None
        _lib.check(L.scp_bn_fwd(x.data_ptr(), H, B, H, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(run_mean),
                                _lib.ptr(run_var), int(training), float(momentum), float(eps), float(p_drop), int(seed),
                                seed_dev, mean.data_ptr(), rstd.data_ptr(), y.data_ptr(), H, _lib.stream()), "scp_bn_fwd")
        ctx.seed_dev = seed_dev
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.cfg = (bool(training), float(p_drop), int(seed))
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, gamma, mean, rstd = ctx.saved_tensors
        training, p_drop, seed = ctx.cfg
        dy = _f32c(dy)
        B, H = x.shape
        dx = torch.empty_like(x)
        dg = _empty((H,), x) if gamma is not None else None
        db = _empty((H,), x) if gamma is not None else None
        _lib.check(L.scp_bn_bwd(dy.data_ptr(), H, x.data_ptr(), H, B, H, _lib.ptr(gamma), mean.data_ptr(),
                                rstd.data_ptr(), int(training), p_drop, seed, ctx.seed_dev, dx.data_ptr(), H, _lib.ptr(dg),
                                _lib.ptr(db), _lib.stream()), "scp_bn_bwd")
        return dx, dg, db, None, None, None, None, None, None, None


# ---- BatchNorm under data parallelism: batch statistics over the GLOBAL batch -------------------------------------
# The trainers slice every global batch by rank; with per-rank statistics a BatchNorm layer would see a different batch
# than the single-process run.  set_bn_sync(GradSync) makes training-mode batch_norm() combine the ranks' statistics
# (each rank's two-pass mean / M2 and row count, all-gathered and merged in rank order with the parallel-variance
# formula) and its backward exchange the two per-feature sums the input gradient needs.
_bn_sync = {"sync": None, "w": None}


def set_bn_sync(sync):
    """`sync`: a distributed.GradSync (every rank must then run the same BatchNorm layers in the same order) or None."""
    _bn_sync["sync"], _bn_sync["w"] = sync, None


def bn_sync_active():
    return _bn_sync["sync"] is not None


def set_bn_sync_weight(w):
    """This rank's share n_r / n_total of the step's label elements (GradSync.share_of), set before backward(): the
    ranks' losses enter the global-batch loss with these weights, and so do their terms in BatchNorm's backward sums."""
    _bn_sync["w"] = w


def combine_bn_stats(rows, H):
    """rows: [world, 2H + 1] = per rank (mean[H] | M2[H] | n).  Returns the global (mean, biased variance, N):
    M2 = sum_r M2_r + sum_r n_r (mean_r - mean)^2.  Pure tensor arithmetic, identical on every rank."""
    n = rows[:, 2 * H].double()
    N = n.sum()
    mr, m2r = rows[:, :H].double(), rows[:, H:2 * H].double()
    mean = (mr * n[:, None]).sum(0) / N
    m2 = m2r.sum(0) + (n[:, None] * (mr - mean) ** 2).sum(0)
    return mean.float(), (m2 / N).float(), N.float()


class SyncBatchNormFn(torch.autograd.Function):
    """BatchNormFn with the statistics of the global batch.  forward: local (mean, M2, n) -> all-gather -> global mean /
    variance -> the normalisation kernel run with them (its 'given statistics' mode) -> running statistics updated with
    the global values.  backward: the kernel's given-statistics mode yields g*rstd*dy' and this rank's sums
    (sum dy', sum dy'*xhat); the sums are weighted by this rank's loss share w_r, all-reduced, and the two mean terms of
    the training-mode gradient are subtracted:  dx = g*rstd*(dy' - (S1 + xhat*S2) / (N*w_r))."""

    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, momentum, eps, p_drop, seed):
        L = _lib.lib()
        sync = _bn_sync["sync"]
        x = _f32c(x)
        _lib.require_cuda(x)
        B, H = x.shape
        mean_r = x.mean(0)
        m2_r = (x - mean_r).square().sum(0)
        rows = sync.gather_rows(torch.cat([mean_r, m2_r, x.new_full((1,), float(B))]))
        mean, var, N = combine_bn_stats(rows, H)
        y = torch.empty_like(x)
        smean, rstd = _empty((H,), x), _empty((H,), x)
        seed_dev = _seed_dev() if p_drop > 0 else None
        _lib.check(L.scp_bn_fwd(x.data_ptr(), H, B, H, _lib.ptr(gamma), _lib.ptr(beta), mean.data_ptr(), var.data_ptr(), 0,
                                0.0, float(eps), float(p_drop), int(seed), seed_dev, smean.data_ptr(), rstd.data_ptr(),
                                y.data_ptr(), H, _lib.stream()), "scp_bn_fwd")
        if run_mean is not None:
            run_mean.mul_(1.0 - momentum).add_(mean, alpha=momentum)
            run_var.mul_(1.0 - momentum).add_(var * (N / torch.clamp(N - 1.0, min=1.0)), alpha=momentum)
        ctx.seed_dev, ctx.sync = seed_dev, sync
        ctx.save_for_backward(x, gamma, smean, rstd, N)
        ctx.cfg = (float(p_drop), int(seed))
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, gamma, mean, rstd, N = ctx.saved_tensors
        p_drop, seed = ctx.cfg
        dy = _f32c(dy)
        B, H = x.shape
        dx = torch.empty_like(x)
        dg, db = _empty((H,), x), _empty((H,), x)
        _lib.check(L.scp_bn_bwd(dy.data_ptr(), H, x.data_ptr(), H, B, H, _lib.ptr(gamma), mean.data_ptr(), rstd.data_ptr(), 0,
                                p_drop, seed, ctx.seed_dev, dx.data_ptr(), H, dg.data_ptr(), db.data_ptr(), _lib.stream()),
                   "scp_bn_bwd")
        w = _bn_sync["w"]
        if w is None:
            raise RuntimeError("synchronised BatchNorm backward needs this rank's loss share: call "
                               "ops.set_bn_sync_weight(GradSync.share_of(n_valid, device)) before backward()")
        S = ctx.sync._sum(torch.cat([db, dg]) * w)
        # A rank whose shard holds rows but no valid label element has share w = 0: its gradients are multiplied by 0 in
        # the exchange anyway, so the mean terms are dropped there instead of dividing by zero (0 * NaN would poison
        # the all-reduced gradients of every rank).  The gradient the other ranks' losses send through this rank's rows
        # of the global statistics is lost in that (degenerate) step.
        inv = torch.where(w > 0, 1.0 / (N * torch.clamp(w, min=1e-30)), torch.zeros_like(N * w))
        coef = rstd * inv if gamma is None else gamma * rstd * inv
        xhat = (x - mean) * rstd
        dx -= coef * (S[:H] + xhat * S[H:])
        if gamma is None:
            return dx, None, None, None, None, None, None, None, None
        return dx, dg, db, None, None, None, None, None, None


def batch_norm(x, bn, training, p_drop=0.0, seed=0):
    """nn.BatchNorm1d `bn` applied by the HIP kernel (batch statistics + in-place running-statistics update when
    training and tracking, running statistics otherwise), followed by dropout p_drop."""
    use_batch = training or bn.running_mean is None
    if training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    mom = bn.momentum
    if mom is None:                               # cumulative moving average
        mom = 1.0 / float(bn.num_batches_tracked) if (training and bn.num_batches_tracked is not None) else 0.0
    if training and _bn_sync["sync"] is not None:
        return SyncBatchNormFn.apply(x, bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
                                     bn.running_var if bn.track_running_stats else None, mom, bn.eps, p_drop, seed)
    return BatchNormFn.apply(x, bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
                             bn.running_var if bn.track_running_stats else None, use_batch, mom, bn.eps, p_drop, seed)


# ------------------------------------------------------------------------------------------------
# fused decoders + losses
# ------------------------------------------------------------------------------------------------
def _dec_ws(B, F, K, dev):
    L = _lib.lib()
    ws_f = torch.empty(L.scp_dec_ws_floats(B, F, K), dtype=torch.float32, device=dev)
    ws_d = torch.empty(L.scp_dec_ws_doubles(B, F), dtype=torch.float64, device=dev)
    return ws_f, ws_d


_ones = {}


def backward_root(loss):
    """loss.backward() for a scalar loss that IS the root of the step (the trainers' case): the fused decoder losses
    take the upstream gradient to be exactly 1, so neither the ones tensor autograd would create nor the
    multiplication of their stored gradients by it is launched."""
    one = _ones.get(loss.device)
    if one is None:
        one = _ones[loss.device] = torch.ones((), dtype=torch.float32, device=loss.device)
    if loss.dtype != torch.float32 or loss.ndim != 0:
        loss.backward()
        return
    _cfg["root_loss"] = loss
    try:
        loss.backward(one)
    finally:
        _cfg["root_loss"] = None


def _scale_all(grads, g):
    """grads *= g (the scalar upstream gradient of a fused loss) for every tensor of the list, in one multi-tensor
    launch; the buffers are owned by the Function that produced them.  Skipped when the loss is the root handed to
    backward_root() (g is the constant 1)."""
    one = _ones.get(g.device)
    if _cfg.get("root_loss") is not None and one is not None and g.data_ptr() == one.data_ptr():
        return
    ts = [t for t in grads if t is not None]
    if ts:
        torch._foreach_mul_(ts, g.reshape(()).to(ts[0].dtype))


class BerDecoderLossFn(torch.autograd.Function):
    """loss = binarized_loss(y, sigmoid(h W^T + b) [* sigmoid(lrow)] [* sigmoid(fs)])  (model.py:164-190, loss.py:36-41)
    with dW, db, dh, dlrow, dfs produced by the same kernel pass."""

    @staticmethod
    def forward(ctx, h, W, b, fs, lrow, tiles):
        L = _lib.lib()
        h, Wc, bc = _f32c(h), _f32c(W), _f32c(b)
        _lib.require_cuda(h, Wc, bc)
        fsc = _f32c(fs)
        lr = None if lrow is None else _f32c(lrow).reshape(-1)
        Bn, K = h.shape
        F = Wc.shape[0]
        if tiles.shape != (Bn, F):
            raise ValueError(f"label tiles are {tiles.shape}, decoder output is {(Bn, F)}")
        lrow_n = 0 if lr is None else lr.numel()
        dev = h.device
        train = any(ctx.needs_input_grad[:5])
        ws_f, ws_d = _dec_ws(Bn, F, K, dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        dW = db = dh = dfs = dl = None
        if train:
            dW = torch.empty((F, K), dtype=torch.float32, device=dev)
            db = torch.empty(F, dtype=torch.float32, device=dev)
            dh = torch.empty((Bn, K), dtype=torch.float32, device=dev)
            dfs = torch.empty(F, dtype=torch.float32, device=dev) if fsc is not None else None
            dl = torch.empty(Bn, dtype=torch.float32, device=dev) if lrow_n == Bn else None
        tm = _cfg["decoder_timer"]
        if tm is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(L.scp_ber_dec_bce(int(train), Bn, F, K, h.data_ptr(), K, Wc.data_ptr(), K, bc.data_ptr(),
                                     _lib.ptr(fsc), _lib.ptr(lr), lrow_n, _lib.ptr(tiles.ptr), _lib.ptr(tiles.key),
                                     _lib.ptr(tiles.val), _lib.ptr(tiles.nnan), None, loss.data_ptr(), _lib.ptr(dW), K,
                                     _lib.ptr(db), _lib.ptr(dfs), _lib.ptr(dh), K, _lib.ptr(dl), ws_f.data_ptr(),
                                     ws_d.data_ptr(), _ns(), _lib.stream()), "scp_ber_dec_bce")
        if tm is not None:
            e1.record()
            tm.append((e0, e1, "ber", Bn, F, K, train))
        ctx.grads = (dh, dW, db, dfs, dl)
        ctx.lshape = None if lrow is None else lrow.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dh, dW, db, dfs, dl = ctx.grads
        ctx.grads = None
        _scale_all((dh, dW, db, dfs, dl), g)
        dl_ = None if dl is None else dl.reshape(ctx.lshape)
        return dh, dW, db, dfs, dl_, None


def decoder_depth_supported(K: int) -> bool:
    """The fused decoders (scp_ber_dec_bce, scp_nb_dec_ll) hold the whole contraction depth K = latent width + one-hot
    covariate columns in registers / LDS: K <= 64 with bf16 MFMA inputs, K <= 48 in the fp32-equivalent mode.  Deeper
    decoders (many covariate levels) take the materialised forward + the dense likelihood kernels instead."""
    return int(K) <= (64 if _ns() == 1 else 48)


def ber_decoder_loss(h, W, b, fs, lrow, tiles):
    return BerDecoderLossFn.apply(h, W, b, fs, lrow, tiles)


class NbDecoderLossFn(torch.autograd.Function):
    """loss = raw_loss('nb') of Output_Module_Raw (model.py:229-259, loss.py:46-48), fused with its backward.
    lib_L: per-cell library size (sample_factor) or exp(leaky_relu(latent_lib)); lat: latent_lib or None."""

    @staticmethod
    def forward(ctx, h, W, b, fs, theta_param, lib_L, lat, tiles):
        L = _lib.lib()
        h, Wc, bc = _f32c(h), _f32c(W), _f32c(b)
        _lib.require_cuda(h, Wc, bc)
        fsc, thp = _f32c(fs), _f32c(theta_param)
        Lr = _f32c(lib_L).reshape(-1)
        latc = None if lat is None else _f32c(lat).reshape(-1)
        Bn, K = h.shape
        F = Wc.shape[0]
        if tiles.shape != (Bn, F):
            raise ValueError(f"label tiles are {tiles.shape}, decoder output is {(Bn, F)}")
        if Lr.numel() != Bn:
            raise ValueError("library size must have one entry per cell")
        dev = h.device
        train = any(ctx.needs_input_grad[:7])
        ws_f, ws_d = _dec_ws(Bn, F, K, dev)
        rowbuf = torch.empty(2 * Bn, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        dW = db = dh = dfs = dth = dlat = None
        if train:
            dW = torch.empty((F, K), dtype=torch.float32, device=dev)
            db = torch.empty(F, dtype=torch.float32, device=dev)
            dh = torch.empty((Bn, K), dtype=torch.float32, device=dev)
            dfs = torch.empty(F, dtype=torch.float32, device=dev) if fsc is not None else None
            dth = torch.empty(F, dtype=torch.float32, device=dev) if thp is not None else None
            dlat = torch.empty(Bn, dtype=torch.float32, device=dev) if latc is not None else None
        tm = _cfg["decoder_timer"]
        if tm is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.check(L.scp_nb_dec_ll(int(train), Bn, F, K, h.data_ptr(), K, Wc.data_ptr(), K, bc.data_ptr(), _lib.ptr(fsc),
                                   _lib.ptr(thp), Lr.data_ptr(), _lib.ptr(latc), _lib.ptr(tiles.ptr), _lib.ptr(tiles.key),
                                   _lib.ptr(tiles.val), None, loss.data_ptr(), _lib.ptr(dW), K, _lib.ptr(db),
                                   _lib.ptr(dfs), _lib.ptr(dth), _lib.ptr(dh), K, _lib.ptr(dlat), rowbuf.data_ptr(),
                                   ws_f.data_ptr(), ws_d.data_ptr(), _ns(), _lib.stream()), "scp_nb_dec_ll")
        if tm is not None:
            e1.record()
            tm.append((e0, e1, "nb", Bn, F, K, train))
        ctx.grads = (dh, dW, db, dfs, dth, dlat)
        ctx.latshape = None if lat is None else lat.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dh, dW, db, dfs, dth, dlat = ctx.grads
        ctx.grads = None
        _scale_all((dh, dW, db, dfs, dth, dlat), g)
        dlat_ = None if dlat is None else dlat.reshape(ctx.latshape)
        return dh, dW, db, dfs, dth, None, dlat_, None


def nb_decoder_loss(h, W, b, fs, theta_param, lib_L, lat, tiles):
    return NbDecoderLossFn.apply(h, W, b, fs, theta_param, lib_L, lat, tiles)


# ------------------------------------------------------------------------------------------------
# (ZI)NB loss on materialised pre-activations (ZINB predictor, VAE heads)
# ------------------------------------------------------------------------------------------------
class RawRowsLossFn(torch.autograd.Function):
    """(ZI)NB loss of a decoder head from its materialised pre-activations, one row kernel forward and one backward
    (loss.py:46-51).  tz None: the predictor's head, raw_loss(y, L*softmax(relu(z)), exp(relu(z)), ., pi)
    (model.py:229-259, dispersion 'feature-cell', no feature factor).  tz given: the VAE decoder's heads,
    raw_loss(y, L*softmax(z), softplus(tz) + 1e-8, ., pi) (pairedVAE.py:147-187).  pi None: plain NB."""

    @staticmethod
    def forward(ctx, z, pi, L, y, tz):
        lib = _lib.lib()
        z, y = _f32c(z), _f32c(y)
        pi = None if pi is None else _f32c(pi)
        tz = None if tz is None else _f32c(tz)
        Lr = _f32c(L).reshape(-1)
        _lib.require_cuda(z, y, Lr, pi, tz)
        B, G = z.shape
        if y.shape != z.shape or Lr.numel() != B or (pi is not None and pi.shape != z.shape) or \
                (tz is not None and tz.shape != z.shape):
            raise ValueError("raw_rows_loss: shape mismatch")
        fast = 1 if get_precision() == "bf16" else 0
        lse, T = _empty((B,), z), _empty((B,), z)
        part = torch.empty(B, dtype=torch.float64, device=z.device)
        loss = _empty((1,), z)
        _lib.check(lib.scp_raw_rows_fwd(z.data_ptr(), G, _lib.ptr(tz), G, _lib.ptr(pi), G, y.data_ptr(), G, Lr.data_ptr(),
                                        B, G, fast, lse.data_ptr(), T.data_ptr(), part.data_ptr(), loss.data_ptr(),
                                        _lib.stream()), "scp_raw_rows_fwd")
        ctx.save_for_backward(z, pi, Lr, y, lse, T, tz)
        ctx.l_shape, ctx.fast = L.shape, fast
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        lib = _lib.lib()
        z, pi, Lr, y, lse, T, tz = ctx.saved_tensors
        B, G = z.shape
        g = _f32c(g).reshape(1)
        dz = torch.empty_like(z)
        dpi = None if pi is None else torch.empty_like(pi)
        dtz = None if tz is None else torch.empty_like(tz)
        dL = _empty((B,), z) if ctx.needs_input_grad[2] else None
        _lib.check(lib.scp_raw_rows_bwd(z.data_ptr(), G, _lib.ptr(tz), G, _lib.ptr(pi), G, y.data_ptr(), G, Lr.data_ptr(),
                                        lse.data_ptr(), T.data_ptr(), g.data_ptr(), B, G, ctx.fast, dz.data_ptr(), G,
                                        _lib.ptr(dtz), G, _lib.ptr(dpi), G, _lib.ptr(dL), _lib.stream()),
                   "scp_raw_rows_bwd")
        return dz, dpi, (None if dL is None else dL.reshape(ctx.l_shape)), None, dtz


class RawRowsFeatLossFn(torch.autograd.Function):
    """The predictor's (ZI)NB head with per-feature parameters (model.py:239-252): fs = feature_scalar
    (scale = softmax(relu(z) * sigmoid(fs))), thp = output_inverse_dispersion (theta = exp(thp), dispersion 'feature');
    either may be None.  Same row kernels as RawRowsLossFn plus a column-parallel kernel for dfs / dthp."""

    @staticmethod
    def forward(ctx, z, pi, L, y, fs, thp):
        lib = _lib.lib()
        z, y = _f32c(z), _f32c(y)
        pi = None if pi is None else _f32c(pi)
        fs = None if fs is None else _f32c(fs)
        thp = None if thp is None else _f32c(thp)
        Lr = _f32c(L).reshape(-1)
        _lib.require_cuda(z, y, Lr, pi, fs, thp)
        B, G = z.shape
        if y.shape != z.shape or Lr.numel() != B or (pi is not None and pi.shape != z.shape) or \
                (fs is not None and fs.numel() != G) or (thp is not None and thp.numel() != G):
            raise ValueError("raw_rows_loss: shape mismatch")
        fast = 1 if get_precision() == "bf16" else 0
        lse, T = _empty((B,), z), _empty((B,), z)
        part = torch.empty(B, dtype=torch.float64, device=z.device)
        loss = _empty((1,), z)
        _lib.check(lib.scp_raw_rows_feat_fwd(z.data_ptr(), G, _lib.ptr(pi), G, y.data_ptr(), G, Lr.data_ptr(), _lib.ptr(fs),
                                             _lib.ptr(thp), B, G, fast, lse.data_ptr(), T.data_ptr(), part.data_ptr(),
                                             loss.data_ptr(), _lib.stream()), "scp_raw_rows_feat_fwd")
        ctx.save_for_backward(z, pi, Lr, y, lse, T, fs, thp)
        ctx.l_shape, ctx.fast = L.shape, fast
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g):
        lib = _lib.lib()
        z, pi, Lr, y, lse, T, fs, thp = ctx.saved_tensors
        B, G = z.shape
        g = _f32c(g).reshape(1)
        dz = torch.empty_like(z)
        dpi = None if pi is None else torch.empty_like(pi)
        dL = _empty((B,), z) if ctx.needs_input_grad[2] else None
        dfs = torch.empty_like(fs) if (fs is not None and ctx.needs_input_grad[4]) else None
        dthp = torch.empty_like(thp) if (thp is not None and ctx.needs_input_grad[5]) else None
        ws = None
        if dfs is not None or dthp is not None:
            ws = _empty((lib.scp_raw_rows_feat_ws_floats(B, G),), z)
        _lib.check(lib.scp_raw_rows_feat_bwd(z.data_ptr(), G, _lib.ptr(pi), G, y.data_ptr(), G, Lr.data_ptr(), _lib.ptr(fs),
                                             _lib.ptr(thp), lse.data_ptr(), T.data_ptr(), g.data_ptr(), B, G, ctx.fast,
                                             dz.data_ptr(), G, _lib.ptr(dpi), G, _lib.ptr(dL), _lib.ptr(dfs), _lib.ptr(dthp),
                                             _lib.ptr(ws), _lib.stream()), "scp_raw_rows_feat_bwd")
        return dz, dpi, (None if dL is None else dL.reshape(ctx.l_shape)), None, dfs, dthp


def raw_rows_loss(z, pi, L, y, tz=None, fs=None, thp=None):
    if fs is not None or thp is not None:
        if tz is not None:
            raise ValueError("raw_rows_loss: per-feature parameters belong to the predictor's head (tz must be None)")
        return RawRowsFeatLossFn.apply(z, pi, L, y, fs, thp)
    return RawRowsLossFn.apply(z, pi, L, y, tz)


# ------------------------------------------------------------------------------------------------
# masked MSE, reparameterisation + KL
# ------------------------------------------------------------------------------------------------
class MseMaskedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mu):
        L = _lib.lib()
        x, mu = _f32c(x), _f32c(mu)
        _lib.require_cuda(x, mu)
        if x.shape != mu.shape:
            raise ValueError("scaled_loss: shape mismatch")
        rows = x.shape[0] if x.ndim > 1 else 1
        cols = x.numel() // rows
        lc = torch.empty(2, dtype=torch.float32, device=x.device)
        ws = torch.empty(520, dtype=torch.float64, device=x.device)
        _lib.check(L.scp_mse_masked_fwd(x.data_ptr(), cols, mu.data_ptr(), cols, rows, cols, lc.data_ptr(),
                                        ws.data_ptr(), _lib.stream()), "scp_mse_masked_fwd")
        ctx.save_for_backward(x, mu, lc)
        ctx.rc = (rows, cols)
        return lc[0].clone()

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        x, mu, lc = ctx.saved_tensors
        rows, cols = ctx.rc
        g = _f32c(g).reshape(1)
        dmu = torch.empty_like(mu)
        _lib.check(L.scp_mse_masked_bwd(x.data_ptr(), cols, mu.data_ptr(), cols, rows, cols, lc.data_ptr(),
                                        g.data_ptr(), dmu.data_ptr(), cols, _lib.stream()), "scp_mse_masked_bwd")
        return None, dmu


def mse_masked(x, mu):
    return MseMaskedFn.apply(x, mu)


class ReparamKlFn(torch.autograd.Function):
    """(mu, s_raw, eps) -> sigma = softplus(s_raw)+1e-7, z = mu + eps*sigma, kl = mean KL(N(mu,sigma)||N(0,1))."""

    @staticmethod
    def forward(ctx, mu, sraw, eps):
        L = _lib.lib()
        mu, sraw, eps = _f32c(mu), _f32c(sraw), _f32c(eps)
        _lib.require_cuda(mu, sraw, eps)
        n = mu.numel()
        sigma, z = torch.empty_like(mu), torch.empty_like(mu)
        kl = torch.empty(1, dtype=torch.float32, device=mu.device)
        ws = torch.empty(520, dtype=torch.float64, device=mu.device)
        _lib.check(L.scp_reparam_kl_fwd(mu.data_ptr(), sraw.data_ptr(), eps.data_ptr(), n, sigma.data_ptr(), z.data_ptr(),
                                        kl.data_ptr(), ws.data_ptr(), _lib.stream()), "scp_reparam_kl_fwd")
        ctx.save_for_backward(mu, sraw, eps, sigma)
        return sigma, z, kl.reshape(())

    @staticmethod
    def backward(ctx, _dsigma, dz, dkl):
        L = _lib.lib()
        mu, sraw, eps, sigma = ctx.saved_tensors
        n = mu.numel()
        dzc = None if dz is None else _f32c(dz)
        gk = None if dkl is None else _f32c(dkl).reshape(1)
        dmu, dsraw = torch.empty_like(mu), torch.empty_like(mu)
        _lib.check(L.scp_reparam_kl_bwd(mu.data_ptr(), sraw.data_ptr(), eps.data_ptr(), sigma.data_ptr(), _lib.ptr(dzc),
                                        None, _lib.ptr(gk), n, dmu.data_ptr(), dsraw.data_ptr(), _lib.stream()),
                   "scp_reparam_kl_bwd")
        if _dsigma is not None:
            # a caller differentiating through the returned sigma itself (e.g. its own KL from mu, sigma)
            dsraw = dsraw + _dsigma * torch.where(sraw > 20, torch.ones_like(sraw), torch.sigmoid(sraw))
        return dmu, dsraw, None


def reparam_kl(mu, sraw, eps):
    return ReparamKlFn.apply(mu, sraw, eps)


# ------------------------------------------------------------------------------------------------
# inference heads
# ------------------------------------------------------------------------------------------------
def ber_prob(z, lrow=None, fs=None, out_dtype=torch.float32):
    """sigmoid(z) [* sigmoid(lrow)] [* sigmoid(fs)] (model.py:173-189), fp32 or bf16 output; inference only."""
    L = _lib.lib()
    z = _f32c(z)
    _lib.require_cuda(z, lrow, fs)
    rows, cols = z.shape
    out = torch.empty((rows, cols), dtype=out_dtype, device=z.device)
    lr = None if lrow is None else _f32c(lrow).reshape(-1)
    if lr is not None and lr.numel() not in (1, rows):
        raise ValueError("ber_prob: the sample factor needs 1 or one entry per row")
    _lib.check(L.scp_ber_prob(z.data_ptr(), cols, rows, cols, _lib.ptr(lr), 0 if lr is None else lr.numel(),
                              _lib.ptr(_f32c(fs)), out.data_ptr(), cols, int(out_dtype == torch.bfloat16), _lib.stream()),
               "scp_ber_prob")
    return out


def relu_softmax_rows(z, fs=None, apply_relu=True, want_theta=False, out_dtype=torch.float32):
    """scale = softmax(relu(z) [* sigmoid(fs)]) per row and, on request, theta = exp(relu(z)) (model.py:238-252);
    inference only."""
    L = _lib.lib()
    z = _f32c(z)
    _lib.require_cuda(z, fs)
    rows, cols = z.shape
    scale = torch.empty((rows, cols), dtype=out_dtype, device=z.device)
    theta = torch.empty_like(z) if want_theta else None
    _lib.check(L.scp_relu_softmax_rows(z.data_ptr(), cols, rows, cols, _lib.ptr(_f32c(fs)), int(apply_relu),
                                       scale.data_ptr(), cols, _lib.ptr(theta), cols, int(out_dtype == torch.bfloat16),
                                       _lib.stream()), "scp_relu_softmax_rows")
    return scale, theta


# ------------------------------------------------------------------------------------------------
# optimiser
# ------------------------------------------------------------------------------------------------
class FusedFirstLayer:
    """Registration of a first-layer weight whose AdamW step runs inside its weight-gradient kernel
    (FusedAdamW.fuse_first_layer).  Holds weak references only; the registry below is keyed by id(parameter), so
    nothing is attached to the Parameter object itself (it stays picklable / deep-copyable as before)."""

    def __init__(self, opt, param, group):
        self._opt, self._param, self.group = weakref.ref(opt), weakref.ref(param), group
        self.pending = False

    @property
    def opt(self):
        return self._opt()

    @property
    def param(self):
        return self._param()

    def begin_step(self, W):
        """Called by the layer's backward: optimiser state and hyper-parameters of the step about to be applied."""
        p, opt = self.param, self.opt
        if p is None or opt is None or W.data_ptr() != p.data_ptr():
            raise RuntimeError("fused first layer: backward sees a copy of the weights, not the registered parameter")
        if self.pending:
            raise RuntimeError("fused first layer: two backward() passes without optimizer.step()")
        st = opt.state[p]
        if len(st) == 0:
            _new_adam_state(opt, p)
        if row_ld(st["exp_avg"]) != row_ld(p) or row_ld(st["exp_avg_sq"]) != row_ld(p):
            _relayout_state(st, p)
        t = st["step"] + 1
        b1, b2 = self.group["betas"]
        sh = shadow_of(p)
        self.pending = True
        dyn = opt.step_state
        if dyn is not None:               # captured step: bias corrections come from the device words
            opt._dyn_note(p, (b1, b2))
            return st["exp_avg"], st["exp_avg_sq"], sh, (self.group["lr"], self.group["weight_decay"], b1, b2,
                                                         self.group["eps"], 1.0, 1.0, 1.0, dyn.bc_ptr)
        return st["exp_avg"], st["exp_avg_sq"], sh, (self.group["lr"], self.group["weight_decay"], b1, b2,
                                                     self.group["eps"], 1 - b1 ** t, 1 - b2 ** t, 1.0, None)


_fused = {}            # id(parameter) -> FusedFirstLayer


class GradSink:
    """bf16 exchange buffer of a first-layer weight under data parallelism (precision 'bf16'): the weight-gradient
    kernel writes dW there as bf16, the all-reduce sums it in place and FusedAdamW reads it as the gradient, so the
    fp32 gradient, its cast and the copy back never exist.  `view` is a contiguous bf16 [H, F] slice of the buffer
    scpair_amd.distributed.GradSync reduces; `written` is set by backward and consumed by FusedAdamW.step."""

    def __init__(self, param, view):
        self._param, self.view, self.written = weakref.ref(param), view, False

    @property
    def param(self):
        return self._param()


_sinks = {}            # id(parameter) -> GradSink


def set_grad_sink(W, view):
    if view is None:
        _sinks.pop(id(W), None)
        return None
    if view.dtype != torch.bfloat16 or tuple(view.shape) != tuple(W.shape) or not view.is_contiguous():
        raise ValueError("gradient sink must be a contiguous bf16 tensor of the weight's shape")
    for k in [k for k, h in _sinks.items() if h.param is None]:
        del _sinks[k]
    _sinks[id(W)] = GradSink(W, view)
    return _sinks[id(W)]


def grad_sink_of(W):
    h = _sinks.get(id(W))
    if h is None:
        return None
    if h.param is not W:
        del _sinks[id(W)]
        return None
    return h


def fused_handle(W):
    h = _fused.get(id(W))
    if h is None:
        return None
    if h.param is not W or h.opt is None:
        del _fused[id(W)]
        return None
    return h


import struct as _struct

_ADAM_FMT = _struct.Struct("@PPPPqfffffPqqqPq")     # ScpAdamTensor, packed without building ctypes objects per tensor
assert _ADAM_FMT.size == C.sizeof(_lib.AdamTensor)


_NOCACHE = bool(_os.environ.get("SCPAIR_ADAM_NOCACHE"))      # A/B measurements of step()'s per-parameter cache


class FusedAdamW(torch.optim.Optimizer):
    """Drop-in for torch.optim.AdamW as the reference configures it (model.py:698-701): same param_groups
    / state layout ('step', 'exp_avg', 'exp_avg_sq'), one fused multi-tensor HIP launch per step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._grads_released = False    # the last step() set every gradient it consumed to None
        self.last_step_numel = (0, 0)
        self._checked = {}              # id(p) -> data_ptr whose device / contiguity checks have passed
        self._pc = {}                   # id(p) -> (p, data_ptr, state dict, exp_avg ptr, exp_avg_sq ptr, numel, last dim): the
                                        # per-parameter constants of step() for plain contiguous parameters without a shadow
        self._bufs = {}                 # (beta1, beta2, eps) -> bytearray of packed ScpAdamTensor descriptors
        self.step_state = None          # StepState while a step of this optimiser is being captured / replayed
        self._dyn_params = {}           # id(p) -> p: parameters the captured step updates
        self._dyn_betas = None

    def __setstate__(self, state):
        super().__setstate__(state)     # load_state_dict() lands here with fresh state dicts: the cached ones are stale
        self._pc = {}

    # -- a step captured in a HIP graph: 'step' counters advance on the host, bias corrections live on the device ---
    def _dyn_note(self, p, betas):
        if self._dyn_betas is None:
            self._dyn_betas = tuple(betas)
        elif self._dyn_betas != tuple(betas):
            raise RuntimeError("a captured step needs the same betas in every parameter group")
        self._dyn_params[id(p)] = p

    def begin_capture(self, state):
        self.step_state, self._dyn_params, self._dyn_betas = state, {}, None

    def end_capture(self):
        """Capture done: eager step() calls compute their bias corrections on the host again; the captured step keeps
        reading the device words, and dyn_advance() keeps the host counters of its parameters in step."""
        self.step_state = None
        steps = {self.state[p]["step"] for p in self._dyn_params.values()}
        if len(steps) > 1:
            raise RuntimeError("a captured step needs every parameter it updates at the same step count")

    def release_capture(self):
        self.step_state, self._dyn_params, self._dyn_betas = None, {}, None

    def dyn_advance(self):
        """Before a replay: advance the host-side step counters of the parameters the captured step updates and
        return the bias corrections (1 - beta1^t, 1 - beta2^t) of that step (computed as step() does)."""
        t = None
        for p in self._dyn_params.values():
            st = self.state[p]
            st["step"] += 1
            t = st["step"]
        if t is None:
            return 1.0, 1.0
        b1, b2 = self._dyn_betas
        return 1 - b1 ** t, 1 - b2 ** t

    # -- AdamW step of the sparse first-layer weights inside their weight-gradient kernel ----------------------
    def fuse_first_layer(self, *params):
        """Opt in for `params` (2-D first-layer weights of a HipFCNN fed with BatchTiles, precision 'bf16', no
        covariate block): their gradient is consumed by the AdamW update in the epilogue of the kernel that computes
        it, during backward(), and step() only advances their step counter.  Same arithmetic as step(); meant for
        single-process training with exactly one backward() per step() (no gradient exchange, clipping or
        accumulation in between).  Layers that do not meet the conditions keep the ordinary path."""
        group_of = {id(q): g for g in self.param_groups for q in g["params"]}
        for p in params:
            if id(p) not in group_of:
                raise ValueError("fuse_first_layer: parameter is not managed by this optimiser")
            if p.ndim != 2:
                raise ValueError("fuse_first_layer: expects 2-D weights")
            for k in [k for k, h in _fused.items() if h.param is None or h.opt is None]:
                del _fused[k]
            _fused[id(p)] = FusedFirstLayer(self, p, group_of[id(p)])

    def fuse_weight_update(self, *params):
        """The same opt-in for any 2-D weight whose weight-gradient kernel can apply the AdamW step in its epilogue:
        the sparse first layers (scp_spmm_dw_adamw_bf16) and the wide plain nn.Linear layers of precision 'bf16' (ZINB
        G x G zero-inflation layer, VAE 900 <-> G layers: scp_gemm_bf16_adamw).  Same conditions as
        fuse_first_layer; a layer that does not meet them in a given forward keeps the ordinary path."""
        self.fuse_first_layer(*params)

    def unfuse(self):
        for g in self.param_groups:
            for p in g["params"]:
                h = fused_handle(p)
                if h is not None and h.opt is self:
                    if h.pending:
                        raise RuntimeError("unfuse: a fused update is pending, call step() first")
                    del _fused[id(p)]

    def zero_grad(self, set_to_none: bool = True):
        """torch's zero_grad(set_to_none=True) as a plain loop (the trainers call it right before every backward(),
        40 parameters per predictor); after step(release_grads=True) it finds nothing to drop."""
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    p.grad = None

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0, release_grads: bool = False):
        """One AdamW step on every parameter that has a gradient (or a written bf16 gradient sink).
        release_grads: set the consumed `.grad`s to None afterwards, which makes the next zero_grad() free."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        fmt = _ADAM_FMT
        by_hyper = {}
        keep = []
        dyn = self.step_state
        checked = self._checked
        pc = self._pc
        released = release_grads
        n32 = n16 = 0
        for group in self.param_groups:
            b1, b2 = group["betas"]
            lr, wd = group["lr"], group["weight_decay"]
            key = (b1, b2, group["eps"])
            for p in group["params"]:
                fz = _fused.get(id(p)) if _fused else None
                if fz is not None and fz.pending and fz.opt is self and fz.param is p:
                    if grad_scale != 1.0 and not getattr(p, "_scp_feature_sharded", False):
                        # (a feature-sharded block's gradient already carries the ranks' shares: nothing left to scale)
                        raise RuntimeError("FusedAdamW: grad_scale is not available for weights updated during backward()")
                    if p.grad is not None:
                        raise RuntimeError("FusedAdamW: a weight updated during backward() also received a gradient")
                    if dyn is None:
                        self.state[p]["step"] += 1
                    fz.pending = False
                    continue
                g16 = 0
                g = p.grad
                ce = pc.get(id(p)) if (g is not None and dyn is None and not _NOCACHE) else None
                if ce is not None and ce[0] is p and g.is_contiguous():
                    # fast path: a plain contiguous parameter whose storage, moments and (absent) shadow are as cached
                    st = ce[2]
                    ea = st.get("exp_avg")
                    if ce[1] == p.data_ptr() and ea is not None and ea.data_ptr() == ce[3] and id(p) not in _shadows:
                        t = st["step"] = st["step"] + 1
                        ent = by_hyper.get(key)
                        if ent is None:
                            ent = by_hyper[key] = [self._desc_buf(key), 0]
                        buf, n = ent
                        if (n + 1) * fmt.size > len(buf):
                            buf.extend(bytes(len(buf)))
                        fmt.pack_into(buf, n * fmt.size, ce[1], g.data_ptr(), ce[3], ce[4], ce[5], lr, wd, 1 - b1 ** t,
                                      1 - b2 ** t, grad_scale, 0, ce[6], 0, 0, 0, 0)
                        ent[1] = n + 1
                        keep.append(g)
                        n32 += ce[5]
                        if released:
                            p.grad = None
                        continue
                if g is None:
                    sk = _sinks.get(id(p)) if _sinks else None
                    if sk is None or not sk.written or sk.param is not p:
                        continue
                    g, g16, sk.written = sk.view, 1, False      # bf16 gradient in the exchange buffer
                elif not g.is_contiguous():
                    g = g.contiguous()
                pptr = p.data_ptr()
                if checked.get(id(p)) != pptr:                  # device / layout checks once per (parameter, storage)
                    _lib.require_cuda(p)
                    if not p.is_contiguous() and (p.ndim != 2 or row_ld(p) is None):
                        raise RuntimeError("FusedAdamW needs contiguous parameters (or row-padded 2-D weights)")
                    checked[id(p)] = pptr
                st = self.state[p]
                if len(st) == 0:
                    _new_adam_state(self, p)
                if dyn is None:
                    t = st["step"] = st["step"] + 1
                    bc1, bc2, bc_dev = 1 - b1 ** t, 1 - b2 ** t, 0
                else:                         # captured step: counters advance in dyn_advance(), once per replay
                    self._dyn_note(p, (b1, b2))
                    bc1, bc2, bc_dev = 1.0, 1.0, dyn.bc_ptr
                sh = shadow_of(p, create=False) if p.ndim == 2 else None
                ent = by_hyper.get(key)
                if ent is None:
                    ent = by_hyper[key] = [self._desc_buf(key), 0]
                buf, n = ent
                if (n + 1) * fmt.size > len(buf):
                    buf.extend(bytes(len(buf)))
                ld_p = 0
                if p.ndim == 2 and p.stride(0) != p.shape[1]:   # row-padded weight: moments share its layout
                    ld_p = p.stride(0)
                    if st["exp_avg"].stride(0) != ld_p or st["exp_avg_sq"].stride(0) != ld_p:
                        _relayout_state(st, p)
                fmt.pack_into(buf, n * fmt.size, pptr, g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                              p.numel(), lr, wd, bc1, bc2, grad_scale, 0 if sh is None else sh.data_ptr(), p.shape[-1],
                              0 if sh is None else sh.shape[1], g16, bc_dev, ld_p)
                ent[1] = n + 1
                keep.append(g)
                if g16:
                    n16 += p.numel()
                else:
                    n32 += p.numel()
                if dyn is None and not g16 and sh is None and ld_p == 0 and p.is_contiguous():
                    pc[id(p)] = (p, pptr, st, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), p.shape[-1])
                if released and not g16:
                    p.grad = None
        stream = _lib.stream()
        for (b1, b2, eps), (buf, n) in by_hyper.items():
            arr = (C.c_char * len(buf)).from_buffer(buf)
            _lib.check(L.scp_adamw_multi(C.cast(arr, C.POINTER(_lib.AdamTensor)), n, b1, b2, eps, stream), "scp_adamw_multi")
            del arr
        self._grads_released = released
        self.last_step_numel = (n32, n16)       # parameters stepped from fp32 / bf16 gradients by this call
        return loss

    def _desc_buf(self, key):
        b = self._bufs.get(key)
        if b is None:
            b = self._bufs[key] = bytearray(_ADAM_FMT.size * 64)
        return b


# ------------------------------------------------------------------------------------------------
# stand-alone likelihoods on dense tensors (the reference's own call form)
# ------------------------------------------------------------------------------------------------
class BceMaskedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        L = _lib.lib()
        x, p = _f32c(x), _f32c(p)
        _lib.require_cuda(x, p)
        if x.shape != p.shape:
            raise ValueError("binarized_loss: shape mismatch")
        n = x.numel()
        lc = torch.empty(2, dtype=torch.float32, device=x.device)
        ws = torch.empty(520, dtype=torch.float64, device=x.device)
        _lib.check(L.scp_bce_masked_fwd(x.data_ptr(), p.data_ptr(), n, lc.data_ptr(), ws.data_ptr(), _lib.stream()),
                   "scp_bce_masked_fwd")
        ctx.save_for_backward(x, p, lc)
        return lc[0].clone()

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        x, p, lc = ctx.saved_tensors
        g = _f32c(g).reshape(1)
        dp = torch.empty_like(p)
        _lib.check(L.scp_bce_masked_bwd(x.data_ptr(), p.data_ptr(), x.numel(), lc.data_ptr(), g.data_ptr(),
                                        dp.data_ptr(), _lib.stream()), "scp_bce_masked_bwd")
        return None, dp


def bce_masked(x, p):
    return BceMaskedFn.apply(x, p)


class NbLogLikFn(torch.autograd.Function):
    """Element-wise log NB / ZINB likelihood (scVI_distribution.py:29-105); pi=None selects NB."""

    @staticmethod
    def forward(ctx, x, mu, theta, pi):
        L = _lib.lib()
        x, mu = _f32c(x), _f32c(mu)
        _lib.require_cuda(x, mu, theta, pi)
        shape = torch.broadcast_shapes(x.shape, mu.shape, theta.shape, *(() if pi is None else (pi.shape,)))
        cols = shape[-1]
        theta_vec = theta.ndim == 1 and theta.numel() == cols and len(shape) > 1
        xe = _f32c(x.expand(shape)); me = _f32c(mu.expand(shape))
        te = _f32c(theta) if theta_vec else _f32c(theta.expand(shape))
        pe = None if pi is None else _f32c(pi.expand(shape))
        n = xe.numel()
        out = torch.empty(shape, dtype=torch.float32, device=xe.device)
        _lib.check(L.scp_nbll_fwd(xe.data_ptr(), me.data_ptr(), te.data_ptr(), int(theta_vec), cols, _lib.ptr(pe), n,
                                  out.data_ptr(), _lib.stream()), "scp_nbll_fwd")
        ctx.save_for_backward(xe, me, te, pe)
        ctx.meta = (theta_vec, cols, mu.shape, theta.shape, None if pi is None else pi.shape)
        if theta_vec:
            out = out.reshape((1,) * 0 + tuple(shape))
        return out

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        xe, me, te, pe = ctx.saved_tensors
        theta_vec, cols, mu_shape, th_shape, pi_shape = ctx.meta
        g = _f32c(g.expand(xe.shape))
        n = xe.numel()
        dmu, dth = torch.empty_like(xe), torch.empty_like(xe)
        dpi = torch.empty_like(xe) if pe is not None else None
        _lib.check(L.scp_nbll_bwd(xe.data_ptr(), me.data_ptr(), te.data_ptr(), int(theta_vec), cols, _lib.ptr(pe),
                                  g.data_ptr(), n, dmu.data_ptr(), dth.data_ptr(), _lib.ptr(dpi), _lib.stream()),
                   "scp_nbll_bwd")

        def unb(t, shape):
            if t is None or shape is None:
                return None
            return t.sum_to_size(shape) if tuple(t.shape) != tuple(shape) else t
        return None, unb(dmu, mu_shape), unb(dth, th_shape), unb(dpi, pi_shape)


def nb_loglik(x, mu, theta, pi=None):
    return NbLogLikFn.apply(x, mu, theta, pi)
