"""Times the hand-written wide bf16 GEMM (scp_gemm_bf16) beside rocBLAS (benchmarks/rocblas_cmp: a benchmark-only library, not part of libscpair_hip.so) on the shapes of the wide
nn.Linear layers: ZINB G x G (model.py:228) and VAE 900 <-> G (pairedVAE.py:111-119), batch 1024.
usage: python benchmarks/wgemm_bench.py [G]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctypes as C
from scpair_amd import ops, _lib

_cmp = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rocblas_cmp", "librocblas_cmp.so"))
_cmp.cmp_gemm_rocblas_bf16.restype = C.c_int
_cmp.cmp_gemm_rocblas_bf16.argtypes = [C.c_int] * 5 + [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p]


def gemm_lib(ta, tb, m, n, k, A, lda, B, ldb, Cm, ldc):
    rc = _cmp.cmp_gemm_rocblas_bf16(ta, tb, m, n, k, A.data_ptr(), lda, B.data_ptr(), ldb, Cm.data_ptr(), ldc, _lib.stream())
    if rc:
        raise RuntimeError(f"rocBLAS comparison GEMM failed ({rc})")

G = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = torch.device("cuda:0")


def bf(rows, cols):
    ld = (cols + 7) // 8 * 8
    t = torch.zeros((rows, ld), dtype=torch.bfloat16, device=dev)
    t[:, :cols] = torch.randn((rows, cols), device=dev).to(torch.bfloat16)
    return t, ld


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def layer(M, N, K):
    """nn.Linear(K -> N) on a batch of M: forward, input gradient, weight gradient."""
    x, ldx = bf(M, K)
    W, ldw = bf(N, K)
    dy, ldy = bf(M, N)
    y = torch.empty((M, N), device=dev)
    dx = torch.empty((M, K), device=dev)
    dW = torch.empty((N, K), device=dev)
    bias = torch.randn((N,), device=dev)
    rows = [
        ("fwd y = x W^T", M, N, K, lambda: ops.gemm_bf16(0, 0, M, N, K, x, ldx, W, ldw, y, N, bias=bias),
         lambda: gemm_lib(1, 0, N, M, K, W, ldw, x, ldx, y, N)),
        ("dx = dy W", M, K, N, lambda: ops.gemm_bf16(0, 1, M, K, N, dy, ldy, W, ldw, dx, K),
         lambda: gemm_lib(0, 0, K, M, N, W, ldw, dy, ldy, dx, K)),
        ("dW = dy^T x", N, K, M, lambda: ops.gemm_bf16(1, 1, N, K, M, dy, ldy, x, ldx, dW, K),
         lambda: gemm_lib(0, 1, K, N, M, x, ldx, dy, ldy, dW, K)),
    ]
    for name, m, n, k, ours, lib in rows:
        t1, t2 = timeit(ours), timeit(lib)
        fl = 2.0 * m * n * k
        ns = _lib.lib().scp_gemm_bf16_nsplit(m, n, k)
        print(f"{name:14s} {m:6d} x {n:6d} x {k:6d}  K-ranges {ns}  hand-written {t1:7.3f} ms {fl / t1 / 1e9:7.1f} TFLOP/s "
              f"({fl / t1 / 1e9 / 2500 * 100:4.1f} % of 2.5 PF)   rocBLAS {t2:7.3f} ms {fl / t2 / 1e9:7.1f} TFLOP/s", flush=True)


print(f"ZINB zero-inflation layer {G} -> {G}, batch 1024")
layer(1024, G, G)
print(f"VAE encoder layer {G} -> 900, batch 1024")
layer(1024, 900, G)
print(f"VAE decoder layer 900 -> {G}, batch 1024")
layer(1024, G, 900)
