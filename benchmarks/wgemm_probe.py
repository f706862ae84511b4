"""One shape of the wide GEMM, timed: python benchmarks/wgemm_probe.py a_col b_col M N K  (env SCPAIR_WGEMM_NSPLIT / _DBG)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scpair_amd import ops, _lib
ac, bc, M, N, K = [int(v) for v in sys.argv[1:6]]
dev = torch.device("cuda:0")
def bf(rows, cols):
    ld = (cols + 7) // 8 * 8
    return torch.randn((rows, ld), device=dev).to(torch.bfloat16), ld
A, lda = bf(K, M) if ac else bf(M, K)
B, ldb = bf(K, N) if bc else bf(N, K)
C = torch.empty((M, N), device=dev)
f = lambda: ops.gemm_bf16(ac, bc, M, N, K, A, lda, B, ldb, C, N)
for _ in range(3): f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): f()
b.record(); torch.cuda.synchronize()
t = a.elapsed_time(b) / 10
print(f"a_col {ac} b_col {bc} {M}x{N}x{K} nsplit {_lib.lib().scp_gemm_bf16_nsplit(M, N, K)} dbg {os.environ.get('SCPAIR_WGEMM_DBG','0')}: {t:.3f} ms {2.0*M*N*K/t/1e9:.0f} TFLOP/s", flush=True)
