// rocBLAS bf16 GEMM as the COMPARISON for the hand-written wide GEMM (scpair_amd/csrc/wgemm.hip, scp_gemm_bf16) on the
// shapes of the wide nn.Linear layers (ZINB G x G model.py:228, VAE 900 <-> G pairedVAE.py:111-119).  Benchmark-only:
// built into benchmarks/rocblas_cmp/librocblas_cmp.so by scpair_amd/csrc/build.sh and loaded by benchmarks/wgemm_bench.py;
// the product library libscpair_hip.so neither contains nor links it.
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>

static rocblas_handle g_handle[16] = {};

// Column-major (BLAS) convention:  C[m x n] = op(A)[m x k] * op(B)[k x n],  ta / tb: 0 = N, 1 = T.
// A row-major matrix R[r][c] with leading dimension ld is the column-major matrix [c x r] with the same ld, so
//   y  = x W^T   (row-major [M,N]) : ta=1 tb=0  m=N n=M k=K  A=W(ld K)   B=x(ld K)   C=y(ld N)
//   dx = dy W    (row-major [M,K]) : ta=0 tb=0  m=K n=M k=N  A=W(ld K)   B=dy(ld N)  C=dx(ld K)
//   dW = dy^T x  (row-major [N,K]) : ta=0 tb=1  m=K n=N k=M  A=x(ld K)   B=dy(ld N)  C=dW(ld K)
// Returns 0, or 1000 + the failing rocBLAS status / -1 for bad arguments.
extern "C" int cmp_gemm_rocblas_bf16(int ta, int tb, int m, int n, int k, const void* A, long lda, const void* B, long ldb,
                                     float* C, long ldc, void* stream_) {
  if (m <= 0 || n <= 0 || k <= 0 || !A || !B || !C) return -1;
  if (lda < (ta ? k : m) || ldb < (tb ? n : k) || ldc < m) return -1;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -1;
  if (!g_handle[dev]) {
    rocblas_status st = rocblas_create_handle(&g_handle[dev]);
    if (st != rocblas_status_success) return 1000 + (int)st;
  }
  rocblas_handle h = g_handle[dev];
  rocblas_status st = rocblas_set_stream(h, (hipStream_t)stream_);
  if (st != rocblas_status_success) return 1000 + (int)st;
  const float alpha = 1.f, beta = 0.f;
  st = rocblas_gemm_ex(h, ta ? rocblas_operation_transpose : rocblas_operation_none,
                       tb ? rocblas_operation_transpose : rocblas_operation_none, m, n, k, &alpha, A,
                       rocblas_datatype_bf16_r, (rocblas_int)lda, B, rocblas_datatype_bf16_r, (rocblas_int)ldb, &beta, C,
                       rocblas_datatype_f32_r, (rocblas_int)ldc, C, rocblas_datatype_f32_r, (rocblas_int)ldc,
                       rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
  return st == rocblas_status_success ? 0 : 1000 + (int)st;
}
