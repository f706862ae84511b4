// rocBLAS bf16 GEMM, kept as the COMPARISON for the hand-written wide GEMM (wgemm.hip, scp_gemm_bf16) that runs the
// wide nn.Linear layers (ZINB G x G model.py:228, VAE 900 <-> G pairedVAE.py:111-119) in precision mode 'bf16':
// benchmarks/wgemm_bench.py times both on the same operands.  No product path calls this entry point.
#include <rocblas/rocblas.h>
#include "common.h"
#include "scpair_hip.h"

static rocblas_handle g_handle[16] = {};

static int get_handle(rocblas_handle* out) {
  int dev = 0;
  SCP_HIP(hipGetDevice(&dev));
  SCP_CHECK(dev >= 0 && dev < 16, SCP_ERR_ARG, "scp_gemm_lib_bf16: device index %d out of range", dev);
  if (!g_handle[dev]) {
    rocblas_status st = rocblas_create_handle(&g_handle[dev]);
    SCP_CHECK(st == rocblas_status_success, SCP_ERR_HIP, "rocblas_create_handle failed (%d)", (int)st);
  }
  *out = g_handle[dev];
  return SCP_OK;
}

// Column-major (BLAS) convention:  C[m x n] = op(A)[m x k] * op(B)[k x n],  ta / tb: 0 = N, 1 = T.
// A row-major matrix R[r][c] with leading dimension ld is the column-major matrix [c x r] with the same ld, so
//   y  = x W^T   (row-major [M,N]) : ta=1 tb=0  m=N n=M k=K  A=W(ld K)   B=x(ld K)   C=y(ld N)
//   dx = dy W    (row-major [M,K]) : ta=0 tb=0  m=K n=M k=N  A=W(ld K)   B=dy(ld N)  C=dx(ld K)
//   dW = dy^T x  (row-major [N,K]) : ta=0 tb=1  m=K n=N k=M  A=x(ld K)   B=dy(ld N)  C=dW(ld K)
extern "C" int scp_gemm_lib_bf16(int ta, int tb, int m, int n, int k, const void* A, long lda, const void* B, long ldb,
                                 float* C, long ldc, void* stream_) {
  SCP_CHECK(m > 0 && n > 0 && k > 0, SCP_ERR_ARG, "scp_gemm_lib_bf16: m,n,k must be positive (%d,%d,%d)", m, n, k);
  SCP_CHECK(A && B && C, SCP_ERR_ARG, "scp_gemm_lib_bf16: null pointer");
  SCP_CHECK(lda >= (ta ? k : m) && ldb >= (tb ? n : k) && ldc >= m, SCP_ERR_ARG,
            "scp_gemm_lib_bf16: leading dimension too small");
  rocblas_handle h;
  int rc = get_handle(&h);
  if (rc) return rc;
  rocblas_status st = rocblas_set_stream(h, (hipStream_t)stream_);
  SCP_CHECK(st == rocblas_status_success, SCP_ERR_HIP, "rocblas_set_stream failed (%d)", (int)st);
  const float alpha = 1.f, beta = 0.f;
  st = rocblas_gemm_ex(h, ta ? rocblas_operation_transpose : rocblas_operation_none,
                       tb ? rocblas_operation_transpose : rocblas_operation_none, m, n, k, &alpha, A,
                       rocblas_datatype_bf16_r, (rocblas_int)lda, B, rocblas_datatype_bf16_r, (rocblas_int)ldb, &beta, C,
                       rocblas_datatype_f32_r, (rocblas_int)ldc, C, rocblas_datatype_f32_r, (rocblas_int)ldc,
                       rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
  SCP_CHECK(st == rocblas_status_success, SCP_ERR_HIP, "rocblas_gemm_ex failed (%d)", (int)st);
  return SCP_OK;
}
