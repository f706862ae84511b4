// FOSCTTM / FOSFTTM neighbour counts (evaluator.py:14-71): for paired embeddings x, y [n x d]
//   D[i][j]   = sum_k (x[i][k] - y[j][k])^2          (direct form, fp32, k ascending -- evaluator.py:18-21)
//   lt12[i]   = #{ j : D[i][j] < D[i][i] }           (FOSCTTM domain 1 -> 2, evaluator.py:38)
//   lt21[i]   = #{ j : D[j][i] < D[i][i] }           (FOSCTTM domain 2 -> 1, evaluator.py:39)
//   gt12[i]   = #{ j : D[i][j] > D[i][i] }           (FOSFTTM, evaluator.py:62)
// The reference materialises the n x n x d difference tensor on the CPU (O(n^2 d) memory, unusable beyond ~20 k
// cells); here one workgroup owns a 64 x 64 tile of D, both operand tiles sit in LDS, each thread produces a 4 x 4
// patch, and the counts are accumulated with integer atomics (order-independent, exact).
#include "common.h"
#include "scpair_hip.h"

#define ET 64
#define EKC 64        // embedding columns staged per pass

__global__ __launch_bounds__(256) void fos_counts_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ y,
                                                         long ldy, int n, int d, const float* __restrict__ diag,
                                                         int* __restrict__ lt12, int* __restrict__ lt21,
                                                         int* __restrict__ gt12) {
  __shared__ float xs[ET][EKC + 1], ys[ET][EKC + 1];
  __shared__ int rc[ET], cc[ET], rg[ET];
  const int i0 = blockIdx.y * ET, j0 = blockIdx.x * ET, tid = threadIdx.x;
  if (tid < ET) rc[tid] = cc[tid] = rg[tid] = 0;
  const int ti = (tid >> 4) * 4, tj = (tid & 15) * 4;
  float acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = 0.f;
  for (int k0 = 0; k0 < d; k0 += EKC) {
    const int kc = min(EKC, d - k0);
    __syncthreads();
    for (int idx = tid; idx < ET * kc; idx += 256) {
      int r = idx / kc, k = idx % kc;
      xs[r][k] = (i0 + r < n) ? x[(long)(i0 + r) * ldx + k0 + k] : 0.f;
      ys[r][k] = (j0 + r < n) ? y[(long)(j0 + r) * ldy + k0 + k] : 0.f;
    }
    __syncthreads();
    for (int k = 0; k < kc; k++) {
      float xv[4], yv[4];
#pragma unroll
      for (int a = 0; a < 4; a++) xv[a] = xs[ti + a][k];
#pragma unroll
      for (int b = 0; b < 4; b++) yv[b] = ys[tj + b][k];
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
          float df = xv[a] - yv[b];
          acc[a][b] += df * df;
        }
    }
  }
#pragma unroll
  for (int a = 0; a < 4; a++) {
    const int i = i0 + ti + a;
    if (i >= n) continue;
    const float di = diag[i];
    int c_lt = 0, c_gt = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int j = j0 + tj + b;
      if (j < n) {
        c_lt += acc[a][b] < di;
        c_gt += acc[a][b] > di;
      }
    }
    if (c_lt) atomicAdd(&rc[ti + a], c_lt);
    if (c_gt) atomicAdd(&rg[ti + a], c_gt);
  }
#pragma unroll
  for (int b = 0; b < 4; b++) {
    const int j = j0 + tj + b;
    if (j >= n) continue;
    const float dj = diag[j];
    int c = 0;
#pragma unroll
    for (int a = 0; a < 4; a++)
      if (i0 + ti + a < n) c += acc[a][b] < dj;
    if (c) atomicAdd(&cc[tj + b], c);
  }
  __syncthreads();
  if (tid < ET) {
    if (i0 + tid < n) {
      if (rc[tid]) atomicAdd(&lt12[i0 + tid], rc[tid]);
      if (rg[tid]) atomicAdd(&gt12[i0 + tid], rg[tid]);
    }
    if (j0 + tid < n && cc[tid]) atomicAdd(&lt21[j0 + tid], cc[tid]);
  }
}

__global__ __launch_bounds__(256) void fos_diag_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ y,
                                                       long ldy, int n, int d, float* __restrict__ diag) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < d; k++) {
    float df = x[(long)i * ldx + k] - y[(long)i * ldy + k];
    s += df * df;
  }
  diag[i] = s;
}

extern "C" int scp_fos_counts(const float* x, long ldx, const float* y, long ldy, int n, int d, float* diag_ws,
                              int32_t* lt12, int32_t* lt21, int32_t* gt12, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SCP_CHECK(n > 0 && d > 0 && x && y && diag_ws && lt12 && lt21 && gt12, SCP_ERR_ARG, "scp_fos_counts: bad argument");
  SCP_HIP(hipMemsetAsync(lt12, 0, sizeof(int32_t) * (size_t)n, stream));
  SCP_HIP(hipMemsetAsync(lt21, 0, sizeof(int32_t) * (size_t)n, stream));
  SCP_HIP(hipMemsetAsync(gt12, 0, sizeof(int32_t) * (size_t)n, stream));
  hipLaunchKernelGGL(fos_diag_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, x, ldx, y, ldy, n, d, diag_ws);
  SCP_LAUNCH_CHECK();
  int t = (n + ET - 1) / ET;
  hipLaunchKernelGGL(fos_counts_kernel, dim3(t, t), dim3(256), 0, stream, x, ldx, y, ldy, n, d, diag_ws, lt12, lt21, gt12);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}
