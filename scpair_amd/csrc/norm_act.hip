// Fused epilogue of one FCNN layer (utils.py:54-64): split-K slab reduce + bias -> [LayerNorm] ->
// LeakyReLU/ReLU -> Dropout, forward and backward.  (BatchNorm1d, which the reference places AFTER
// the activation, is a separate column-statistics kernel: bn_fwd / bn_bwd below.)
//
// One wave per row; a row (H <= 64*MAXR columns) lives in registers, so the forward reads the slabs
// once and writes xhat + out once, the backward reads dOut + xhat once and writes dY once.
#include "common.h"
#include "scpair_hip.h"

template <int MAXR>
__global__ __launch_bounds__(256) void ln_act_fwd_kernel(const float* __restrict__ Y, long ldy, long slab_stride,
                                                         int nslab, const float* __restrict__ bias,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int B, int H, float eps,
                                                         float slope, float p_drop, uint64_t seed,
                                                         float* __restrict__ xhat, long ldx, float* __restrict__ rstd_out,
                                                         float* __restrict__ out, long ldo) {
  int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (row >= B) return;
  float v[MAXR];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXR; i++) {
    int c = lane + 64 * i;
    v[i] = (bias && c < H) ? bias[c] : 0.f;
  }
  // slab-major order: MAXR independent coalesced loads in flight per slab, slabs summed in a fixed order
  for (int sl = 0; sl < nslab; sl++) {
    const float* yr = Y + (long)sl * slab_stride + (long)row * ldy;
#pragma unroll
    for (int i = 0; i < MAXR; i++) {
      int c = lane + 64 * i;
      if (c < H) v[i] += yr[c];
    }
  }
#pragma unroll
  for (int i = 0; i < MAXR; i++) s += v[i];
  float rstd = 1.f, mean = 0.f;
  if (gamma) {
    mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXR; i++) {
      int c = lane + 64 * i;
      float d = (c < H) ? v[i] - mean : 0.f;
      q += d * d;
    }
    float var = wave_sum(q) / (float)H;
    rstd = 1.f / sqrtf(var + eps);
    if (lane == 0 && rstd_out) rstd_out[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < MAXR; i++) {
    int c = lane + 64 * i;
    if (c < H) {
      float xh = gamma ? (v[i] - mean) * rstd : v[i];
      float z = gamma ? xh * gamma[c] + beta[c] : xh;
      float a = z > 0.f ? z : z * slope;
      a *= keep_scale(seed, (uint64_t)row * H + c, p_drop);
      if (xhat) xhat[(long)row * ldx + c] = xh;
      out[(long)row * ldo + c] = a;
    }
  }
}

// backward: dOut -> dY (grad wrt the pre-LN linear output) + per-block column partials of
// dgamma, dbeta and dbias(=column sums of dY)
template <int MAXR>
__global__ __launch_bounds__(256) void ln_act_bwd_kernel(const float* __restrict__ dOut, long ldo,
                                                         const float* __restrict__ xhat, long ldx,
                                                         const float* __restrict__ rstd_in,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int B, int H, float slope,
                                                         float p_drop, uint64_t seed, float* __restrict__ dY, long ldy,
                                                         float* __restrict__ partials /* [grid][3][H] */) {
  __shared__ float red[3][4][64];
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int nwaves = gridDim.x * 4;
  float ag[MAXR], ab[MAXR], ad[MAXR];
#pragma unroll
  for (int i = 0; i < MAXR; i++) ag[i] = ab[i] = ad[i] = 0.f;
  for (int row = blockIdx.x * 4 + wave; row < B; row += nwaves) {
    float dxh[MAXR], xh[MAXR];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXR; i++) {
      int c = lane + 64 * i;
      dxh[i] = 0.f;
      xh[i] = 0.f;
      if (c < H) {
        float x = xhat[(long)row * ldx + c];
        float g = dOut[(long)row * ldo + c] * keep_scale(seed, (uint64_t)row * H + c, p_drop);
        float z = gamma ? x * gamma[c] + beta[c] : x;
        float dz = z > 0.f ? g : g * slope;
        xh[i] = x;
        if (gamma) {
          ag[i] += dz * x;
          ab[i] += dz;
          dxh[i] = dz * gamma[c];
          s1 += dxh[i];
          s2 += dxh[i] * x;
        } else {
          dxh[i] = dz;
        }
      }
    }
    float rstd = 1.f;
    if (gamma) {
      s1 = wave_sum(s1) / (float)H;
      s2 = wave_sum(s2) / (float)H;
      rstd = rstd_in[row];
    }
#pragma unroll
    for (int i = 0; i < MAXR; i++) {
      int c = lane + 64 * i;
      if (c < H) {
        float d = gamma ? rstd * (dxh[i] - s1 - xh[i] * s2) : dxh[i];
        dY[(long)row * ldy + c] = d;
        ad[i] += d;
      }
    }
  }
  // cross-wave reduce of the column partials, one 64-column strip at a time
#pragma unroll
  for (int i = 0; i < MAXR; i++) {
    if (64 * i >= H) break;
    __syncthreads();
    red[0][wave][lane] = ag[i];
    red[1][wave][lane] = ab[i];
    red[2][wave][lane] = ad[i];
    __syncthreads();
    if (wave < 3) {
      int c = lane + 64 * i;
      if (c < H) {
        float t = red[wave][0][lane] + red[wave][1][lane] + red[wave][2][lane] + red[wave][3][lane];
        partials[((long)blockIdx.x * 3 + wave) * H + c] = t;
      }
    }
  }
}

// out[w][c] = sum_blk partials[blk][w][c]   (fixed order -> deterministic)
__global__ void colsum_reduce_kernel(const float* __restrict__ partials, int nblk, int nw, int H,
                                     float* __restrict__ o0, float* __restrict__ o1, float* __restrict__ o2) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nw * H) return;
  int w = idx / H, c = idx % H;
  float* o = w == 0 ? o0 : (w == 1 ? o1 : o2);
  if (!o) return;
  float s = 0.f;
  for (int b = 0; b < nblk; b++) s += partials[((long)b * nw + w) * H + c];
  o[c] = s;
}

extern "C" int scp_ln_act_bwd_grid(int B) {
  int g = (B + 3) / 4;
  return g > 128 ? 128 : (g < 1 ? 1 : g);
}

extern "C" int scp_ln_act_fwd(const float* Y, long ldy, long slab_stride, int nslab, const float* bias,
                              const float* gamma, const float* beta, int B, int H, float eps, float slope,
                              float p_drop, uint64_t seed, float* xhat, long ldx, float* rstd, float* out, long ldo,
                              void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SCP_CHECK(B > 0 && H > 0, SCP_ERR_ARG, "scp_ln_act_fwd: B,H must be positive");
  SCP_CHECK(Y && out, SCP_ERR_ARG, "scp_ln_act_fwd: null pointer");
  SCP_CHECK((gamma == nullptr) == (beta == nullptr), SCP_ERR_ARG, "scp_ln_act_fwd: gamma and beta go together");
  SCP_CHECK(H <= 4096, SCP_ERR_SHAPE, "scp_ln_act_fwd: H=%d > 4096 unsupported", H);
  SCP_CHECK(p_drop >= 0.f && p_drop < 1.f, SCP_ERR_ARG, "scp_ln_act_fwd: dropout p must be in [0,1)");
  int grid = (B + 3) / 4;
#define LAUNCH(R)                                                                                                    \
  hipLaunchKernelGGL((ln_act_fwd_kernel<R>), dim3(grid), dim3(256), 0, stream, Y, ldy, slab_stride, nslab, bias,     \
                     gamma, beta, B, H, eps, slope, p_drop, seed, xhat, ldx, rstd, out, ldo)
  if (H <= 64) LAUNCH(1);
  else if (H <= 256) LAUNCH(4);
  else if (H <= 1024) LAUNCH(16);
  else LAUNCH(64);
#undef LAUNCH
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

extern "C" int scp_ln_act_bwd(const float* dOut, long ldo, const float* xhat, long ldx, const float* rstd,
                              const float* gamma, const float* beta, int B, int H, float slope, float p_drop,
                              uint64_t seed, float* dY, long ldy, float* dgamma, float* dbeta, float* dbias,
                              float* partials_ws, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SCP_CHECK(B > 0 && H > 0, SCP_ERR_ARG, "scp_ln_act_bwd: B,H must be positive");
  SCP_CHECK(dOut && xhat && dY && partials_ws, SCP_ERR_ARG, "scp_ln_act_bwd: null pointer");
  SCP_CHECK(H <= 4096, SCP_ERR_SHAPE, "scp_ln_act_bwd: H=%d > 4096 unsupported", H);
  int grid = scp_ln_act_bwd_grid(B);
#define LAUNCH(R)                                                                                                  \
  hipLaunchKernelGGL((ln_act_bwd_kernel<R>), dim3(grid), dim3(256), 0, stream, dOut, ldo, xhat, ldx, rstd, gamma,   \
                     beta, B, H, slope, p_drop, seed, dY, ldy, partials_ws)
  if (H <= 64) LAUNCH(1);
  else if (H <= 256) LAUNCH(4);
  else if (H <= 1024) LAUNCH(16);
  else LAUNCH(64);
#undef LAUNCH
  SCP_LAUNCH_CHECK();
  int n = 3 * H;
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, partials_ws, grid, 3, H,
                     dgamma, dbeta, dbias);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

// ---- generic helpers ---------------------------------------------------------------------------
// out[r][c] = sum_s slabs[s][r][c] (+ bias[c])
__global__ void slab_reduce_kernel(const float* __restrict__ S, long slab_stride, int nslab, long n, int ncol, long ld,
                                   const float* __restrict__ bias, float* __restrict__ out, long ldo) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  long r = i / ncol;
  int c = (int)(i % ncol);
  float a = bias ? bias[c] : 0.f;
  for (int s = 0; s < nslab; s++) a += S[(long)s * slab_stride + r * ld + c];
  out[r * ldo + c] = a;
}

extern "C" int scp_slab_reduce(const float* slabs, long slab_stride, int nslab, int rows, int cols, long ld,
                               const float* bias, float* out, long ldo, void* stream_) {
  SCP_CHECK(slabs && out && rows > 0 && cols > 0 && nslab > 0, SCP_ERR_ARG, "scp_slab_reduce: bad argument");
  long n = (long)rows * cols;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, slabs,
                     slab_stride, nslab, n, cols, ld, bias, out, ldo);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

// column sums of a [rows][cols] matrix (bias gradients): two-stage, deterministic
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, long ld, int rows, int cols,
                                                             float* __restrict__ partials) {
  int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  int nchunk = gridDim.y;
  int per = (rows + nchunk - 1) / nchunk;
  int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  float s = 0.f;
  for (int r = r0; r < r1; r++) s += X[(long)r * ld + c];
  partials[(long)blockIdx.y * cols + c] = s;
}

extern "C" int scp_colsum(const float* X, long ld, int rows, int cols, float* out, float* partials_ws /* [32][cols] */,
                          void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SCP_CHECK(X && out && partials_ws && rows > 0 && cols > 0, SCP_ERR_ARG, "scp_colsum: bad argument");
  int nchunk = rows < 32 ? 1 : 32;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((cols + 255) / 256, nchunk), dim3(256), 0, stream, X, ld, rows, cols,
                     partials_ws);
  SCP_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((cols + 255) / 256), dim3(256), 0, stream, partials_ws, nchunk, 1, cols,
                     out, (float*)nullptr, (float*)nullptr);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}
