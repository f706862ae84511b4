// Fused epilogue of one FCNN layer (utils.py:54-64): split-K slab reduce + bias -> [LayerNorm] ->
// LeakyReLU/ReLU -> Dropout, forward and backward.  (BatchNorm1d, which the reference places AFTER
// the activation, is a separate column-statistics kernel: bn_fwd / bn_bwd below.)
//
// A row lives in registers, so the forward reads the slabs once and writes xhat + out once, the backward
// reads dOut + xhat once and writes dY once.  Narrow rows (H <= 64) use one wave per row; wider rows use
// one 256-thread workgroup per row (16 waves per CU at batch 1024, several slab loads in flight per lane).
#include <mutex>

#include "common.h"
#include "scpair_hip.h"

template <int MAXR>
__global__ __launch_bounds__(256) void ln_act_fwd_kernel(const float* __restrict__ Y, long ldy, long slab_stride,
                                                         int nslab, const float* __restrict__ bias,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int B, int H, float eps,
                                                         float slope, float p_drop, SeedArg seed,
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
      a *= keep_scale(seed_of(seed), (uint64_t)row * H + c, p_drop);
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
                                                         float p_drop, SeedArg seed, float* __restrict__ dY, long ldy,
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
        float g = dOut[(long)row * ldo + c] * keep_scale(seed_of(seed), (uint64_t)row * H + c, p_drop);
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


// sum over the 256 threads of a workgroup, result in every thread (red: 4 floats per value)
__device__ __forceinline__ void block_allsum2(float& a, float& b, float (*red)[4]) {
  a = wave_sum(a);
  b = wave_sum(b);
  int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    red[0][w] = a;
    red[1][w] = b;
  }
  __syncthreads();
  a = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  b = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
}

// forward, one workgroup per row, H <= 256*CPT
template <int CPT>
__global__ __launch_bounds__(256) void ln_act_fwd_row_kernel(const float* __restrict__ Y, long ldy, long slab_stride,
                                                             int nslab, const float* __restrict__ bias,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int B, int H, float eps,
                                                             float slope, float p_drop, SeedArg seed,
                                                             float* __restrict__ xhat, long ldx,
                                                             float* __restrict__ rstd_out, float* __restrict__ out,
                                                             long ldo) {
  __shared__ float red[2][4];
  int row = blockIdx.x, tid = threadIdx.x;
  float v[CPT];
#pragma unroll
  for (int i = 0; i < CPT; i++) {
    int c = tid + 256 * i;
    v[i] = (bias && c < H) ? bias[c] : 0.f;
  }
  const float* yr = Y + (long)row * ldy;
  int sl = 0;
  // four slabs of loads in flight; slabs are still added in ascending order
  for (; sl + 4 <= nslab; sl += 4) {
    float t[4][CPT];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int i = 0; i < CPT; i++) {
        int c = tid + 256 * i;
        t[j][i] = (c < H) ? yr[(long)(sl + j) * slab_stride + c] : 0.f;
      }
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int i = 0; i < CPT; i++) v[i] += t[j][i];
  }
  for (; sl < nslab; sl++)
#pragma unroll
    for (int i = 0; i < CPT; i++) {
      int c = tid + 256 * i;
      if (c < H) v[i] += yr[(long)sl * slab_stride + c];
    }
  float rstd = 1.f, mean = 0.f;
  if (gamma) {
    float s = 0.f, dummy = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; i++) s += v[i];
    block_allsum2(s, dummy, red);
    mean = s / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; i++) {
      int c = tid + 256 * i;
      float d = (c < H) ? v[i] - mean : 0.f;
      q += d * d;
    }
    block_allsum2(q, dummy, red);
    rstd = 1.f / sqrtf(q / (float)H + eps);
    if (tid == 0 && rstd_out) rstd_out[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < CPT; i++) {
    int c = tid + 256 * i;
    if (c < H) {
      float xh = gamma ? (v[i] - mean) * rstd : v[i];
      float z = gamma ? xh * gamma[c] + beta[c] : xh;
      float a = z > 0.f ? z : z * slope;
      a *= keep_scale(seed_of(seed), (uint64_t)row * H + c, p_drop);
      if (xhat) xhat[(long)row * ldx + c] = xh;
      out[(long)row * ldo + c] = a;
    }
  }
}

// backward, one workgroup per row (rows strided over the grid); every thread owns CPT columns, so the
// column partials of dgamma / dbeta / dbias need no cross-wave step
template <int CPT>
__global__ __launch_bounds__(256) void ln_act_bwd_row_kernel(const float* __restrict__ dOut, long ldo,
                                                             const float* __restrict__ xhat, long ldx,
                                                             const float* __restrict__ rstd_in,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int B, int H,
                                                             float slope, float p_drop, SeedArg seed,
                                                             float* __restrict__ dY, long ldy,
                                                             float* __restrict__ partials /* [grid][3][H] */) {
  __shared__ float red[2][4];
  int tid = threadIdx.x;
  float ag[CPT], ab[CPT], ad[CPT], gam[CPT], bet[CPT];
#pragma unroll
  for (int i = 0; i < CPT; i++) {
    int c = tid + 256 * i;
    ag[i] = ab[i] = ad[i] = 0.f;
    gam[i] = (gamma && c < H) ? gamma[c] : 1.f;
    bet[i] = (gamma && c < H) ? beta[c] : 0.f;
  }
  for (int row = blockIdx.x; row < B; row += gridDim.x) {
    float dxh[CPT], xh[CPT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; i++) {
      int c = tid + 256 * i;
      dxh[i] = 0.f;
      xh[i] = 0.f;
      if (c < H) {
        float x = xhat[(long)row * ldx + c];
        float g = dOut[(long)row * ldo + c] * keep_scale(seed_of(seed), (uint64_t)row * H + c, p_drop);
        float z = gamma ? x * gam[i] + bet[i] : x;
        float dz = z > 0.f ? g : g * slope;
        xh[i] = x;
        if (gamma) {
          ag[i] += dz * x;
          ab[i] += dz;
          dxh[i] = dz * gam[i];
          s1 += dxh[i];
          s2 += dxh[i] * x;
        } else {
          dxh[i] = dz;
        }
      }
    }
    float rstd = 1.f;
    if (gamma) {
      block_allsum2(s1, s2, red);
      s1 /= (float)H;
      s2 /= (float)H;
      rstd = rstd_in[row];
    }
#pragma unroll
    for (int i = 0; i < CPT; i++) {
      int c = tid + 256 * i;
      if (c < H) {
        float d = gamma ? rstd * (dxh[i] - s1 - xh[i] * s2) : dxh[i];
        dY[(long)row * ldy + c] = d;
        ad[i] += d;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < CPT; i++) {
    int c = tid + 256 * i;
    if (c < H) {
      partials[((long)blockIdx.x * 3 + 0) * H + c] = ag[i];
      partials[((long)blockIdx.x * 3 + 1) * H + c] = ab[i];
      partials[((long)blockIdx.x * 3 + 2) * H + c] = ad[i];
    }
  }
}

// out[w][c] = sum_blk partials[blk][w][c]   (fixed order -> deterministic)
// workgroup = 64 columns x 16 row groups; group y sums blocks y, y+16, ...; the 16 group sums are added in order
__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float* __restrict__ partials, int nblk, int nw, int H,
                                                             float* __restrict__ o0, float* __restrict__ o1,
                                                             float* __restrict__ o2) {
  __shared__ float red[16][64];
  int x = threadIdx.x, y = threadIdx.y;
  int idx = blockIdx.x * 64 + x;
  bool ok = idx < nw * H;
  float s = 0.f;
  if (ok) s = strided_sum16(partials + idx, (long)nw * H, nblk, y);
  red[y][x] = s;
  __syncthreads();
  if (y == 0 && ok) {
    int w = idx / H, c = idx % H;
    float* o = w == 0 ? o0 : (w == 1 ? o1 : o2);
    if (o) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; k++) t += red[k][x];
      o[c] = t;
    }
  }
}

// upper bound of the backward grid (sizes the caller's partials workspace: [grid][3][H])
extern "C" int scp_ln_act_bwd_grid(int B) { return B > 256 ? 256 : (B < 1 ? 1 : B); }
static int ln_bwd_wave_grid(int B) {
  int g = (B + 3) / 4;
  return g > 128 ? 128 : (g < 1 ? 1 : g);
}

extern "C" int scp_ln_act_fwd(const float* Y, long ldy, long slab_stride, int nslab, const float* bias,
                              const float* gamma, const float* beta, int B, int H, float eps, float slope,
                              float p_drop, uint64_t seed_, const uint64_t* seed_dev, float* xhat, long ldx, float* rstd, float* out, long ldo,
                              void* stream_) {
  const SeedArg seed = {seed_, seed_dev};
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
#define LAUNCH_ROW(C)                                                                                               \
  hipLaunchKernelGGL((ln_act_fwd_row_kernel<C>), dim3(B), dim3(256), 0, stream, Y, ldy, slab_stride, nslab, bias,    \
                     gamma, beta, B, H, eps, slope, p_drop, seed, xhat, ldx, rstd, out, ldo)
  if (H <= 64) LAUNCH(1);
  else if (H <= 256) LAUNCH_ROW(1);
  else if (H <= 1024) LAUNCH_ROW(4);
  else LAUNCH_ROW(16);
#undef LAUNCH
#undef LAUNCH_ROW
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

extern "C" int scp_ln_act_bwd(const float* dOut, long ldo, const float* xhat, long ldx, const float* rstd,
                              const float* gamma, const float* beta, int B, int H, float slope, float p_drop,
                              uint64_t seed_, const uint64_t* seed_dev, float* dY, long ldy, float* dgamma, float* dbeta, float* dbias,
                              float* partials_ws, void* stream_) {
  const SeedArg seed = {seed_, seed_dev};
  hipStream_t stream = (hipStream_t)stream_;
  SCP_CHECK(B > 0 && H > 0, SCP_ERR_ARG, "scp_ln_act_bwd: B,H must be positive");
  SCP_CHECK(dOut && xhat && dY && partials_ws, SCP_ERR_ARG, "scp_ln_act_bwd: null pointer");
  SCP_CHECK(H <= 4096, SCP_ERR_SHAPE, "scp_ln_act_bwd: H=%d > 4096 unsupported", H);
  int grid = H <= 64 ? ln_bwd_wave_grid(B) : scp_ln_act_bwd_grid(B);
#define LAUNCH(R)                                                                                                  \
  hipLaunchKernelGGL((ln_act_bwd_kernel<R>), dim3(grid), dim3(256), 0, stream, dOut, ldo, xhat, ldx, rstd, gamma,   \
                     beta, B, H, slope, p_drop, seed, dY, ldy, partials_ws)
#define LAUNCH_ROW(C)                                                                                             \
  hipLaunchKernelGGL((ln_act_bwd_row_kernel<C>), dim3(grid), dim3(256), 0, stream, dOut, ldo, xhat, ldx, rstd,      \
                     gamma, beta, B, H, slope, p_drop, seed, dY, ldy, partials_ws)
  if (H <= 64) LAUNCH(1);
  else if (H <= 256) LAUNCH_ROW(1);
  else if (H <= 1024) LAUNCH_ROW(4);
  else LAUNCH_ROW(16);
#undef LAUNCH
#undef LAUNCH_ROW
  SCP_LAUNCH_CHECK();
  int n = 3 * H;
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((n + 63) / 64), dim3(64, 16), 0, stream, partials_ws, grid, 3, H,
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
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((cols + 63) / 64), dim3(64, 16), 0, stream, partials_ws, nchunk, 1, cols,
                     out, (float*)nullptr, (float*)nullptr);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

// scp_colsum for callers that cannot allocate (HIP graph capture, serving loops): the [32][cols] partial sums live
// in a file-scope device array instead of a caller workspace.  `scratch` is a host-side stand-in for the symbol; its
// device address is looked up once per device (each GPU holds its own copy) and only that address is ever passed to
// kernels.  Every slot that the reduction reads is rewritten by the partial kernel of the same call, so nothing has
// to be cleared between calls; two calls that overlap on one GPU (different streams) would race on the array.
#define SCP_COLSUM_NOALLOC_MAX_COLS 16384
static __device__ __align__(256) float scratch[32 * SCP_COLSUM_NOALLOC_MAX_COLS];      // 2 MiB per GPU
static float* g_scratch_ptr[64] = {nullptr};
static std::mutex g_scratch_mu;

static int colsum_scratch(float** out) {
  int dev = -1;
  SCP_HIP(hipGetDevice(&dev));
  SCP_CHECK(dev >= 0 && dev < 64, SCP_ERR_ARG, "scp_colsum_noalloc: device index %d out of range", dev);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    SCP_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    SCP_CHECK(p != nullptr, SCP_ERR_HIP, "scp_colsum_noalloc: null symbol address");
    g_scratch_ptr[dev] = (float*)p;
  }
  *out = g_scratch_ptr[dev];
  return SCP_OK;
}

extern "C" int scp_colsum_noalloc_max_cols(void) { return SCP_COLSUM_NOALLOC_MAX_COLS; }

extern "C" int scp_colsum_noalloc(const float* X, long ld, int rows, int cols, float* out, void* stream_) {
  SCP_CHECK(X && out && rows > 0 && cols > 0, SCP_ERR_ARG, "scp_colsum_noalloc: bad argument");
  SCP_CHECK(cols <= SCP_COLSUM_NOALLOC_MAX_COLS, SCP_ERR_SHAPE,
            "scp_colsum_noalloc: cols=%d exceeds the built-in workspace (max %d columns); use scp_colsum with a "
            "caller workspace", cols, SCP_COLSUM_NOALLOC_MAX_COLS);
  float* ws = nullptr;
  int rc = colsum_scratch(&ws);
  if (rc) return rc;
  return scp_colsum(X, ld, rows, cols, out, ws, stream_);
}

// ---- BatchNorm1d (training statistics / running statistics) + dropout --------------------------------------------
// The reference places BatchNorm1d AFTER the activation and before Dropout (utils.py:61-63).  Column statistics over
// the batch: one workgroup owns BN_C columns (BN_C x BN_R threads, BN_R row groups), two passes (mean, then centred
// second moment) like the CPU oracle, fixed-order combination of the row groups -> deterministic.  Every pass walks
// the rows in batches of BN_U independent loads (rows past the end re-read the last row and are masked): with one
// dependent load per trip the 1024-row walk of a 900-column layer took 70-96 us on 29 workgroups.
#define BN_C 16
#define BN_R 16
#define BN_U 8
__device__ __forceinline__ float bn_group_sum(float v, float (*red)[BN_C], int cx, int ry) {
  __syncthreads();
  red[ry][cx] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < BN_R; k++) s += red[k][cx];
  return s;
}

__global__ __launch_bounds__(BN_C * BN_R) void bn_fwd_kernel(const float* __restrict__ x, long ldx, int B, int H,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ run_mean, float* __restrict__ run_var,
                                                     int training, float momentum, float eps, float p_drop,
                                                     SeedArg seed, float* __restrict__ save_mean,
                                                     float* __restrict__ save_rstd, float* __restrict__ y, long ldy) {
  __shared__ float red[BN_R][BN_C];
  const int cx = threadIdx.x % BN_C, ry = threadIdx.x / BN_C;
  const int c = blockIdx.x * BN_C + cx;
  const bool ok = c < H;
  const int cc = min(c, H - 1);
  float mean = 0.f, rstd = 1.f;
  if (training) {
    float s = 0.f;
    for (int r0 = ry; r0 < B; r0 += BN_R * BN_U) {
      float t[BN_U];
#pragma unroll
      for (int u = 0; u < BN_U; u++) t[u] = x[(long)min(r0 + BN_R * u, B - 1) * ldx + cc];
#pragma unroll
      for (int u = 0; u < BN_U; u++) s += (r0 + BN_R * u < B) ? t[u] : 0.f;
    }
    mean = bn_group_sum(s, red, cx, ry) / (float)B;
    float q = 0.f;
    for (int r0 = ry; r0 < B; r0 += BN_R * BN_U) {
      float t[BN_U];
#pragma unroll
      for (int u = 0; u < BN_U; u++) t[u] = x[(long)min(r0 + BN_R * u, B - 1) * ldx + cc];
#pragma unroll
      for (int u = 0; u < BN_U; u++) {
        const float d = t[u] - mean;
        q += (r0 + BN_R * u < B) ? d * d : 0.f;
      }
    }
    float var = bn_group_sum(q, red, cx, ry) / (float)B;
    rstd = 1.f / sqrtf(var + eps);
    if (ok && ry == 0) {
      if (run_mean) run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
      if (run_var) run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * ((float)B / (float)max(B - 1, 1));
    }
  } else if (ok) {
    mean = run_mean[c];
    rstd = 1.f / sqrtf(run_var[c] + eps);
  }
  if (!ok) return;
  if (ry == 0) {
    save_mean[c] = mean;
    save_rstd[c] = rstd;
  }
  const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
  const uint64_t sd = seed_of(seed);
  for (int r0 = ry; r0 < B; r0 += BN_R * BN_U) {
    float t[BN_U];
#pragma unroll
    for (int u = 0; u < BN_U; u++) t[u] = x[(long)min(r0 + BN_R * u, B - 1) * ldx + c];
#pragma unroll
    for (int u = 0; u < BN_U; u++) {
      const int r = r0 + BN_R * u;
      if (r < B) y[(long)r * ldy + c] = ((t[u] - mean) * rstd * g + bt) * keep_scale(sd, (uint64_t)r * H + c, p_drop);
    }
  }
}

// dx, dgamma, dbeta.  training: dx = g*rstd*(dy' - mean(dy') - xhat*mean(dy'*xhat)); eval: dx = g*rstd*dy'
// (dy' = dy with the dropout mask of the forward pass recomputed from the seed)
__global__ __launch_bounds__(BN_C * BN_R) void bn_bwd_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ x,
                                                     long ldx, int B, int H, const float* __restrict__ gamma,
                                                     const float* __restrict__ save_mean,
                                                     const float* __restrict__ save_rstd, int training, float p_drop,
                                                     SeedArg seed, float* __restrict__ dx, long lddx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[BN_R][BN_C];
  const int cx = threadIdx.x % BN_C, ry = threadIdx.x / BN_C;
  const int c = blockIdx.x * BN_C + cx;
  const bool ok = c < H;
  const int cc = min(c, H - 1);
  const float mean = save_mean[cc], rstd = save_rstd[cc];
  const uint64_t sd = seed_of(seed);
  float s1 = 0.f, s2 = 0.f;
  for (int r0 = ry; r0 < B; r0 += BN_R * BN_U) {
    float td[BN_U], tx[BN_U];
#pragma unroll
    for (int u = 0; u < BN_U; u++) {
      const long r = min(r0 + BN_R * u, B - 1);
      td[u] = dy[r * lddy + cc];
      tx[u] = x[r * ldx + cc];
    }
#pragma unroll
    for (int u = 0; u < BN_U; u++) {
      const int r = r0 + BN_R * u;
      const float g = (r < B) ? td[u] * keep_scale(sd, (uint64_t)min(r, B - 1) * H + cc, p_drop) : 0.f;
      s1 += g;
      s2 += g * ((tx[u] - mean) * rstd);
    }
  }
  s1 = bn_group_sum(s1, red, cx, ry);
  s2 = bn_group_sum(s2, red, cx, ry);
  if (!ok) return;
  if (ry == 0) {
    if (dgamma) dgamma[c] = s2;
    if (dbeta) dbeta[c] = s1;
  }
  const float g_ = (gamma ? gamma[c] : 1.f) * rstd;
  const float m1 = training ? s1 / (float)B : 0.f, m2 = training ? s2 / (float)B : 0.f;
  for (int r0 = ry; r0 < B; r0 += BN_R * BN_U) {
    float td[BN_U], tx[BN_U];
#pragma unroll
    for (int u = 0; u < BN_U; u++) {
      const long r = min(r0 + BN_R * u, B - 1);
      td[u] = dy[r * lddy + c];
      tx[u] = x[r * ldx + c];
    }
#pragma unroll
    for (int u = 0; u < BN_U; u++) {
      const int r = r0 + BN_R * u;
      if (r < B) {
        const float g = td[u] * keep_scale(sd, (uint64_t)r * H + c, p_drop);
        dx[(long)r * lddx + c] = g_ * (g - m1 - (tx[u] - mean) * rstd * m2);
      }
    }
  }
}

extern "C" int scp_bn_fwd(const float* x, long ldx, int B, int H, const float* gamma, const float* beta, float* run_mean,
                          float* run_var, int training, float momentum, float eps, float p_drop, uint64_t seed_, const uint64_t* seed_dev,
                          float* save_mean, float* save_rstd, float* y, long ldy, void* stream_) {
  const SeedArg seed = {seed_, seed_dev};
  SCP_CHECK(B > 0 && H > 0 && x && y && save_mean && save_rstd, SCP_ERR_ARG, "scp_bn_fwd: bad argument");
  SCP_CHECK(training || (run_mean && run_var), SCP_ERR_ARG, "scp_bn_fwd: eval mode needs running statistics");
  SCP_CHECK((gamma == nullptr) == (beta == nullptr), SCP_ERR_ARG, "scp_bn_fwd: gamma and beta go together");
  SCP_CHECK(p_drop >= 0.f && p_drop < 1.f, SCP_ERR_ARG, "scp_bn_fwd: dropout p must be in [0,1)");
  hipLaunchKernelGGL(bn_fwd_kernel, dim3((H + BN_C - 1) / BN_C), dim3(BN_C * BN_R), 0, (hipStream_t)stream_, x, ldx, B, H, gamma, beta,
                     run_mean, run_var, training, momentum, eps, p_drop, seed, save_mean, save_rstd, y, ldy);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

extern "C" int scp_bn_bwd(const float* dy, long lddy, const float* x, long ldx, int B, int H, const float* gamma,
                          const float* save_mean, const float* save_rstd, int training, float p_drop, uint64_t seed_, const uint64_t* seed_dev,
                          float* dx, long lddx, float* dgamma, float* dbeta, void* stream_) {
  const SeedArg seed = {seed_, seed_dev};
  SCP_CHECK(B > 0 && H > 0 && dy && x && dx && save_mean && save_rstd, SCP_ERR_ARG, "scp_bn_bwd: bad argument");
  hipLaunchKernelGGL(bn_bwd_kernel, dim3((H + BN_C - 1) / BN_C), dim3(BN_C * BN_R), 0, (hipStream_t)stream_, dy, lddy, x, ldx, B, H, gamma,
                     save_mean, save_rstd, training, p_drop, seed, dx, lddx, dgamma, dbeta);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}
