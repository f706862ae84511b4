// Small fused elementwise / reduction kernels of the hot path:
//   masked MSE (loss.py:22-27), reparameterisation + KL (pairedVAE.py:51-58, utils.py:186),
//   decoder inference heads for predict() (model.py:558-591), deterministic partial-sum finalisers.
#include "common.h"
#include "scpair_hip.h"

// ---- deterministic reduction of per-block double partials ---------------------------------------
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double* __restrict__ part, int n, int ncomp,
                                                              double* __restrict__ out) {
  __shared__ double red[8];
  int comp = blockIdx.x;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += part[(long)i * ncomp + comp];
  s = block_sum_d(s, red);
  if (threadIdx.x == 0) out[comp] = s;
}

int scp_reduce_partials(const double* part, int n, int ncomp, double* out, hipStream_t stream) {
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(ncomp), dim3(256), 0, stream, part, n, ncomp, out);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

// ---- masked MSE ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mse_fwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ mu,
                                                      long ldm, int rows, int cols, double* __restrict__ part) {
  __shared__ double red[8];
  long n = (long)rows * cols;
  double s = 0.0, cnt = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    long r = i / cols;
    int c = (int)(i % cols);
    float xv = x[r * ldx + c];
    if (xv == xv) {
      float d = mu[r * ldm + c] - xv;
      s += (double)(d * d);
      cnt += 1.0;
    }
  }
  s = block_sum_d(s, red);
  cnt = block_sum_d(cnt, red);
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = s;
    part[2 * blockIdx.x + 1] = cnt;
  }
}

__global__ void mse_finish_kernel(const double* __restrict__ sums, float* __restrict__ loss_cnt) {
  loss_cnt[0] = (float)(sums[0] / sums[1]);
  loss_cnt[1] = (float)sums[1];
}

__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ mu,
                                                      long ldm, int rows, int cols, const float* __restrict__ loss_cnt,
                                                      const float* __restrict__ gout, float* __restrict__ dmu, long ldd) {
  long n = (long)rows * cols;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  long r = i / cols;
  int c = (int)(i % cols);
  float xv = x[r * ldx + c];
  float sc = 2.f * (gout ? gout[0] : 1.f) / loss_cnt[1];
  dmu[r * ldd + c] = (xv == xv) ? sc * (mu[r * ldm + c] - xv) : 0.f;
}

// loss_cnt[0] = mean((mu-x)^2 over non-NaN x), loss_cnt[1] = count.  ws: >= 2*256+2 doubles
extern "C" int scp_mse_masked_fwd(const float* x, long ldx, const float* mu, long ldm, int rows, int cols,
                                  float* loss_cnt, double* ws, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SCP_CHECK(x && mu && loss_cnt && ws && rows > 0 && cols > 0, SCP_ERR_ARG, "scp_mse_masked_fwd: bad argument");
  long n = (long)rows * cols;
  int grid = (int)((n + 255) / 256);
  if (grid > 256) grid = 256;
  hipLaunchKernelGGL(mse_fwd_kernel, dim3(grid), dim3(256), 0, stream, x, ldx, mu, ldm, rows, cols, ws);
  SCP_LAUNCH_CHECK();
  int rc = scp_reduce_partials(ws, grid, 2, ws + 512, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(1), 0, stream, ws + 512, loss_cnt);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

extern "C" int scp_mse_masked_bwd(const float* x, long ldx, const float* mu, long ldm, int rows, int cols,
                                  const float* loss_cnt, const float* gout, float* dmu, long ldd, void* stream_) {
  SCP_CHECK(x && mu && loss_cnt && dmu && rows > 0 && cols > 0, SCP_ERR_ARG, "scp_mse_masked_bwd: bad argument");
  long n = (long)rows * cols;
  hipLaunchKernelGGL(mse_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, x, ldx, mu,
                     ldm, rows, cols, loss_cnt, gout, dmu, ldd);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

// ---- reparameterisation + KL ------------------------------------------------------------------------
// sigma = softplus(s_raw) + 1e-7 ; z = mu + eps*sigma ; kl = mean(-log sigma + (sigma^2 + mu^2)/2 - 1/2)
__global__ __launch_bounds__(256) void reparam_kl_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ sraw,
                                                             const float* __restrict__ eps, long n,
                                                             float* __restrict__ sigma, float* __restrict__ z,
                                                             double* __restrict__ part) {
  __shared__ double red[8];
  double s = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float m = mu[i];
    float sg = softplus_t(sraw[i]) + 1e-7f;
    sigma[i] = sg;
    z[i] = m + eps[i] * sg;
    s += (double)(-logf(sg) + 0.5f * (sg * sg + m * m) - 0.5f);
  }
  s = block_sum_d(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ void kl_finish_kernel(const double* __restrict__ sums, long n, float* __restrict__ kl) {
  kl[0] = (float)(sums[0] / (double)n);
}

// dmu = dz + c*mu ; dsraw = (dz*eps + c*(sigma - 1/sigma)) * sigmoid(sraw) [threshold branch: 1], c = gkl/n
// dsigma_extra: optional extra gradient wrt sigma (the reference returns sigma; nothing consumes it)
__global__ __launch_bounds__(256) void reparam_kl_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ sraw,
                                                             const float* __restrict__ eps,
                                                             const float* __restrict__ sigma,
                                                             const float* __restrict__ dz, const float* __restrict__ dmu_in,
                                                             const float* __restrict__ gkl, long n,
                                                             float* __restrict__ dmu, float* __restrict__ dsraw) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float c = (gkl ? gkl[0] : 0.f) / (float)n;
  float sg = sigma[i];
  float g = dz ? dz[i] : 0.f;
  dmu[i] = g + c * mu[i] + (dmu_in ? dmu_in[i] : 0.f);
  float ds = g * eps[i] + c * (sg - 1.f / sg);
  float sr = sraw[i];
  float dsp = sr > 20.f ? 1.f : sigmoidf_(sr);
  dsraw[i] = ds * dsp;
}

extern "C" int scp_reparam_kl_fwd(const float* mu, const float* sraw, const float* eps, long n, float* sigma, float* z,
                                  float* kl, double* ws, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SCP_CHECK(mu && sraw && eps && sigma && z && kl && ws && n > 0, SCP_ERR_ARG, "scp_reparam_kl_fwd: bad argument");
  int grid = (int)((n + 255) / 256);
  if (grid > 256) grid = 256;
  hipLaunchKernelGGL(reparam_kl_fwd_kernel, dim3(grid), dim3(256), 0, stream, mu, sraw, eps, n, sigma, z, ws);
  SCP_LAUNCH_CHECK();
  int rc = scp_reduce_partials(ws, grid, 1, ws + 512, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(kl_finish_kernel, dim3(1), dim3(1), 0, stream, ws + 512, n, kl);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

extern "C" int scp_reparam_kl_bwd(const float* mu, const float* sraw, const float* eps, const float* sigma,
                                  const float* dz, const float* dmu_in, const float* gkl, long n, float* dmu,
                                  float* dsraw, void* stream_) {
  SCP_CHECK(mu && sraw && eps && sigma && dmu && dsraw && n > 0, SCP_ERR_ARG, "scp_reparam_kl_bwd: bad argument");
  hipLaunchKernelGGL(reparam_kl_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, mu,
                     sraw, eps, sigma, dz, dmu_in, gkl, n, dmu, dsraw);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

// ---- decoder inference heads (predict / reconstructions) ---------------------------------------------
// Bernoulli: out = sigmoid(z) [* sigmoid(l_row)] [* sigmoid(fs_col)]   (model.py:173-189), in place allowed
__global__ __launch_bounds__(256) void ber_prob_kernel(const float* __restrict__ z, long ldz, int rows, int cols,
                                                       const float* __restrict__ lrow, int lrow_n,
                                                       const float* __restrict__ fs, void* __restrict__ out, long ldo,
                                                       int out_bf16) {
  long n = (long)rows * cols;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  long r = i / cols;
  int c = (int)(i % cols);
  float p = sigmoidf_(z[r * ldz + c]);
  if (lrow) p *= sigmoidf_(lrow[lrow_n == 1 ? 0 : r]);
  if (fs) p *= sigmoidf_(fs[c]);
  if (out_bf16) ((__bf16*)out)[r * ldo + c] = (__bf16)p;
  else ((float*)out)[r * ldo + c] = p;
}

extern "C" int scp_ber_prob(const float* z, long ldz, int rows, int cols, const float* lrow, int lrow_n, const float* fs,
                            void* out, long ldo, int out_bf16, void* stream_) {
  SCP_CHECK(z && out && rows > 0 && cols > 0, SCP_ERR_ARG, "scp_ber_prob: bad argument");
  SCP_CHECK(!(out_bf16 && (const void*)z == out), SCP_ERR_ARG, "scp_ber_prob: bf16 output cannot be written in place");
  long n = (long)rows * cols;
  hipLaunchKernelGGL(ber_prob_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, z, ldz, rows,
                     cols, lrow, lrow_n, fs, out, ldo, out_bf16);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

// Raw: a = relu(z) ; scale = softmax_row(a [* sigmoid(fs)]) ; optional theta = exp(a) ; one block per row
__global__ __launch_bounds__(256) void relu_softmax_kernel(const float* __restrict__ z, long ldz, int cols,
                                                           const float* __restrict__ fs, int apply_relu,
                                                           void* __restrict__ scale, long lds_,
                                                           float* __restrict__ theta, long ldt, int out_bf16) {
  __shared__ float red[8];
  __shared__ float bc;
  long r = blockIdx.x;
  float m = -INFINITY;
  for (int c = threadIdx.x; c < cols; c += 256) {
    float a = z[r * ldz + c];
    if (apply_relu) a = fmaxf(a, 0.f);
    if (fs) a *= sigmoidf_(fs[c]);
    m = fmaxf(m, a);
  }
  m = wave_max(m);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) bc = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  m = bc;
  float s = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) {
    float a = z[r * ldz + c];
    if (apply_relu) a = fmaxf(a, 0.f);
    if (fs) a *= sigmoidf_(fs[c]);
    s += expf(a - m);
  }
  s = block_sum(s, red);
  __syncthreads();
  if (threadIdx.x == 0) bc = s;
  __syncthreads();
  float inv = 1.f / bc;
  for (int c = threadIdx.x; c < cols; c += 256) {
    float a0 = z[r * ldz + c];
    if (apply_relu) a0 = fmaxf(a0, 0.f);
    float a = fs ? a0 * sigmoidf_(fs[c]) : a0;
    float sv = expf(a - m) * inv;
    if (out_bf16) ((__bf16*)scale)[r * lds_ + c] = (__bf16)sv;
    else ((float*)scale)[r * lds_ + c] = sv;
    if (theta) theta[r * ldt + c] = expf(a0);
  }
}

extern "C" int scp_relu_softmax_rows(const float* z, long ldz, int rows, int cols, const float* fs, int apply_relu,
                                     void* scale, long lds_, float* theta, long ldt, int out_bf16, void* stream_) {
  SCP_CHECK(z && scale && rows > 0 && cols > 0, SCP_ERR_ARG, "scp_relu_softmax_rows: bad argument");
  hipLaunchKernelGGL(relu_softmax_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream_, z, ldz, cols, fs, apply_relu,
                     scale, lds_, theta, ldt, out_bf16);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

// ---- masked BCE on probabilities (stand-alone form of loss.py:36-41) -----------------------------------
__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ x, const float* __restrict__ p, long n,
                                                      double* __restrict__ part) {
  __shared__ double red[8];
  double s = 0.0, cnt = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float xv = x[i];
    if (xv == xv) {
      float q = p[i];
      s -= (double)(xv * fmaxf(logf(q), -100.f) + (1.f - xv) * fmaxf(logf(1.f - q), -100.f));
      cnt += 1.0;
    }
  }
  s = block_sum_d(s, red);
  cnt = block_sum_d(cnt, red);
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = s;
    part[2 * blockIdx.x + 1] = cnt;
  }
}

__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ p, long n,
                                                      const float* __restrict__ loss_cnt, const float* __restrict__ gout,
                                                      float* __restrict__ dp) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float xv = x[i], q = p[i];
  float sc = (gout ? gout[0] : 1.f) / loss_cnt[1];
  dp[i] = (xv == xv) ? sc * (q - xv) / fmaxf(q * (1.f - q), 1e-12f) : 0.f;
}

extern "C" int scp_bce_masked_fwd(const float* x, const float* p, long n, float* loss_cnt, double* ws, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SCP_CHECK(x && p && loss_cnt && ws && n > 0, SCP_ERR_ARG, "scp_bce_masked_fwd: bad argument");
  int grid = (int)((n + 255) / 256);
  if (grid > 256) grid = 256;
  hipLaunchKernelGGL(bce_fwd_kernel, dim3(grid), dim3(256), 0, stream, x, p, n, ws);
  SCP_LAUNCH_CHECK();
  int rc = scp_reduce_partials(ws, grid, 2, ws + 512, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(1), 0, stream, ws + 512, loss_cnt);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

extern "C" int scp_bce_masked_bwd(const float* x, const float* p, long n, const float* loss_cnt, const float* gout,
                                  float* dp, void* stream_) {
  SCP_CHECK(x && p && loss_cnt && dp && n > 0, SCP_ERR_ARG, "scp_bce_masked_bwd: bad argument");
  hipLaunchKernelGGL(bce_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, x, p, n,
                     loss_cnt, gout, dp);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

// ---- element-wise NB / ZINB log-likelihood (scVI_distribution.py:29-105), stand-alone form ---------------
// theta may be a per-feature vector (theta_cols = number of columns, theta index = i % cols) or full size.
__global__ __launch_bounds__(256) void nbll_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mu,
                                                       const float* __restrict__ theta, int theta_vec, int cols,
                                                       const float* __restrict__ pi, long n, float* __restrict__ out) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float eps = 1e-8f;
  float xv = x[i], m = mu[i], th = theta[theta_vec ? (i % cols) : i];
  float s = th + m + eps;
  float lt = logf(th + eps), ls = logf(s);
  if (!pi) {
    // reference evaluates every term for every element (x = 0 makes the lgamma terms cancel)
    out[i] = th * (lt - ls) + xv * (logf(m + eps) - ls) + lgammaf(xv + th) - lgammaf(th) - lgammaf(xv + 1.f);
  } else {
    float pv = pi[i];
    float sp = softplus_t(-pv);
    float t = -pv + th * (lt - ls);
    float c0 = softplus_t(t) - sp;
    float c1 = -sp + t + xv * (logf(m + eps) - ls) + lgammaf(xv + th) - lgammaf(th) - lgammaf(xv + 1.f);
    // both branches are computed and 0/1 weighted, as the reference does (0 * inf = NaN is reproduced)
    out[i] = (xv < eps ? 1.f : 0.f) * c0 + (xv > eps ? 1.f : 0.f) * c1;
  }
}

__global__ __launch_bounds__(256) void nbll_bwd_kernel(const float* __restrict__ x, const float* __restrict__ mu,
                                                       const float* __restrict__ theta, int theta_vec, int cols,
                                                       const float* __restrict__ pi, const float* __restrict__ g, long n,
                                                       float* __restrict__ dmu, float* __restrict__ dtheta,
                                                       float* __restrict__ dpi) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float eps = 1e-8f;
  float xv = x[i], m = mu[i], th = theta[theta_vec ? (i % cols) : i], gv = g[i];
  float s = th + m + eps;
  float lt = logf(th + eps), ls = logf(s);
  float dt_dmu = -th / s;
  float dt_dth = lt - ls + th / (th + eps) - th / s;
  float dm, dth, dp = 0.f;
  bool nz = xv > eps, zero = xv < eps;
  float dig = nz ? digammaf_pos(xv + th) - digammaf_pos(th) : 0.f;
  if (!pi) {
    dm = dt_dmu + xv * (1.f / (m + eps) - 1.f / s);
    dth = dt_dth - xv / s + dig;
  } else {
    float pv = pi[i];
    float t = -pv + th * (lt - ls);
    float sig_t = t > 20.f ? 1.f : sigmoidf_(t);
    float sig_npi = (-pv) > 20.f ? 1.f : sigmoidf_(-pv);
    if (zero) {
      dm = sig_t * dt_dmu;
      dth = sig_t * dt_dth;
      dp = -sig_t + sig_npi;
    } else if (nz) {
      dm = dt_dmu + xv * (1.f / (m + eps) - 1.f / s);
      dth = dt_dth - xv / s + dig;
      dp = sig_npi - 1.f;
    } else {
      dm = dth = dp = 0.f;
    }
  }
  dmu[i] = gv * dm;
  dtheta[i] = gv * dth;
  if (dpi) dpi[i] = gv * dp;
}

extern "C" int scp_nbll_fwd(const float* x, const float* mu, const float* theta, int theta_vec, int cols,
                            const float* pi, long n, float* out, void* stream_) {
  SCP_CHECK(x && mu && theta && out && n > 0 && cols > 0, SCP_ERR_ARG, "scp_nbll_fwd: bad argument");
  hipLaunchKernelGGL(nbll_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, x, mu, theta,
                     theta_vec, cols, pi, n, out);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

extern "C" int scp_nbll_bwd(const float* x, const float* mu, const float* theta, int theta_vec, int cols,
                            const float* pi, const float* g, long n, float* dmu, float* dtheta, float* dpi,
                            void* stream_) {
  SCP_CHECK(x && mu && theta && g && dmu && dtheta && n > 0 && cols > 0, SCP_ERR_ARG, "scp_nbll_bwd: bad argument");
  SCP_CHECK((pi == nullptr) == (dpi == nullptr), SCP_ERR_ARG, "scp_nbll_bwd: pi and dpi go together");
  hipLaunchKernelGGL(nbll_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, x, mu, theta,
                     theta_vec, cols, pi, g, n, dmu, dtheta, dpi);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}
