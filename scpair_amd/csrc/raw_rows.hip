// Row kernels for the (ZI)NB decoders whose pre-activations are materialised: the ZINB predictor, whose G x G
// zero-inflation layer needs px_mu as a GEMM operand (model.py:228-259), and the VAE decoder heads.
//
//   z  [B][G]  pre-activation of output_mean's Linear          pi [B][G] zero-inflation logits (null: plain NB)
//   predictor (tz null): a = relu(z), scale = softmax_row(a), mu = L[b] * scale, theta = exp(a)  ('feature-cell')
//     with a feature factor (fs [G], model.py:239-241): scale = softmax_row(a * sigmoid(fs)); with dispersion 'feature'
//     (thp [G], model.py:249-250): theta = exp(thp); their gradients are column sums over the cells, taken by a second,
//     column-parallel kernel over fixed row chunks and added in chunk order (no atomics: bitwise reproducible)
//   VAE decoder (tz [B][G] = output of its dispersion Linear, pairedVAE.py:147-187): a = z (no relu),
//                        scale = softmax_row(a), mu = L[b] * scale, theta = softplus(tz) + 1e-8
//   loss = -mean_{b,j} log (ZI)NB(y; mu, theta, pi)            scVI_distribution.py:29-105, loss.py:46-51
//
// One workgroup per cell row.  Forward: row log-sum-exp, log-likelihood sum and T = sum_j mu_j dll/dmu_j (the
// softmax backward needs it); backward recomputes the cheap element terms and writes dz and dpi.  This replaces
// relu + softmax + scale + exp + the element-wise likelihood and their autograd (10+ passes over [B][G]) by
// 4 reads forward and 3 reads + 2 writes backward.
#include "common.h"
#include "scpair_hip.h"

template <bool FAST> __device__ __forceinline__ float softplus_(float x) {
  if (!FAST) return softplus_t(x);
  return x > 20.f ? x : __logf(1.f + __expf(x));
}

struct RawTerms {
  float ll, dm, dth, dp, mu, th;
};

// element terms shared by forward and backward (same arithmetic as nbll_fwd/bwd_kernel in elementwise.hip)
// FAST (precision mode 'bf16'): hardware exp / log / reciprocal, as in the fused decoders
// sg: sigmoid(feature_scalar) of the column (1 without a feature factor); th_col: exp(theta parameter) of the column
// when use_col (dispersion 'feature')
template <bool WANT_LL, bool WANT_GRAD, bool FAST, bool VAE>
__device__ __forceinline__ RawTerms raw_terms(float zv, float tzv, float xv, float L, float lse, bool zi, float pv,
                                              float sg = 1.f, float th_col = 0.f, bool use_col = false) {
  const float eps = 1e-8f;
  RawTerms t;
  float a = VAE ? zv : fmaxf(zv, 0.f);
  float scale = exp_<FAST>((VAE ? a : a * sg) - lse);
  t.mu = L * scale;
  t.th = VAE ? softplus_<FAST>(tzv) + 1e-8f : (use_col ? th_col : exp_<FAST>(a));
  float m = t.mu, th = t.th;
  float s = th + m + eps;
  float lt = log_<FAST>(th + eps), ls = log_<FAST>(s);
  bool nz = xv > eps, zero = xv < eps;
  t.ll = 0.f; t.dm = 0.f; t.dth = 0.f; t.dp = 0.f;
  float base = th * (lt - ls);
  if (WANT_LL) {
    float c1 = base + xv * (log_<FAST>(m + eps) - ls) + (nz ? lgam3<FAST>(xv, th) : 0.f);
    if (!zi) {
      t.ll = c1;
    } else {
      float sp = softplus_<FAST>(-pv);
      float tt = -pv + base;
      float c0 = softplus_<FAST>(tt) - sp;
      t.ll = (zero ? c0 : 0.f) + (nz ? (c1 - pv - sp) : 0.f);
    }
  }
  if (WANT_GRAD) {
    float is = div_<FAST>(1.f, s);
    float dt_dmu = -th * is;
    float dt_dth = lt - ls + div_<FAST>(th, th + eps) - th * is;
    float dig = nz ? digam2<FAST>(xv, th) : 0.f;
    float dm_nz = dt_dmu + xv * (div_<FAST>(1.f, m + eps) - is);
    float dth_nz = dt_dth - xv * is + dig;
    if (!zi) {
      t.dm = dm_nz;
      t.dth = dth_nz;
    } else {
      float tt = -pv + base;
      float sig_t = tt > 20.f ? 1.f : sigm_<FAST>(tt);
      float sig_npi = (-pv) > 20.f ? 1.f : sigm_<FAST>(-pv);
      if (zero) {
        t.dm = sig_t * dt_dmu;
        t.dth = sig_t * dt_dth;
        t.dp = -sig_t + sig_npi;
      } else if (nz) {
        t.dm = dm_nz;
        t.dth = dth_nz;
        t.dp = sig_npi - 1.f;
      }
    }
  }
  return t;
}

template <bool FAST, bool VAE>
__global__ __launch_bounds__(256) void raw_rows_fwd_kernel(const float* __restrict__ z, long ldz,
                                                           const float* __restrict__ tz, long ldt,
                                                           const float* __restrict__ pi, long ldp,
                                                           const float* __restrict__ y, long ldy,
                                                           const float* __restrict__ Lrow, int B, int G,
                                                           float* __restrict__ lse_out, float* __restrict__ T_out,
                                                           double* __restrict__ part, const float* __restrict__ fs = nullptr,
                                                           const float* __restrict__ thp = nullptr) {
  __shared__ float red[16];
  __shared__ double dred[8];
  __shared__ float bc[2];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* zr = z + (long)b * ldz;
  // pass 1: log-sum-exp of relu(z) (max first, then the sum)
  float m = VAE ? -3.0e38f : 0.f;                  // predictor: relu(z) [* sigmoid(fs)] >= 0
  if (!VAE && fs) {
    for (int j = tid; j < G; j += 256) m = fmaxf(m, zr[j] * sigm_<FAST>(fs[j]));
  } else {
    for (int j = tid; j < G; j += 256) m = fmaxf(m, zr[j]);
  }
  m = wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float se = 0.f;
  if (!VAE && fs) {
    for (int j = tid; j < G; j += 256) se += exp_<FAST>(fmaxf(zr[j], 0.f) * sigm_<FAST>(fs[j]) - m);
  } else {
    for (int j = tid; j < G; j += 256) se += exp_<FAST>((VAE ? zr[j] : fmaxf(zr[j], 0.f)) - m);
  }
  se = block_sum(se, red);
  if (tid == 0) bc[0] = m + logf(se);
  __syncthreads();
  const float lse = bc[0], L = Lrow[b];
  // pass 2: log-likelihood and T
  const bool zi = pi != nullptr;
  double ll = 0.0;
  float T = 0.f;
  for (int j = tid; j < G; j += 256) {
    float pv = zi ? pi[(long)b * ldp + j] : 0.f;
    float tzv = VAE ? tz[(long)b * ldt + j] : 0.f;
    const float sg = (!VAE && fs) ? sigm_<FAST>(fs[j]) : 1.f;
    const float thc = (!VAE && thp) ? exp_<FAST>(thp[j]) : 0.f;
    RawTerms t = raw_terms<true, true, FAST, VAE>(zr[j], tzv, y[(long)b * ldy + j], L, lse, zi, pv, sg, thc, !VAE && thp);
    ll += (double)t.ll;
    T += t.mu * t.dm;
  }
  T = block_sum(T, red);
  double lls = block_sum_d(ll, dred);
  if (tid == 0) {
    lse_out[b] = lse;
    T_out[b] = T;
    part[b] = lls;
  }
}

__global__ void raw_rows_finish_kernel(const double* __restrict__ part, int B, double n_total, float* __restrict__ loss) {
  __shared__ double dred[8];
  double s = 0.0;
  for (int i = threadIdx.x; i < B; i += 256) s += part[i];
  s = block_sum_d(s, dred);
  if (threadIdx.x == 0) loss[0] = (float)(-s / n_total);
}

template <bool FAST, bool VAE>
__global__ __launch_bounds__(256) void raw_rows_bwd_kernel(const float* __restrict__ z, long ldz,
                                                           const float* __restrict__ tz, long ldt,
                                                           const float* __restrict__ pi, long ldp,
                                                           const float* __restrict__ y, long ldy,
                                                           const float* __restrict__ Lrow,
                                                           const float* __restrict__ lse_in,
                                                           const float* __restrict__ T_in,
                                                           const float* __restrict__ gout, int B, int G,
                                                           float* __restrict__ dz, long lddz, float* __restrict__ dtz,
                                                           long lddt, float* __restrict__ dpi, long lddp,
                                                           float* __restrict__ dL, const float* __restrict__ fs = nullptr,
                                                           const float* __restrict__ thp = nullptr) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const float c = -(gout ? gout[0] : 1.f) / ((float)B * (float)G);
  const float lse = lse_in[b], L = Lrow[b], T = T_in[b];
  const bool zi = pi != nullptr;
  for (int j = tid; j < G; j += 256) {
    float zv = z[(long)b * ldz + j];
    float pv = zi ? pi[(long)b * ldp + j] : 0.f;
    float tzv = VAE ? tz[(long)b * ldt + j] : 0.f;
    const float sg = (!VAE && fs) ? sigm_<FAST>(fs[j]) : 1.f;
    const float thc = (!VAE && thp) ? exp_<FAST>(thp[j]) : 0.f;
    RawTerms t = raw_terms<false, true, FAST, VAE>(zv, tzv, y[(long)b * ldy + j], L, lse, zi, pv, sg, thc, !VAE && thp);
    float scale = t.mu / L;
    if (VAE) {
      // softmax backward (g_j = L dm_j): mu dm - scale T ; theta = softplus(tz) + 1e-8: dth * sigmoid(tz)
      dz[(long)b * lddz + j] = c * (t.mu * t.dm - scale * T);
      dtz[(long)b * lddt + j] = c * t.dth * (tzv > 20.f ? 1.f : sigm_<FAST>(tzv));
    } else {
      // theta = exp(a) joins the mean path (not with dispersion 'feature'); the feature factor scales the softmax
      // input; relu mask
      float da = c * (t.mu * t.dm - scale * T);
      if (fs) da *= sg;
      if (!thp) da += c * t.dth * t.th;
      dz[(long)b * lddz + j] = zv > 0.f ? da : 0.f;
    }
    if (zi) dpi[(long)b * lddp + j] = c * t.dp;
  }
  if (dL && tid == 0) dL[b] = c * T / L;
}

// Gradients of the per-feature parameters: thread = column j, workgroup row = a chunk of RAW_COL_ROWS cells.
//   dfs[j]  = sum_b c (mu dm - scale T) relu(z) sg (1 - sg)          dthp[j] = sum_b c dth theta
#define RAW_COL_ROWS 16
template <bool FAST>
__global__ __launch_bounds__(256) void raw_cols_bwd_kernel(const float* __restrict__ z, long ldz, const float* __restrict__ pi,
                                                           long ldp, const float* __restrict__ y, long ldy,
                                                           const float* __restrict__ Lrow, const float* __restrict__ lse_in,
                                                           const float* __restrict__ T_in, const float* __restrict__ gout,
                                                           const float* __restrict__ fs, const float* __restrict__ thp, int B,
                                                           int G, float* __restrict__ ws) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= G) return;
  const int b0 = blockIdx.y * RAW_COL_ROWS, b1 = min(B, b0 + RAW_COL_ROWS);
  const float c = -(gout ? gout[0] : 1.f) / ((float)B * (float)G);
  const float sg = fs ? sigm_<FAST>(fs[j]) : 1.f;
  const float thc = thp ? exp_<FAST>(thp[j]) : 0.f;
  const bool zi = pi != nullptr;
  float afs = 0.f, ath = 0.f;
  for (int b = b0; b < b1; b++) {
    const float zv = z[(long)b * ldz + j];
    const float pv = zi ? pi[(long)b * ldp + j] : 0.f;
    const float L = Lrow[b];
    RawTerms t = raw_terms<false, true, FAST, false>(zv, 0.f, y[(long)b * ldy + j], L, lse_in[b], zi, pv, sg, thc, thp != nullptr);
    const float scale = t.mu / L;
    afs += c * (t.mu * t.dm - scale * T_in[b]) * fmaxf(zv, 0.f);
    ath += c * t.dth * t.th;
  }
  const long nchunk = gridDim.y;
  ws[((long)blockIdx.y) * G + j] = afs * sg * (1.f - sg);
  ws[(nchunk + blockIdx.y) * G + j] = ath;
}

__global__ __launch_bounds__(256) void raw_cols_reduce_kernel(const float* __restrict__ ws, int nchunk, int G,
                                                              float* __restrict__ dfs, float* __restrict__ dthp) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= G) return;
  float a = 0.f, t = 0.f;
  for (int r = 0; r < nchunk; r++) {
    a += ws[(long)r * G + j];
    t += ws[((long)nchunk + r) * G + j];
  }
  if (dfs) dfs[j] = a;
  if (dthp) dthp[j] = t;
}

extern "C" long scp_raw_rows_feat_ws_floats(int B, int G) {
  if (B <= 0 || G <= 0) return 0;
  return 2L * ((B + RAW_COL_ROWS - 1) / RAW_COL_ROWS) * G;
}

extern "C" int scp_raw_rows_feat_fwd(const float* z, long ldz, const float* pi, long ldp, const float* y, long ldy,
                                     const float* Lrow, const float* fs, const float* thp, int B, int G, int fast, float* lse,
                                     float* T, double* part, float* loss, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SCP_CHECK(B > 0 && G > 0 && z && y && Lrow && lse && T && part && loss, SCP_ERR_ARG, "scp_raw_rows_feat_fwd: bad argument");
  if (fast)
    hipLaunchKernelGGL((raw_rows_fwd_kernel<true, false>), dim3(B), dim3(256), 0, stream, z, ldz, nullptr, 0, pi, ldp, y, ldy, Lrow,
                       B, G, lse, T, part, fs, thp);
  else
    hipLaunchKernelGGL((raw_rows_fwd_kernel<false, false>), dim3(B), dim3(256), 0, stream, z, ldz, nullptr, 0, pi, ldp, y, ldy, Lrow,
                       B, G, lse, T, part, fs, thp);
  SCP_LAUNCH_CHECK();
  hipLaunchKernelGGL(raw_rows_finish_kernel, dim3(1), dim3(256), 0, stream, part, B, (double)B * (double)G, loss);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

extern "C" int scp_raw_rows_feat_bwd(const float* z, long ldz, const float* pi, long ldp, const float* y, long ldy,
                                     const float* Lrow, const float* fs, const float* thp, const float* lse, const float* T,
                                     const float* gout, int B, int G, int fast, float* dz, long lddz, float* dpi, long lddp,
                                     float* dL, float* dfs, float* dthp, float* ws, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SCP_CHECK(B > 0 && G > 0 && z && y && Lrow && lse && T && dz, SCP_ERR_ARG, "scp_raw_rows_feat_bwd: bad argument");
  SCP_CHECK((pi == nullptr) == (dpi == nullptr), SCP_ERR_ARG, "scp_raw_rows_feat_bwd: pi and dpi go together");
  SCP_CHECK((!dfs || fs) && (!dthp || thp), SCP_ERR_ARG, "scp_raw_rows_feat_bwd: a gradient is asked for an absent parameter");
  SCP_CHECK(!(dfs || dthp) || ws, SCP_ERR_ARG, "scp_raw_rows_feat_bwd: parameter gradients need the workspace");
  if (fast)
    hipLaunchKernelGGL((raw_rows_bwd_kernel<true, false>), dim3(B), dim3(256), 0, stream, z, ldz, nullptr, 0, pi, ldp, y, ldy, Lrow,
                       lse, T, gout, B, G, dz, lddz, nullptr, 0, dpi, lddp, dL, fs, thp);
  else
    hipLaunchKernelGGL((raw_rows_bwd_kernel<false, false>), dim3(B), dim3(256), 0, stream, z, ldz, nullptr, 0, pi, ldp, y, ldy, Lrow,
                       lse, T, gout, B, G, dz, lddz, nullptr, 0, dpi, lddp, dL, fs, thp);
  SCP_LAUNCH_CHECK();
  if (dfs || dthp) {
    const int nchunk = (B + RAW_COL_ROWS - 1) / RAW_COL_ROWS;
    dim3 grid((unsigned)((G + 255) / 256), (unsigned)nchunk);
    if (fast)
      hipLaunchKernelGGL(raw_cols_bwd_kernel<true>, grid, dim3(256), 0, stream, z, ldz, pi, ldp, y, ldy, Lrow, lse, T, gout, fs, thp,
                         B, G, ws);
    else
      hipLaunchKernelGGL(raw_cols_bwd_kernel<false>, grid, dim3(256), 0, stream, z, ldz, pi, ldp, y, ldy, Lrow, lse, T, gout, fs, thp,
                         B, G, ws);
    SCP_LAUNCH_CHECK();
    hipLaunchKernelGGL(raw_cols_reduce_kernel, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, stream, ws, nchunk, G, dfs, dthp);
    SCP_LAUNCH_CHECK();
  }
  return SCP_OK;
}

extern "C" int scp_raw_rows_fwd(const float* z, long ldz, const float* tz, long ldt, const float* pi, long ldp,
                                const float* y, long ldy, const float* Lrow, int B, int G, int fast, float* lse, float* T,
                                double* part, float* loss, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SCP_CHECK(B > 0 && G > 0 && z && y && Lrow && lse && T && part && loss, SCP_ERR_ARG, "scp_raw_rows_fwd: bad argument");
#define RR_FWD(FA, VA)                                                                                             \
  hipLaunchKernelGGL((raw_rows_fwd_kernel<FA, VA>), dim3(B), dim3(256), 0, stream, z, ldz, tz, ldt, pi, ldp, y, ldy, \
                     Lrow, B, G, lse, T, part)
  if (tz) { if (fast) RR_FWD(true, true); else RR_FWD(false, true); }
  else { if (fast) RR_FWD(true, false); else RR_FWD(false, false); }
#undef RR_FWD
  SCP_LAUNCH_CHECK();
  hipLaunchKernelGGL(raw_rows_finish_kernel, dim3(1), dim3(256), 0, stream, part, B, (double)B * (double)G, loss);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

extern "C" int scp_raw_rows_bwd(const float* z, long ldz, const float* tz, long ldt, const float* pi, long ldp,
                                const float* y, long ldy, const float* Lrow, const float* lse, const float* T,
                                const float* gout, int B, int G, int fast, float* dz, long lddz, float* dtz, long lddt,
                                float* dpi, long lddp, float* dL, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SCP_CHECK(B > 0 && G > 0 && z && y && Lrow && lse && T && dz, SCP_ERR_ARG, "scp_raw_rows_bwd: bad argument");
  SCP_CHECK((pi == nullptr) == (dpi == nullptr), SCP_ERR_ARG, "scp_raw_rows_bwd: pi and dpi go together");
  SCP_CHECK((tz == nullptr) == (dtz == nullptr), SCP_ERR_ARG, "scp_raw_rows_bwd: tz and dtz go together");
#define RR_BWD(FA, VA)                                                                                             \
  hipLaunchKernelGGL((raw_rows_bwd_kernel<FA, VA>), dim3(B), dim3(256), 0, stream, z, ldz, tz, ldt, pi, ldp, y, ldy, \
                     Lrow, lse, T, gout, B, G, dz, lddz, dtz, lddt, dpi, lddp, dL)
  if (tz) { if (fast) RR_BWD(true, true); else RR_BWD(false, true); }
  else { if (fast) RR_BWD(true, false); else RR_BWD(false, false); }
#undef RR_BWD
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}
