// Host API of the fused decoder kernels (kernels: decoder_kernel.h, instantiated per precision in
// decoder_{ber,nb}_ns{1,3}.hip).
#include <cstdlib>
#include "decoder_kernel.h"

int dec_run_ber_ns1(int mode, const DecArgs& g, int kp, int grid, hipStream_t stream);
int dec_run_ber_ns3(int mode, const DecArgs& g, int kp, int grid, hipStream_t stream);
int dec_run_nb_ns1(int mode, const DecArgs& g, int kp, int grid, hipStream_t stream);
int dec_run_nb_ns3(int mode, const DecArgs& g, int kp, int grid, hipStream_t stream);

template <int MODE>
static int dec_dispatch(const DecArgs& g, int ns, int grid, hipStream_t stream) {
  SCP_CHECK(g.K <= 64, SCP_ERR_SHAPE, "decoder: latent+covariate width K=%d > 64 unsupported", g.K);
  int kp = g.K <= 16 ? 16 : (g.K <= 48 ? 48 : 64);
  SCP_CHECK(!(ns == 3 && kp == 64), SCP_ERR_SHAPE, "decoder: K=%d > 48 needs ns = 1 (LDS budget)", g.K);
  SCP_CHECK(ns == 1 || ns == 3, SCP_ERR_ARG, "decoder: ns must be 1 or 3 (got %d)", ns);
  constexpr bool ber = (MODE == MODE_BER_TRAIN || MODE == MODE_BER_EVAL);
  if (ber) return ns == 1 ? dec_run_ber_ns1(MODE, g, kp, grid, stream) : dec_run_ber_ns3(MODE, g, kp, grid, stream);
  return ns == 1 ? dec_run_nb_ns1(MODE, g, kp, grid, stream) : dec_run_nb_ns3(MODE, g, kp, grid, stream);
}

// ---- finishing kernels ------------------------------------------------------------------------------
// Workgroup = 64 outputs x 16 chunk groups (dim3(64,16)): group y folds chunks y, y+16, ...; the 16 group
// results are then combined in ascending order, so every output is a fixed-order (deterministic) reduction.
// dh[b][k] = sum_chunk slab[chunk][b][k]
__global__ __launch_bounds__(1024) void dec_finish_dh_kernel(const float* __restrict__ slab, int nchunk, int B, int K,
                                                             int KP, float* __restrict__ dh, long lddh) {
  __shared__ float red[16][64];
  int x = threadIdx.x, y = threadIdx.y;
  long j = (long)blockIdx.x * 64 + x;   // b * KP + k
  bool ok = j < (long)B * KP;
  float s = ok ? strided_sum16(slab + j, (long)B * KP, nchunk, y) : 0.f;
  red[y][x] = s;
  __syncthreads();
  if (y == 0 && ok) {
    int b = (int)(j / KP), k = (int)(j % KP);
    if (k < K) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 16; q++) t += red[q][x];
      dh[(long)b * lddh + k] = t;
    }
  }
}

// row_mode 0: out0[b] = sum_chunk slab[.][b][0]                    (Ber dl / NB T)
// row_mode 1: (m,s) merge -> out0[b] = m + log s                    (NB log-sum-exp)
// row_mode 2: like 0, and out1[b] = out0[b] * dscale(lat[b])        (NB dlatent_lib = T * lrelu'(lat))
__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
  float mn = fmaxf(m, m2);
  float e1 = (m == -INFINITY) ? 0.f : expf(m - mn), e2 = (m2 == -INFINITY) ? 0.f : expf(m2 - mn);
  s = s * e1 + s2 * e2;
  m = mn;
}
__global__ __launch_bounds__(1024) void dec_finish_rows_kernel(const float* __restrict__ slab, int nchunk, int B,
                                                               int row_mode, float scale_, const float* __restrict__ aux,
                                                               float* __restrict__ out0, float* __restrict__ out1) {
  __shared__ float red[2][16][64];
  int x = threadIdx.x, y = threadIdx.y;
  int b = blockIdx.x * 64 + x;
  bool ok = b < B;
  if (row_mode == 1) {
    float m = -INFINITY, s = 0.f;
    if (ok) {
      int c = y;
      for (; c + 16 < nchunk; c += 32) {
        float2 p0 = *(const float2*)&slab[((long)c * B + b) * 2];
        float2 p1 = *(const float2*)&slab[((long)(c + 16) * B + b) * 2];
        lse_merge(m, s, p0.x, p0.y);
        lse_merge(m, s, p1.x, p1.y);
      }
      for (; c < nchunk; c += 16) {
        float2 p0 = *(const float2*)&slab[((long)c * B + b) * 2];
        lse_merge(m, s, p0.x, p0.y);
      }
    }
    red[0][y][x] = m;
    red[1][y][x] = s;
    __syncthreads();
    if (y == 0 && ok) {
      float mm = -INFINITY, ss = 0.f;
      for (int q = 0; q < 16; q++) lse_merge(mm, ss, red[0][q][x], red[1][q][x]);
      out0[b] = mm + logf(ss);
    }
  } else {
    float s = ok ? strided_sum16(slab + (long)b * 2, (long)B * 2, nchunk, y) : 0.f;
    red[0][y][x] = s;
    __syncthreads();
    if (y == 0 && ok) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 16; q++) t += red[0][q][x];
      out0[b] = t;
      if (row_mode == 2 && out1) out1[b] = t * scale_ * ((aux[b] > 0.f) ? 1.f : 0.01f);
    }
  }
}

__global__ void dec_finish_loss_kernel(const double* __restrict__ sum, double n_total, const int32_t* __restrict__ nnan,
                                       float* __restrict__ loss) {
  double n = n_total - (nnan ? (double)nnan[0] : 0.0);
  loss[0] = (float)(sum[0] / n);
}

int scp_reduce_partials(const double* part, int n, int ncomp, double* out, hipStream_t stream);


extern "C" int scp_dec_chunks(int F) { return (F + DEC_NPT * FT - 1) / (DEC_NPT * FT); }
extern "C" int scp_dec_kp(int K) { return K <= 16 ? 16 : (K <= 48 ? 48 : 64); }

// cell blocks per workgroup: enough (chunk x cell-group) workgroups to fill the chip a few times over
static int dec_cpg(int B, int F) {
  int nchunk = scp_dec_chunks(F), ncb = (B + CB - 1) / CB;
  static int target = 0;
  if (!target) {
    const char* e = getenv("SCPAIR_DEC_WGS");
    target = e ? atoi(e) : 2048;
    if (target < 1) target = 2048;
  }
  int ncg = (target + nchunk - 1) / nchunk;
  if (ncg > ncb) ncg = ncb;
  if (ncg < 1) ncg = 1;
  return (ncb + ncg - 1) / ncg;
}
static int dec_ncg(int B, int F) {
  int ncb = (B + CB - 1) / CB, cpg = dec_cpg(B, F);
  return (ncb + cpg - 1) / cpg;
}

static int check_common(const char* who, int B, int F, int K, const float* h, const float* W, const float* bias,
                        const void* ws) {
  SCP_CHECK(B > 0 && F > 0 && K > 0, SCP_ERR_ARG, "%s: sizes must be positive (B=%d F=%d K=%d)", who, B, F, K);
  SCP_CHECK(h && W && bias && ws, SCP_ERR_ARG, "%s: null pointer", who);
  return SCP_OK;
}

// Workspace layout, nchunk = scp_dec_chunks(F), KP = scp_dec_kp(K), ncg = cell groups:
//   ws_d : doubles [nchunk*ncg + 8]
//   ws_f : floats  [nchunk*B*KP] dh slabs | [nchunk*B*2] row slabs | [ncg*F*K] dW slabs | [3*ncg*F] db/dfs/dtheta slabs
extern "C" long scp_dec_ws_floats(int B, int F, int K) {
  long nchunk = scp_dec_chunks(F), ncg = dec_ncg(B, F);
  return nchunk * (long)B * scp_dec_kp(K) + nchunk * (long)B * 2 + ncg * (long)F * K + 3 * ncg * (long)F;
}
extern "C" long scp_dec_ws_doubles(int B, int F) { return (long)scp_dec_chunks(F) * dec_ncg(B, F) + 8; }

struct DecPlan {
  int nchunk, KP, cpg, ncg;
  float *dh_slab, *row_slab, *dW_slab, *db_slab, *dfs_slab, *dth_slab;
};

static DecPlan make_plan(DecArgs& g, float* ws_f, float* dW, long lddw, float* db, float* dfs, float* dtheta) {
  DecPlan p;
  p.nchunk = scp_dec_chunks(g.F); p.KP = scp_dec_kp(g.K); p.cpg = dec_cpg(g.B, g.F); p.ncg = dec_ncg(g.B, g.F);
  p.dh_slab = ws_f;
  p.row_slab = ws_f ? p.dh_slab + (long)p.nchunk * g.B * p.KP : nullptr;
  p.dW_slab = ws_f ? p.row_slab + (long)p.nchunk * g.B * 2 : nullptr;
  p.db_slab = ws_f ? p.dW_slab + (long)p.ncg * g.F * g.K : nullptr;
  p.dfs_slab = ws_f ? p.db_slab + (long)p.ncg * g.F : nullptr;
  p.dth_slab = ws_f ? p.dfs_slab + (long)p.ncg * g.F : nullptr;
  g.cpg = p.cpg;
  g.dh_slab = p.dh_slab; g.row_slab = p.row_slab;
  if (p.ncg == 1) {
    g.dW = dW; g.lddw = lddw; g.cg_stride_w = 0; g.db = db; g.dfs = dfs; g.dtheta = dtheta;
  } else {
    g.dW = p.dW_slab; g.lddw = g.K; g.cg_stride_w = (long)g.F * g.K;
    g.db = p.db_slab; g.dfs = dfs ? p.dfs_slab : nullptr; g.dtheta = dtheta ? p.dth_slab : nullptr;
  }
  return p;
}

// sum the per-cell-group slabs of the per-feature gradients (fixed order -> deterministic)
static int finish_feature_grads(const DecPlan& p, const DecArgs& g, float* dW, long lddw, float* db, float* dfs,
                                float* dtheta, hipStream_t stream) {
  if (p.ncg == 1) return SCP_OK;
  int rc = scp_slab_reduce(p.dW_slab, (long)g.F * g.K, p.ncg, g.F, g.K, g.K, nullptr, dW, lddw, stream);
  if (rc) return rc;
  rc = scp_slab_reduce(p.db_slab, g.F, p.ncg, 1, g.F, g.F, nullptr, db, g.F, stream);
  if (rc) return rc;
  if (dfs) {
    rc = scp_slab_reduce(p.dfs_slab, g.F, p.ncg, 1, g.F, g.F, nullptr, dfs, g.F, stream);
    if (rc) return rc;
  }
  if (dtheta) {
    rc = scp_slab_reduce(p.dth_slab, g.F, p.ncg, 1, g.F, g.F, nullptr, dtheta, g.F, stream);
    if (rc) return rc;
  }
  return SCP_OK;
}

extern "C" int scp_ber_dec_bce(int train, int B, int F, int K, const float* h, long ldh, const float* W, long ldw,
                               const float* bias, const float* fs, const float* lrow, int lrow_n, const int32_t* t_ptr,
                               const int32_t* t_key, const float* t_val, const int32_t* nnan, const float* gout,
                               float* loss, float* dW, long lddw, float* db, float* dfs, float* dh, long lddh,
                               float* dlrow, float* ws_f, double* ws_d, int ns, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_common("scp_ber_dec_bce", B, F, K, h, W, bias, ws_d);
  if (rc) return rc;
  SCP_CHECK(t_ptr && t_key && loss && ws_f, SCP_ERR_ARG, "scp_ber_dec_bce: null pointer");
  SCP_CHECK(!train || (dW && db && dh), SCP_ERR_ARG, "scp_ber_dec_bce: training needs dW, db, dh");
  SCP_CHECK(lrow_n == 0 || lrow_n == 1 || lrow_n == B, SCP_ERR_ARG, "scp_ber_dec_bce: lrow_n must be 0, 1 or B");
  DecArgs g = {};
  g.B = B; g.F = F; g.K = K; g.h = h; g.ldh = ldh; g.W = W; g.ldw = ldw; g.bias = bias; g.fs = fs;
  g.lrow = lrow; g.lrow_n = lrow_n; g.t_ptr = t_ptr; g.t_key = t_key; g.t_val = t_val; g.nnan = nnan; g.gout = gout;
  g.loss_part = ws_d;
  DecPlan p = make_plan(g, ws_f, dW, lddw, db, fs ? dfs : nullptr, nullptr);
  int nblk = p.nchunk * p.ncg;
  rc = train ? dec_dispatch<MODE_BER_TRAIN>(g, ns, p.nchunk, stream) : dec_dispatch<MODE_BER_EVAL>(g, ns, p.nchunk, stream);
  if (rc) return rc;
  rc = scp_reduce_partials(ws_d, nblk, 1, ws_d + nblk, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(dec_finish_loss_kernel, dim3(1), dim3(1), 0, stream, ws_d + nblk, (double)B * (double)F, nnan, loss);
  SCP_LAUNCH_CHECK();
  if (train) {
    rc = finish_feature_grads(p, g, dW, lddw, db, fs ? dfs : nullptr, nullptr, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(dec_finish_dh_kernel, dim3((unsigned)(((long)B * p.KP + 63) / 64)), dim3(64, 16), 0, stream, p.dh_slab,
                       p.nchunk, B, K, p.KP, dh, lddh);
    SCP_LAUNCH_CHECK();
    if (dlrow && lrow && lrow_n == B) {
      hipLaunchKernelGGL(dec_finish_rows_kernel, dim3((B + 63) / 64), dim3(64, 16), 0, stream, p.row_slab, p.nchunk, B, 0,
                         1.f, (const float*)nullptr, dlrow, (float*)nullptr);
      SCP_LAUNCH_CHECK();
    }
  }
  return SCP_OK;
}

// NB decoder (likelihood 'nb'): loss = -mean log NB(y; L*softmax(a'), theta)
//   lrow = library size L per cell (sample_factor) or exp(leaky_relu(latent_lib)) precomputed by the caller
//   lat  = latent_lib (only to form dlat = -(1/N) T lrelu'(lat)); null when lrow is data
extern "C" int scp_nb_dec_ll(int train, int B, int F, int K, const float* h, long ldh, const float* W, long ldw,
                             const float* bias, const float* fs, const float* theta_param, const float* lrow,
                             const float* lat, const int32_t* t_ptr, const int32_t* t_key, const float* t_val,
                             const float* gout, float* loss, float* dW, long lddw, float* db, float* dfs, float* dtheta,
                             float* dh, long lddh, float* dlat, float* rowbuf /* [2*B] */, float* ws_f, double* ws_d,
                             int ns, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_common("scp_nb_dec_ll", B, F, K, h, W, bias, ws_d);
  if (rc) return rc;
  SCP_CHECK(t_ptr && t_key && loss && lrow && rowbuf && ws_f, SCP_ERR_ARG, "scp_nb_dec_ll: null pointer");
  SCP_CHECK(!train || (dW && db && dh), SCP_ERR_ARG, "scp_nb_dec_ll: training needs dW, db, dh");
  DecArgs g = {};
  g.B = B; g.F = F; g.K = K; g.h = h; g.ldh = ldh; g.W = W; g.ldw = ldw; g.bias = bias; g.fs = fs;
  g.theta_param = theta_param; g.lrow = lrow; g.lrow_n = B; g.t_ptr = t_ptr; g.t_key = t_key; g.t_val = t_val;
  g.gout = gout; g.loss_part = ws_d;
  DecPlan p = make_plan(g, ws_f, dW, lddw, db, fs ? dfs : nullptr, theta_param ? dtheta : nullptr);
  int nblk = p.nchunk * p.ncg;
  float* lse = rowbuf;
  float* T = rowbuf + B;
  // pass 1: row log-sum-exp
  rc = dec_dispatch<MODE_NB_STATS>(g, ns, p.nchunk, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(dec_finish_rows_kernel, dim3((B + 63) / 64), dim3(64, 16), 0, stream, p.row_slab, p.nchunk, B, 1, 1.f,
                     (const float*)nullptr, lse, (float*)nullptr);
  SCP_LAUNCH_CHECK();
  // pass 2: loss and T
  g.rowlse = lse;
  rc = dec_dispatch<MODE_NB_LOSS>(g, ns, p.nchunk, stream);
  if (rc) return rc;
  rc = scp_reduce_partials(ws_d, nblk, 1, ws_d + nblk, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(dec_finish_loss_kernel, dim3(1), dim3(1), 0, stream, ws_d + nblk, (double)B * (double)F,
                     (const int32_t*)nullptr, loss);
  SCP_LAUNCH_CHECK();
  if (!train) return SCP_OK;
  float sc = -1.f / ((float)B * (float)F);
  hipLaunchKernelGGL(dec_finish_rows_kernel, dim3((B + 63) / 64), dim3(64, 16), 0, stream, p.row_slab, p.nchunk, B,
                     (dlat && lat) ? 2 : 0, sc, lat, T, dlat);
  SCP_LAUNCH_CHECK();
  // pass 3: gradients
  g.rowT = T;
  rc = dec_dispatch<MODE_NB_GRAD>(g, ns, p.nchunk, stream);
  if (rc) return rc;
  rc = finish_feature_grads(p, g, dW, lddw, db, fs ? dfs : nullptr, theta_param ? dtheta : nullptr, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(dec_finish_dh_kernel, dim3((unsigned)(((long)B * p.KP + 63) / 64)), dim3(64, 16), 0, stream, p.dh_slab,
                     p.nchunk, B, K, p.KP, dh, lddh);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}
