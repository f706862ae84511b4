// Host API of the fused decoder kernels (kernels: decoder_kernel.h, instantiated per precision in
// decoder_{ber,nb}_ns{1,3}.hip).
#include <cstdlib>
#include "decoder_kernel.h"

int dec_run_ber_ns1(int mode, const DecArgs& g, int kp, int grid, hipStream_t stream);
int dec_run_ber_ns3(int mode, const DecArgs& g, int kp, int grid, hipStream_t stream);
int dec_run_nb_ns1(int mode, const DecArgs& g, int kp, int grid, hipStream_t stream);
int dec_run_nb_ns3(int mode, const DecArgs& g, int kp, int grid, hipStream_t stream);

#ifdef DEC_PROF
static unsigned long long* prof_dev = nullptr;
extern "C" int scp_dec_prof_read(unsigned long long* out, int n) {
  if (!prof_dev) return -1;
  hipDeviceSynchronize();
  return (int)hipMemcpy(out, prof_dev, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost);
}
#endif

template <int MODE>
static int dec_dispatch(const DecArgs& g_, int ns, int grid, hipStream_t stream) {
  DecArgs g = g_;
#ifdef DEC_PROF
  if (!prof_dev) { hipMalloc(&prof_dev, sizeof(unsigned long long) * 12 * 4096); }
  hipMemsetAsync(prof_dev, 0, sizeof(unsigned long long) * 12 * 4096, stream);
  g.prof = prof_dev;
#endif
  SCP_CHECK(g.K <= 64, SCP_ERR_SHAPE, "decoder: latent+covariate width K=%d > 64 unsupported", g.K);
  int kp = g.K <= 16 ? 16 : (g.K <= 48 ? 48 : 64);
  SCP_CHECK(!(ns == 3 && kp == 64), SCP_ERR_SHAPE, "decoder: K=%d > 48 needs ns = 1 (LDS budget)", g.K);
  SCP_CHECK(ns == 1 || ns == 3, SCP_ERR_ARG, "decoder: ns must be 1 or 3 (got %d)", ns);
  constexpr bool ber = (MODE == MODE_BER_TRAIN || MODE == MODE_BER_EVAL);
  if (ber) return ns == 1 ? dec_run_ber_ns1(MODE, g, kp, grid, stream) : dec_run_ber_ns3(MODE, g, kp, grid, stream);
  return ns == 1 ? dec_run_nb_ns1(MODE, g, kp, grid, stream) : dec_run_nb_ns3(MODE, g, kp, grid, stream);
}

// ---- finishing reductions --------------------------------------------------------------------------
// Workgroup = 64 outputs x 16 chunk groups (dim3(64,16)): group y folds chunks y, y+16, ...; the 16 group
// results are then combined in ascending order, so every output is a fixed-order (deterministic) reduction.
// row_mode 0: out0[b] = sum_chunk slab[.][b][0]                    (Ber dl / NB T)
// row_mode 1: (m,s) merge -> out0[b] = m + log s                    (NB log-sum-exp)
// row_mode 2: like 0, and out1[b] = out0[b] * dscale(lat[b])        (NB dlatent_lib = T * lrelu'(lat))
__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
  float mn = fmaxf(m, m2);
  float e1 = (m == -INFINITY) ? 0.f : expf(m - mn), e2 = (m2 == -INFINITY) ? 0.f : expf(m2 - mn);
  s = s * e1 + s2 * e2;
  m = mn;
}

// ---- all finishing reductions of a decoder pass in ONE launch ------------------------------------------------------
// The workgroups (dim3(64, 16)) of the launch are split into ranges, one per job: the dh slabs, a per-cell row
// reduction, up to four cell-group slab sums (dW, db, dfs, dtheta) and the loss.  Each job keeps the fixed summation
// order of the stand-alone kernels above.
struct FinishJobs {
  // dh[b][k] = sum over chunks (blocks [0, n_dh))
  const float* dh_slab; float* dh; long lddh; int K, KP, n_dh;
  // rows (blocks [n_dh, n_dh + n_rows)): row_mode as dec_finish_rows_kernel
  const float* row_slab; int row_mode; float row_scale; const float* row_aux; float* row_out0; float* row_out1; int n_rows;
  // cell-group slab sums: out[r][c] = sum_y slab[y * stride + r * cols + c]  (blocks of 64 x 16 elements each)
  const float* sl_src[4]; float* sl_dst[4]; long sl_stride[4]; long sl_n[4]; int sl_cols[4]; long sl_ldo[4]; int sl_blk0[4];
  int nslab_jobs, ncg;
  // loss (one block, the last): loss = sum(part[0..nblk)) / (n_total - nnan)
  const double* loss_part; int nblk; double n_total; const int32_t* nnan; float* loss; int loss_blk;
  int nchunk, B;
};

__global__ __launch_bounds__(1024) void dec_finish_all_kernel(FinishJobs J) {
  __shared__ float red[2][16][64];
  __shared__ double dred[16];
  const int x = threadIdx.x, y = threadIdx.y;
  const int blk = blockIdx.x;
  if (blk < J.n_dh) {
    const long j = (long)blk * 64 + x;   // b * KP + k
    const bool ok = j < (long)J.B * J.KP;
    const float s = ok ? strided_sum16(J.dh_slab + j, (long)((J.B + CB - 1) / CB * CB) * J.KP, J.nchunk, y) : 0.f;
    red[0][y][x] = s;
    __syncthreads();
    if (y == 0 && ok) {
      const int b = (int)(j / J.KP), k = (int)(j % J.KP);
      if (k < J.K) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; q++) t += red[0][q][x];
        J.dh[(long)b * J.lddh + k] = t;
      }
    }
    return;
  }
  if (blk < J.n_dh + J.n_rows) {
    const int b = (blk - J.n_dh) * 64 + x;
    const bool ok = b < J.B;
    const float* slab = J.row_slab;
    if (J.row_mode == 1) {
      float m = -INFINITY, s = 0.f;
      if (ok) {
        int c = y;
        for (; c + 16 < J.nchunk; c += 32) {
          const float2 p0 = *(const float2*)&slab[((long)c * J.B + b) * 2];
          const float2 p1 = *(const float2*)&slab[((long)(c + 16) * J.B + b) * 2];
          lse_merge(m, s, p0.x, p0.y);
          lse_merge(m, s, p1.x, p1.y);
        }
        for (; c < J.nchunk; c += 16) {
          const float2 p0 = *(const float2*)&slab[((long)c * J.B + b) * 2];
          lse_merge(m, s, p0.x, p0.y);
        }
      }
      red[0][y][x] = m;
      red[1][y][x] = s;
      __syncthreads();
      if (y == 0 && ok) {
        float mm = -INFINITY, ss = 0.f;
        for (int q = 0; q < 16; q++) lse_merge(mm, ss, red[0][q][x], red[1][q][x]);
        J.row_out0[b] = mm + logf(ss);
      }
    } else {
      const float s = ok ? strided_sum16(slab + (long)b * 2, (long)J.B * 2, J.nchunk, y) : 0.f;
      red[0][y][x] = s;
      __syncthreads();
      if (y == 0 && ok) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; q++) t += red[0][q][x];
        J.row_out0[b] = t;
        if (J.row_mode == 2 && J.row_out1) J.row_out1[b] = t * J.row_scale * ((J.row_aux[b] > 0.f) ? 1.f : 0.01f);
      }
    }
    return;
  }
  if (blk == J.loss_blk) {
    // fixed order: thread t adds partials t, t + 1024, ...; waves, then the 16 wave sums, in ascending order
    const int t = y * 64 + x;
    double s = 0.0;
    for (int i = t; i < J.nblk; i += 1024) s += J.loss_part[i];
    s = wave_sum_d(s);
    if (x == 0) dred[y] = s;
    __syncthreads();
    if (t == 0) {
      double tot = 0.0;
      for (int q = 0; q < 16; q++) tot += dred[q];
      const double n = J.n_total - (J.nnan ? (double)J.nnan[0] : 0.0);
      J.loss[0] = (float)(tot / n);
    }
    return;
  }
  for (int q = J.nslab_jobs - 1; q >= 0; q--) {
    if (blk >= J.sl_blk0[q]) {
      const long i = ((long)(blk - J.sl_blk0[q]) * 16 + y) * 64 + x;
      if (i < J.sl_n[q]) {
        float sacc = 0.f;
        for (int g_ = 0; g_ < J.ncg; g_++) sacc += J.sl_src[q][(long)g_ * J.sl_stride[q] + i];
        const long r = i / J.sl_cols[q], c = i - r * J.sl_cols[q];
        J.sl_dst[q][r * J.sl_ldo[q] + c] = sacc;
      }
      return;
    }
  }
}

// Lays the jobs out over the block index space and launches; slab jobs only when ncg > 1.
static int launch_finish(FinishJobs& J, hipStream_t stream) {
  int nblocks = J.n_dh + J.n_rows;
  for (int q = 0; q < J.nslab_jobs; q++) {
    J.sl_blk0[q] = nblocks;
    nblocks += (int)((J.sl_n[q] + 1023) / 1024);
  }
  J.loss_blk = J.loss ? nblocks++ : -1;
  if (nblocks == 0) return SCP_OK;
  hipLaunchKernelGGL(dec_finish_all_kernel, dim3(nblocks), dim3(64, 16), 0, stream, J);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}


static int dec_chunks_npt(int F, int npt) { return (F + npt * FT - 1) / (npt * FT); }
extern "C" int scp_dec_chunks(int F) { return dec_chunks_npt(F, DEC_NPT); }
extern "C" int scp_dec_kp(int K) { return K <= 16 ? 16 : (K <= 48 ? 48 : 64); }

// Launch geometry of a pass: chunks of `npt` feature tiles x cell groups of `cpg` cell blocks -- enough workgroups
// to fill the chip a few times over
struct PassGeom { int nchunk, cpg, ncg; };
static PassGeom pass_geom(int B, int F, int npt) {
  PassGeom q;
  q.nchunk = dec_chunks_npt(F, npt);
  const int ncb = (B + CB - 1) / CB;
  static int target = 0;
  if (!target) {
    const char* e = getenv("SCPAIR_DEC_WGS");
    target = e ? atoi(e) : 2048;
    if (target < 1) target = 2048;
  }
  int ncg = (target + q.nchunk - 1) / q.nchunk;
  if (ncg > ncb) ncg = ncb;
  if (ncg < 1) ncg = 1;
  q.cpg = (ncb + ncg - 1) / ncg;
  {
    const char* e = getenv(npt == DEC_NPT ? "SCPAIR_DEC_CPG_TRAIN" : "SCPAIR_DEC_CPG_EVAL");   // experiments
    if (e && atoi(e) > 0) q.cpg = atoi(e) < ncb ? atoi(e) : ncb;
  }
  q.ncg = (ncb + q.cpg - 1) / q.cpg;
  return q;
}
static int dec_ncg(int B, int F) { return pass_geom(B, F, DEC_NPT).ncg; }

static int check_common(const char* who, int B, int F, int K, const float* h, const float* W, const float* bias,
                        const void* ws) {
  SCP_CHECK(B > 0 && F > 0 && K > 0, SCP_ERR_ARG, "%s: sizes must be positive (B=%d F=%d K=%d)", who, B, F, K);
  SCP_CHECK(h && W && bias && ws, SCP_ERR_ARG, "%s: null pointer", who);
  return SCP_OK;
}

// Workspace layout, nchunk = scp_dec_chunks(F), KP = scp_dec_kp(K), ncg = cell groups:
//   ws_d : doubles [nchunk*ncg + 8]
//   ws_f : floats  [nchunk*Bp*KP] dh slabs (Bp = B rounded up to whole 64-cell blocks) | [nchunk*B*2] row slabs | [ncg*F*K] dW slabs | [3*ncg*F] db/dfs/dtheta slabs
extern "C" long scp_dec_ws_floats(int B, int F, int K) {
  long nchunk = scp_dec_chunks(F), ncg = dec_ncg(B, F);
  const long Bp = (B + CB - 1) / CB * CB;
  return nchunk * Bp * scp_dec_kp(K) + nchunk * (long)B * 2 + ncg * (long)F * K + 3 * ncg * (long)F;
}
extern "C" long scp_dec_ws_doubles(int B, int F) {
  const PassGeom t = pass_geom(B, F, DEC_NPT), e = pass_geom(B, F, DEC_NPT_EVAL);
  const long nt = (long)t.nchunk * t.ncg, ne = (long)e.nchunk * e.ncg;
  return (nt > ne ? nt : ne) + 8;
}

struct DecPlan {
  int nchunk, KP, cpg, ncg;     // training passes
  PassGeom ev;                  // passes without gradients
  float *dh_slab, *row_slab, *dW_slab, *db_slab, *dfs_slab, *dth_slab;
};

static DecPlan make_plan(DecArgs& g, float* ws_f, float* dW, long lddw, float* db, float* dfs, float* dtheta) {
  DecPlan p;
  const PassGeom tg = pass_geom(g.B, g.F, DEC_NPT);
  p.nchunk = tg.nchunk; p.KP = scp_dec_kp(g.K); p.cpg = tg.cpg; p.ncg = tg.ncg;
  p.ev = pass_geom(g.B, g.F, DEC_NPT_EVAL);
  p.dh_slab = ws_f;
  p.row_slab = ws_f ? p.dh_slab + (long)p.nchunk * ((g.B + CB - 1) / CB * CB) * p.KP : nullptr;
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

// the cell-group slab sums of the per-feature gradients as jobs of the finishing launch (none when ncg == 1)
static void add_feature_jobs(FinishJobs& J, const DecPlan& p, const DecArgs& g, float* dW, long lddw, float* db, float* dfs,
                             float* dtheta) {
  J.ncg = p.ncg;
  J.nslab_jobs = 0;
  if (p.ncg == 1) return;
  auto add = [&](const float* src, long stride, long n, int cols, float* dst, long ldo) {
    const int q = J.nslab_jobs++;
    J.sl_src[q] = src; J.sl_stride[q] = stride; J.sl_n[q] = n; J.sl_cols[q] = cols; J.sl_dst[q] = dst; J.sl_ldo[q] = ldo;
  };
  add(p.dW_slab, (long)g.F * g.K, (long)g.F * g.K, g.K, dW, lddw);
  add(p.db_slab, g.F, g.F, g.F, db, g.F);
  if (dfs) add(p.dfs_slab, g.F, g.F, g.F, dfs, g.F);
  if (dtheta) add(p.dth_slab, g.F, g.F, g.F, dtheta, g.F);
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
  const int nchunk = train ? p.nchunk : p.ev.nchunk;
  const int nblk = nchunk * (train ? p.ncg : p.ev.ncg);
  if (!train) g.cpg = p.ev.cpg;
  rc = train ? dec_dispatch<MODE_BER_TRAIN>(g, ns, nchunk, stream) : dec_dispatch<MODE_BER_EVAL>(g, ns, nchunk, stream);
  if (rc) return rc;
  FinishJobs J = {};
  J.nchunk = nchunk; J.B = B;
  J.loss_part = ws_d; J.nblk = nblk; J.n_total = (double)B * (double)F; J.nnan = nnan; J.loss = loss;
  if (train) {
    add_feature_jobs(J, p, g, dW, lddw, db, fs ? dfs : nullptr, nullptr);
    J.dh_slab = p.dh_slab; J.dh = dh; J.lddh = lddh; J.K = K; J.KP = p.KP; J.n_dh = (int)(((long)B * p.KP + 63) / 64);
    if (dlrow && lrow && lrow_n == B) {
      J.row_slab = p.row_slab; J.row_mode = 0; J.row_scale = 1.f; J.row_out0 = dlrow; J.n_rows = (B + 63) / 64;
    }
  }
  return launch_finish(J, stream);
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
  const int nblk = p.ev.nchunk * p.ev.ncg;
  float* lse = rowbuf;
  float* T = rowbuf + B;
  // pass 1: row log-sum-exp
  g.cpg = p.ev.cpg;
  rc = dec_dispatch<MODE_NB_STATS>(g, ns, p.ev.nchunk, stream);
  if (rc) return rc;
  {
    FinishJobs J = {};
    J.nchunk = p.ev.nchunk; J.B = B;
    J.row_slab = p.row_slab; J.row_mode = 1; J.row_scale = 1.f; J.row_out0 = lse; J.n_rows = (B + 63) / 64;
    rc = launch_finish(J, stream);
    if (rc) return rc;
  }
  // pass 2: loss and T
  g.rowlse = lse;
  rc = dec_dispatch<MODE_NB_LOSS>(g, ns, p.ev.nchunk, stream);
  if (rc) return rc;
  {
    FinishJobs J = {};
    J.nchunk = p.ev.nchunk; J.B = B;
    J.loss_part = ws_d; J.nblk = nblk; J.n_total = (double)B * (double)F; J.loss = loss;
    if (train) {
      J.row_slab = p.row_slab; J.row_mode = (dlat && lat) ? 2 : 0; J.row_scale = -1.f / ((float)B * (float)F);
      J.row_aux = lat; J.row_out0 = T; J.row_out1 = dlat; J.n_rows = (B + 63) / 64;
    }
    rc = launch_finish(J, stream);
    if (rc) return rc;
  }
  if (!train) return SCP_OK;
  // pass 3: gradients
  g.rowT = T;
  g.cpg = p.cpg;
  rc = dec_dispatch<MODE_NB_GRAD>(g, ns, p.nchunk, stream);
  if (rc) return rc;
  {
    FinishJobs J = {};
    J.nchunk = p.nchunk; J.B = B;
    add_feature_jobs(J, p, g, dW, lddw, db, fs ? dfs : nullptr, theta_param ? dtheta : nullptr);
    J.dh_slab = p.dh_slab; J.dh = dh; J.lddh = lddh; J.K = K; J.KP = p.KP; J.n_dh = (int)(((long)B * p.KP + 63) / 64);
    rc = launch_finish(J, stream);
    if (rc) return rc;
  }
  return SCP_OK;
}
