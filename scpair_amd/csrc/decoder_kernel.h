// Fused linear decoder + likelihood kernels (gfx950).
//
//   Bernoulli decoder + BCE      : model.py:164-190 (Output_Module_Ber) + loss.py:36-41
//   (ZI)NB decoder + log-lik     : model.py:229-259 (Output_Module_Raw) + loss.py:46-51,
//                                  scVI_distribution.py:29-105
//
// The [cells x features] logits are never written to HBM.  A 256-thread workgroup owns a chunk of
// NPT feature tiles (64 features each) and walks over all 64-cell blocks of the minibatch:
//     Z tile (64x64)  = h_blk (64 x K) * W_tile^T          bf16 MFMA 32x32x16, fp32 accumulate
//     epilogue        : bias, sigmoid / relu+softmax+NB terms, loss, dZ          (registers)
//     dW tile (64xK) += dZ^T * h_blk   (accumulated in registers across ALL cell blocks -> complete)
//     dh blk  (64xK) += dZ * W_tile    (accumulated across the chunk's tiles -> one slab per chunk)
// Labels come from the minibatch tile list (tiles.hip) and are densified into LDS per (block, tile).
// Reductions across workgroups are slabs summed in a fixed order by a finishing kernel, so results
// are bitwise reproducible run to run.
//
// NB needs row statistics (softmax over ALL genes, model.py:242), so it runs as three passes over
// the same skeleton, recomputing the K<=64 deep GEMM instead of storing [cells x genes] tensors:
//     pass STATS : per-row log-sum-exp of a' = relu(z)*[sigmoid(fs)]
//     pass LOSS  : log-likelihood sum, T_row = sum_j (dll/dmu_j * mu_j)
//     pass GRAD  : dZ -> dW, dh, db (+ feature_scalar / feature-dispersion gradients)
#pragma once
#include <type_traits>
#include "common.h"
#include "scpair_hip.h"

#define CB 64        // cells per block
#define FT 64        // features per tile
#define LDZ 72       // padded leading dim of the 64-wide bf16 images

// _TH: the NB likelihood / gradient passes with dispersion = 'feature' (theta_param given).  The default, dispersion
// 'feature-cell' (theta = exp(relu(z)) >= 1), is its own instantiation: theta + eps == theta in fp32 (eps = 1e-8 is below
// half an ulp of 1), so log(theta + eps) is relu(z) itself and theta / (theta + eps) is 1 -- one logarithm (and in the
// gradient pass one reciprocal) per element less, and no run-time selects on the dispersion kind.
enum { MODE_BER_TRAIN = 0, MODE_BER_EVAL = 1, MODE_NB_STATS = 2, MODE_NB_LOSS = 3, MODE_NB_GRAD = 4, MODE_NB_LOSS_TH = 5,
       MODE_NB_GRAD_TH = 6 };

struct DecArgs {
  int B, F, K;                 // cells, output features, contraction depth (latent + covariates)
  const float* h; long ldh;    // [B][K]
  const float* W; long ldw;    // [F][K]
  const float* bias;           // [F]
  const float* fs;             // feature_scalar [F] or null
  const float* lrow; int lrow_n;   // per-cell library term (Ber: logit; NB: library size L) or null
  const float* rowlse;         // NB: per-cell log-sum-exp              [B]
  const float* rowT;           // NB: per-cell T                        [B]
  const float* theta_param;    // NB dispersion='feature': log theta    [F] or null (feature-cell)
  // labels (tile list of the output modality)
  const int32_t* t_ptr; const int32_t* t_key; const float* t_val;
  const int32_t* nnan;         // device scalar: number of NaN labels (Ber) or null
  const float* gout;           // upstream gradient scalar or null (=1)
  // outputs
  double* loss_part;           // [grid]
  float* dW; long lddw;        // [F][K]
  float* db;                   // [F]
  float* dfs;                  // [F]
  float* dtheta;               // [F]
  float* dh_slab;              // [grid][B][KP]
  float* row_slab;             // [grid][B][2]   (Ber: dl ; NB stats: m,s ; NB loss: T)
  // cell groups: blockIdx.y owns cell blocks [y*cpg, (y+1)*cpg); per-feature outputs of group y go to
  // dW + y*cg_stride_w, db/dfs/dtheta + y*F (summed by the finishing kernels when ncg > 1)
  int cpg;
  long cg_stride_w;
#ifdef DEC_PROF
  unsigned long long* prof;    // diagnostic build only: per-phase timestamps of one wave
#endif
};

#ifdef DEC_PROF
#define PROF_MARK(m) do { if (prof_on) g.prof[prof_it * 12 + (m)] = clock64(); } while (0)
#else
#define PROF_MARK(m) do { } while (0)
#endif

template <int NS>
__device__ __forceinline__ void img_put(__bf16* img, int rows, int ld, int row, int col, float v) {
  __bf16 t[3];
  bf16_split<NS>(v, t);
#pragma unroll
  for (int s = 0; s < NS; s++) img[(s * rows + row) * ld + col] = t[s];
}

// A-operand fragment (32 rows m0.., 16 k-values k0..) of an image stored k-major, img[k][LDZ] with m contiguous: the
// transposing LDS read hands lane i of a 16-lane group column i of a 4 (k) x 16 (m) block (lane 4q+p addresses row q,
// columns 4p..4p+3), so dZ^T -- written once, as the weight gradient's operand -- also serves the dh product.
// EXEC must be all ones (callers branch wave-uniformly).
typedef __attribute__((ext_vector_type(4))) short dec_short4;
// bf16 elements of the dZ^T image region: NS images of [64][LDZ], at least the [64][65] floats of the row-reduction tile
template <int NS> constexpr int dec_dz_elems() { return (NS * FT * LDZ * 2 >= CB * 65 * 4) ? NS * FT * LDZ : CB * 65 * 2; }

template <int LD = LDZ>
__device__ __forceinline__ bf16x8 dec_read_tr(const __bf16* img, int k0, int m0, int lane) {
  const int g = lane >> 5, mb = ((lane >> 4) & 1) * 16, q = (lane & 15) >> 2, p = lane & 3;
  typedef __attribute__((address_space(3))) dec_short4 lds_s4;
  const __bf16* a = img + (k0 + 8 * g + q) * LD + m0 + mb + 4 * p;
  union { dec_short4 h[2]; bf16x8 v; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)a);
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(a + 4 * LD));
  return u.v;
}

template <int NS, int KP, int NPT, int MODE, bool VEC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void dec_kernel(DecArgs g) {
  constexpr bool IS_BER = (MODE == MODE_BER_TRAIN || MODE == MODE_BER_EVAL);
  constexpr bool NB_LOSS = (MODE == MODE_NB_LOSS || MODE == MODE_NB_LOSS_TH);
  constexpr bool NB_GRAD = (MODE == MODE_NB_GRAD || MODE == MODE_NB_GRAD_TH);
  constexpr bool THP = (MODE == MODE_NB_LOSS_TH || MODE == MODE_NB_GRAD_TH);       // dispersion = 'feature'
  constexpr bool TRAIN = (MODE == MODE_BER_TRAIN || NB_GRAD);
  constexpr bool NEED_LABELS = (MODE != MODE_NB_STATS);
  constexpr int LDK_ = KP + 8;
  constexpr int NTK = (KP + 31) / 32;      // 32-wide column tiles covering K
  constexpr bool FAST = (NS == 1);
  constexpr int DZ_ELEMS = dec_dz_elems<NS>();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* hA = (__bf16*)smem_raw;                 // [NS][64][LDK_]   A of Z; B of dW through the transposing read
  __bf16* WB = hA + NS * CB * LDK_;               // [NS][64][LDK_]   B of Z; B of dh through the transposing read
  __bf16* dzT = WB + NS * FT * LDK_;              // [NS][64][LDZ]    A of dW ([feature][cell]); A of dh through the
                                                  // transposing read; also the scratch tile rt[64][65] of the row reductions
  float* ytile = (float*)(dzT + DZ_ELEMS);        // [64][64]
  float* rowv = ytile + CB * FT;                  // [3][64]  per-cell inputs
  float* rred = rowv + 3 * CB;                    // [2][2][64] per-cell partial reductions
  float* cred = ytile;                            // [2][NPT][3][64] per-feature partials; aliases ytile (used after the loops)
  __shared__ double dred[8];
  __shared__ int tflag;                           // label tile holds a value other than 0/1 (or NaN)

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, hq = lane >> 5;
  const int ti = wave >> 1, tj = wave & 1;        // Z tile: cell half, feature half
  const int f_chunk0 = blockIdx.x * NPT * FT;
  const int ncb = (g.B + CB - 1) / CB;
  const int cb_beg = blockIdx.y * g.cpg, cb_end = min(ncb, cb_beg + g.cpg);
  const int bid = blockIdx.y * gridDim.x + blockIdx.x;

  float inv_n = 1.f;
  if (TRAIN) {
    double n = (double)g.B * (double)g.F;
    if (IS_BER && g.nnan) n -= (double)g.nnan[0];
    inv_n = (float)((g.gout ? (double)g.gout[0] : 1.0) / n);
  }

  f32x16 accW[NPT];
  float db_acc[NPT], dfs_acc[NPT], dth_acc[NPT];
#pragma unroll
  for (int t = 0; t < NPT; t++) {
#pragma unroll
    for (int e = 0; e < 16; e++) accW[t][e] = 0.f;
    db_acc[t] = dfs_acc[t] = dth_acc[t] = 0.f;
  }
  double loss_acc = 0.0;

  // ---- software pipeline state: operands of the UPCOMING iteration live in registers -------------------
  // Staging map of a 64 x KP operand block: thread -> (row sp = tid/4, columns [sq*KQ, sq*KQ + KQ)); one base
  // address per operand and immediate offsets keep the address arithmetic out of the register file.
  constexpr int KQ = KP / 4;
  constexpr int EPF = 2;                        // label entries prefetched per thread
  const int sp = tid >> 2, sq = tid & 3;
  // VEC (chosen by the host: K and the leading dimensions multiples of 4, 16-byte aligned bases, operands below 1 GiB):
  // the operand blocks come in through raw buffer loads.  The hardware range check returns zeros for rows past the
  // end, and a per-thread column offset that is either the chunk's byte offset or a value past every buffer does the
  // same for the K-padding columns -- no predication, no branches, 32-bit offsets.  Otherwise: scalar predicated loads.
  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc((void*)g.W, 0, VEC ? (int)(((long)(g.F - 1) * g.ldw + g.K) * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_h =
      __builtin_amdgcn_make_buffer_rsrc((void*)g.h, 0, VEC ? (int)(((long)(g.B - 1) * g.ldh + g.K) * 4) : 0, 0x00020000);
  int coff[KQ / 4];
#pragma unroll
  for (int i = 0; i < KQ / 4; i++) coff[i] = (sq * KQ + 4 * i < g.K) ? (sq * KQ + 4 * i) * 4 : (1 << 30);
  float wreg[KQ], hreg[KQ];
  int ekey[EPF];
  float evl[EPF];
  // implicit-ones data (t_val == null): the value loads read the key words instead and the result is ignored where the
  // entry is used -- a branch around the loads would end in register copies that wait for every load in flight
  const float* e_val = g.t_val ? g.t_val : (const float*)g.t_key;
  int ep0 = 0, ep1 = 0;                         // entry range of the upcoming iteration
  int pn0 = 0, pn1 = 0;                         // entry range of the iteration after it (ptr loaded one step earlier)
  auto load_rows = [&](float* reg, const float* base, long ld, int row, int nrows, __amdgpu_buffer_rsrc_t rsrc) {
    if constexpr (VEC) {
      const int roff = row * (int)ld * 4;
#pragma unroll
      for (int i = 0; i < KQ / 4; i++) {
        const float4 t = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, roff + coff[i], 0, 0));
        reg[4 * i] = t.x; reg[4 * i + 1] = t.y; reg[4 * i + 2] = t.z; reg[4 * i + 3] = t.w;
      }
    } else {
      const float* src = base + (long)row * ld + sq * KQ;
      const bool rok = row < nrows;
#pragma unroll
      for (int i = 0; i < KQ; i++) reg[i] = (rok && sq * KQ + i < g.K) ? src[i] : 0.f;
    }
  };
  // The per-feature scalars of a tile (bias, feature_scalar, log theta) and the per-cell inputs of a block travel with
  // the operand prefetch: loaded where they are used, they would wait for every prefetch load issued before them
  // (loads return in order), once per iteration.  Indices are clamped instead of predicated.
  float bias_n = 0.f, fs_n = 0.f, th_n = 0.f;      // upcoming tile, feature tj*32 + r
  float row_n[3] = {0.f, 0.f, 0.f};                // upcoming block, cell (tid & 63): lrow | rowlse | rowT
  auto load_w = [&](int f0n) {
    load_rows(wreg, g.W, g.ldw, f0n + sp, g.F, rsrc_w);
    const int fc = min(f0n + (wave & 1) * 32 + (lane & 31), g.F - 1);
    bias_n = g.bias[fc];
    if (g.fs) fs_n = g.fs[fc];
    if (THP) th_n = g.theta_param[fc];
  };
  auto load_h = [&](int c0n) {
    load_rows(hreg, g.h, g.ldh, c0n + sp, g.B, rsrc_h);
    const int cc = min(c0n + lane, g.B - 1);
    if (g.lrow) row_n[0] = g.lrow[g.lrow_n == 1 ? 0 : cc];
    if (!IS_BER && MODE != MODE_NB_STATS && g.rowlse) row_n[1] = g.rowlse[cc];
    if (NB_GRAD && g.rowT) row_n[2] = g.rowT[cc];
  };
  // registers -> row-major image [NS][64][LDK_] (8-byte stores); rows past the end and the K-padding columns arrive as
  // zeros from either load path.  No transposed images: the dW / dh products read these through the transposing LDS
  // read (columns past KP hold whatever lies there and only reach output columns >= K, which are never stored)
  auto put_rows = [&](const float* reg, __bf16* img) {
    __bf16 tt[KQ][3];
#pragma unroll
    for (int i = 0; i < KQ; i++) bf16_split<NS>(reg[i], tt[i]);
#pragma unroll
    for (int s2 = 0; s2 < NS; s2++) {
#pragma unroll
      for (int i = 0; i < KQ / 4; i++) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = tt[4 * i + j][s2];
        *(bf16x4*)&img[(s2 * 64 + sp) * LDK_ + sq * KQ + 4 * i] = o;
      }
    }
  };
  // The tile-range words are uniform, and a uniform address makes these scalar loads: scalar loads share their counter
  // with LDS traffic, so the wait in front of the next barrier would expose their full latency (a miss every time,
  // the range array is large) once per iteration.  An opaque per-lane zero keeps them vector loads (counted with the
  // rest of the prefetch and consumed an iteration later).
  int vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  auto load_ptr = [&](int cbx, int f0x) {
    long base = (long)cbx * g.F + vzero;
    pn0 = g.t_ptr[base + f0x];
    pn1 = g.t_ptr[base + min(f0x + FT, g.F)];
  };
  auto load_entries = [&]() {
    ep0 = pn0; ep1 = pn1;
#pragma unroll
    for (int i = 0; i < EPF; i++) {
      const int e = ep0 + tid + 256 * i;
      const int ec = max(min(e, ep1 - 1), 0);     // clamped (the entry arrays hold at least one element): no predicated load
      ekey[i] = g.t_key[ec];                      // validity is recomputed from (ep0, ep1) where the entry is used:
      evl[i] = e_val[ec];                         // no use of the loaded values here (it would wait for the loads)
    }
  };
  auto clear_labels = [&]() {
    float4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (int idx = tid; idx < CB * FT / 4; idx += 256) ((float4*)ytile)[idx] = z4;
    if (tid == 0) tflag = 0;
  };
  if (TRAIN && NEED_LABELS) clear_labels();      // first iteration; the barrier at its top orders it before the scatter
  const int ntl = min(NPT, (g.F - f_chunk0 + FT - 1) / FT);     // valid feature tiles of this chunk (>= 1)
  if (cb_beg < cb_end) {
    load_h(cb_beg * CB);
    load_w(f_chunk0);
    if (NEED_LABELS) {
      load_ptr(cb_beg, f_chunk0);
      load_entries();
      // range of the second iteration
      if (ntl > 1) load_ptr(cb_beg, f_chunk0 + FT);
      else if (cb_beg + 1 < cb_end) load_ptr(cb_beg + 1, f_chunk0);
    }
  }

#ifdef DEC_PROF
  const bool prof_on = (tid == 0) && (blockIdx.x == gridDim.x / 2) && (blockIdx.y == 0) && g.prof;
  int prof_it = 0;
#endif
  for (int cb = cb_beg; cb < cb_end; cb++) {
    const int c0 = cb * CB;
    f32x16 accH;
#pragma unroll
    for (int e = 0; e < 16; e++) accH[e] = 0.f;
    float racc0[16], racc1[16];
#pragma unroll
    for (int e = 0; e < 16; e++) {
      racc0[e] = 0.f;        // NB_STATS: running maximum, floored at 0 (a' = relu(z) sigmoid(fs) >= 0)
      racc1[e] = 0.f;
    }

    // The tile loop stays rolled (one copy of the body keeps the register budget at 2 waves/SIMD): the tile in
    // flight always accumulates into slot 0 and the per-tile accumulators rotate by one slot per iteration, so
    // after the NPT iterations of a cell block every slot is back in place.
#pragma unroll 1
    for (int t = 0; t < NPT; t++) {
      const int f0 = f_chunk0 + t * FT;
      if (t < ntl) {                              // uniform
      PROF_MARK(0);
      __syncthreads();                            // previous iteration's readers are done
      PROF_MARK(1);
      if (t == 0) {
        // ---- h block (both layouts) and the per-cell inputs, from the prefetch registers ---------------
        put_rows(hreg, hA);
        if (tid < CB) {
          const bool ok = c0 + tid < g.B;
          if (IS_BER) {
            rowv[tid] = g.lrow ? sigm_<FAST>(ok ? row_n[0] : 0.f) : 1.f;
          } else {
            rowv[tid] = (g.lrow && ok) ? row_n[0] : 1.f;                   // L
            rowv[CB + tid] = (g.rowlse && ok) ? row_n[1] : 0.f;            // lse
            rowv[2 * CB + tid] = (g.rowT && ok) ? row_n[2] : 0.f;          // T
          }
        }
      }
      // ---- W tile (both layouts) from the prefetch registers, clear label tile ---------------------
      put_rows(wreg, WB);
      const float bias_c = bias_n, fs_c = fs_n, th_c = th_n;      // this tile's scalars, before the next prefetch
      if (NEED_LABELS) {
        if (!TRAIN) {
          // (the training passes clear the tile right after the barrier that ends their epilogue, see below: the
          // barrier at the top of the iteration then separates the clear from the scatter and this one is not needed)
          clear_labels();
          PROF_MARK(2);
          __syncthreads();
        }
        PROF_MARK(3);
        bool odd = false;
#pragma unroll
        for (int i = 0; i < EPF; i++)
          if (ep0 + tid + 256 * i < ep1) {
            const float v = g.t_val ? evl[i] : 1.f;
            ytile[(ekey[i] & 63) * FT + ((ekey[i] >> 6) - f0)] = v;
            odd |= !(v == 1.f);
          }
        for (int e = ep0 + tid + 256 * EPF; e < ep1; e += 256) {
          int key = g.t_key[e];
          float v = g.t_val ? g.t_val[e] : 1.f;
          ytile[(key & 63) * FT + ((key >> 6) - f0)] = v;
          odd |= !(v == 1.f);
        }
        if (IS_BER && odd) tflag = 1;
      }
      PROF_MARK(11);
      // ---- issue the loads of the NEXT iteration; they complete under the MFMA + epilogue below -------
      {
        const bool more_t = (t + 1 < ntl);
        const int cbn = more_t ? cb : cb + 1;
        const int f0n = more_t ? f0 + FT : f_chunk0;
        if (cbn < cb_end) {
          // the entry loads come first: their addresses need the range words fetched one iteration ago, and a wait
          // for those placed after this iteration's operand loads would wait for these too (loads return in order)
          if (NEED_LABELS) {
            load_entries();                       // uses the range fetched one iteration ago
            // range of the iteration after the next one
            const bool more_t2 = more_t ? (t + 2 < ntl) : (1 < ntl);
            const int cb2 = more_t2 ? cbn : cbn + 1;
            const int f02 = more_t2 ? f0n + FT : f_chunk0;
            if (cb2 < cb_end) load_ptr(cb2, f02);
          }
          load_w(f0n);
          if (!more_t) load_h(cbn * CB);
        }
      }
      PROF_MARK(4);
      __syncthreads();
      PROF_MARK(5);
      // ---- Z = h W^T -----------------------------------------------------------------------------
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; e++) acc[e] = 0.f;
#pragma unroll
      for (int kk = 0; kk < KP / 16; kk++) {
        bf16x8 a[NS], b[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) {
          a[s] = *(const bf16x8*)&hA[(s * CB + ti * 32 + r) * LDK_ + kk * 16 + hq * 8];
          b[s] = *(const bf16x8*)&WB[(s * FT + tj * 32 + r) * LDK_ + kk * 16 + hq * 8];
        }
#pragma unroll
        for (int sa = 0; sa < NS; sa++)
#pragma unroll
          for (int sb = 0; sb < NS; sb++)
            if (sa + sb < NS) acc = mfma32(a[sa], b[sb], acc);
      }
      // ---- epilogue ------------------------------------------------------------------------------
      PROF_MARK(6);
      const int fl = tj * 32 + r;                 // feature inside tile
      const int f = f0 + fl;
      const bool f_ok = f < g.F;
      const float bcol = f_ok ? bias_c : 0.f;
      float fcol = 1.f;
      if (g.fs && f_ok) fcol = sigm_<FAST>(fs_c);
      float thcol = 0.f;
      if (THP && f_ok) thcol = exp_<FAST>(th_c);
      float dzv[16];
      float lsum = 0.f;     // fp32 partial of this tile, folded into the fp64 accumulator once per tile
      // Bernoulli, interior tile with 0/1 labels only: no bounds / NaN / soft-label handling (same algebra)
      const bool ber_fast = IS_BER && (f0 + FT <= g.F) && (c0 + CB <= g.B) && (tflag == 0);
      if (ber_fast) {
        // Straight line over pairs of elements (the compiler packs them into v_pk_* forms).  Deferred to the points where
        // the sums leave the workgroup: the 1 / n of the mean (inv_n) and the (1 - sr) of the per-cell gradient; the
        // likelihood is summed in log2 units (one v_log per element, no product) and scaled once per tile; bias and
        // the -log2(e) of the exponential share one fma.
        typedef float f2 __attribute__((ext_vector_type(2)));
        constexpr float L2E = 1.4426950408889634f;
        const float nb2 = -bcol * L2E;
        f2 su2 = {0.f, 0.f}, sd2 = {0.f, 0.f}, ls2 = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          f2 uu, dd, ll;
#pragma unroll
          for (int j = 0; j < 2; j++) {
            const int cl = ti * 32 + acc_row(e + j, lane);
            float px;
            if constexpr (FAST) px = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(fmaf(acc[e + j], -L2E, nb2)));
            else px = sigm_<false>(acc[e + j] + bcol);
            const float sr = rowv[cl];
            const float q = px * sr * fcol;
            const float omq = 1.f - q;
            const float yv = ytile[cl * FT + fl];
            const float arg = (yv == 0.f) ? omq : q;
            if constexpr (FAST) ll[j] = fmaxf(__builtin_amdgcn_logf(arg), -100.f * L2E);
            else ll[j] = fmaxf(logf(arg), -100.f);
            if (TRAIN) {
              const float u = div_<FAST>(q - yv, fmaxf(q * omq, 1e-12f)) * q;     // dL/dq * q  (x n)
              uu[j] = u;
              dd[j] = u * (1.f - px);
              dzv[e + j] = dd[j];
              racc0[e + j] += u;
            }
          }
          ls2 -= ll;
          if (TRAIN) {
            su2 += uu;
            sd2 += dd;
          }
        }
        lsum = (ls2[0] + ls2[1]) * (FAST ? 0.69314718f : 1.f);
        if (TRAIN) {
          dfs_acc[0] += (su2[0] + su2[1]) * (1.f - fcol);
          db_acc[0] += sd2[0] + sd2[1];
        }
      } else if (NB_LOSS && (f0 + FT <= g.F) && (c0 + CB <= g.B)) {
        // NB likelihood pass, interior tile.  Phase 1 treats every label as 0 in straight-line code (the transcendental
        // chains of a group of elements interleave); phase 2 adds the terms of the non-zero labels (data-dependent
        // loops).  Same algebra as the general loop below.  (The GRAD pass keeps the general loop: the same split
        // spills there, its register budget is spent on the dW / dh accumulators.)
        const float eps = 1e-8f;
        constexpr int EH = 8;                         // elements per group
#pragma unroll
        for (int e0 = 0; e0 < 16; e0 += EH) {
          __builtin_amdgcn_sched_barrier(0);          // keep the groups apart: interleaving all 16 chains spills
          float mu_[EH], rs_[EH], ls_[EH];            // ls = log(theta + mu + eps)
#pragma unroll
          for (int j = 0; j < EH; j++) {
            const int e = e0 + j;
            const int cl = ti * 32 + acc_row(e, lane);
            const float zz = acc[e] + bcol;
            const float a_ = fmaxf(zz, 0.f);
            const float ap = a_ * fcol;
            const float L = rowv[cl], lse = rowv[CB + cl];
            const float scale = exp_<FAST>(ap - lse);
            const float mu = L * scale;
            const float th = THP ? thcol : exp_<FAST>(a_);
            const float s_ = th + mu + eps;
            const float lt = (THP || !FAST) ? log_<FAST>(th + eps) : a_, ls = log_<FAST>(s_);
            const float rs = div_<FAST>(1.f, s_);
            const float gmu = -th * rs;
            mu_[j] = mu;
            rs_[j] = rs;
            lsum -= th * (lt - ls);
            racc0[e] += gmu * mu;
            ls_[j] = ls;
          }
#pragma unroll
          for (int j = 0; j < EH; j++) {
            const int e = e0 + j;
            const int cl = ti * 32 + acc_row(e, lane);
            const float yv = ytile[cl * FT + fl];
            if (yv > eps) {
              const float th = THP ? thcol : exp_<FAST>(fmaxf(acc[e] + bcol, 0.f));
              const float mu = mu_[j];
              const float gnz = yv * (div_<FAST>(1.f, mu + eps) - rs_[j]);
              lsum -= yv * (log_<FAST>(mu + eps) - ls_[j]) + lgam3<FAST>(yv, th);
              racc0[e] += gnz * mu;
            }
          }
        }
      } else
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int cl = ti * 32 + acc_row(e, lane);
        const bool ok = f_ok && (c0 + cl < g.B);
        dzv[e] = 0.f;
        if (IS_BER) {
          float zz = acc[e] + bcol;
          float px = sigm_<FAST>(zz);
          float sr = rowv[cl];
          float q = px * sr * fcol;
          float yv = ytile[cl * FT + fl];
          if (ok && yv == yv) {
            float omq = 1.f - q;
            if (yv == 0.f || yv == 1.f) {
              // binary label: only one of the two BCE terms is non-zero
              float arg = (yv == 0.f) ? omq : q;
              lsum -= fmaxf(log_<FAST>(arg), -100.f);
            } else {
              lsum -= yv * fmaxf(log_<FAST>(q), -100.f) + (1.f - yv) * fmaxf(log_<FAST>(omq), -100.f);
            }
            if (TRAIN) {
              float dq = div_<FAST>(q - yv, fmaxf(q * omq, 1e-12f));       // inv_n and (1 - sr): applied where the sums leave
              float t1 = dq * px;
              dzv[e] = t1 * sr * fcol * (1.f - px);
              racc0[e] += t1 * fcol * sr;
              dfs_acc[0] += t1 * sr * fcol * (1.f - fcol);
              db_acc[0] += dzv[e];
            }
          }
        } else {
          float zz = acc[e] + bcol;
          float a_ = fmaxf(zz, 0.f);
          float ap = a_ * fcol;
          if (MODE == MODE_NB_STATS) {
            // branch-free: an element outside the matrix leaves the maximum alone and adds 0
            const float m_old = racc0[e];
            const float m_new = fmaxf(m_old, ok ? ap : 0.f);
            const float term = exp_<FAST>(ap - m_new);
            racc1[e] = racc1[e] * exp_<FAST>(m_old - m_new) + (ok ? term : 0.f);
            racc0[e] = m_new;
          } else if (ok) {
            const float eps = 1e-8f;
            float L = rowv[cl], lse = rowv[CB + cl];
            float scale = exp_<FAST>(ap - lse);
            float mu = L * scale;
            float th = THP ? thcol : exp_<FAST>(a_);
            float s_ = th + mu + eps;
            float lt = (THP || !FAST) ? log_<FAST>(th + eps) : a_, ls = log_<FAST>(s_);
            float yv = ytile[cl * FT + fl];
            float rs = div_<FAST>(1.f, s_);
            float gmu = -th * rs;
            if (yv > eps) gmu += yv * (div_<FAST>(1.f, mu + eps) - rs);
            if (NB_LOSS) {
              float ll = th * (lt - ls);
              if (yv > eps) ll += yv * (log_<FAST>(mu + eps) - ls) + lgam3<FAST>(yv, th);
              lsum -= ll;
              racc0[e] += gmu * mu;
            } else {   // MODE_NB_GRAD
              float gth = lt - ls + ((THP || !FAST) ? div_<FAST>(th, th + eps) : 1.f) - th * rs;
              if (yv > eps) gth += -yv * rs + digam2<FAST>(yv, th);
              float T = rowv[2 * CB + cl];
              float dap = mu * gmu - scale * T;
              float da = dap * fcol;
              if (THP)
                dth_acc[0] -= gth * th;
              else
                da += gth * th;
              float d = (zz > 0.f) ? -da : 0.f;                 // inv_n: applied where the sums leave the workgroup
              dzv[e] = d;
              db_acc[0] += d;
              dfs_acc[0] -= dap * a_ * fcol * (1.f - fcol);
            }
          }
        }
      }
      loss_acc += (double)lsum;
      PROF_MARK(7);
      if (TRAIN) {
        // dzT[feature][cell]: registers 4q..4q+3 are 4 consecutive cells -> one 8-byte store per term.  (No [cell][feature]
        // image: the dh product reads dzT through the transposing LDS read.)
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
          __bf16 tt[4][3];
#pragma unroll
          for (int j = 0; j < 4; j++) bf16_split<NS>(dzv[4 * q4 + j], tt[j]);
          const int cl0 = ti * 32 + 8 * q4 + 4 * hq;
#pragma unroll
          for (int s2 = 0; s2 < NS; s2++) {
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; j++) o[j] = tt[j][s2];
            *(bf16x4*)&dzT[(s2 * FT + fl) * LDZ + cl0] = o;
          }
        }
        PROF_MARK(8);
        __syncthreads();
        PROF_MARK(9);
        if (NEED_LABELS) clear_labels();          // every wave is past its epilogue: the label tile is free again
        // ---- dW tile += dZ^T h ; dh block += dZ W -----------------------------------------------
        const int mi = wave & 1, nj = wave >> 1;
        if (nj < NTK) {
#pragma unroll
          for (int kk = 0; kk < CB / 16; kk++) {
            bf16x8 a[NS], b[NS], a2[NS], b2[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) {
              a[s] = *(const bf16x8*)&dzT[(s * FT + mi * 32 + r) * LDZ + kk * 16 + hq * 8];
              b[s] = dec_read_tr<LDK_>(hA + s * CB * LDK_, kk * 16, nj * 32, lane);
              a2[s] = dec_read_tr(dzT + s * FT * LDZ, kk * 16, mi * 32, lane);
              b2[s] = dec_read_tr<LDK_>(WB + s * FT * LDK_, kk * 16, nj * 32, lane);
            }
#pragma unroll
            for (int sa = 0; sa < NS; sa++)
#pragma unroll
              for (int sb = 0; sb < NS; sb++)
                if (sa + sb < NS) {
                  accW[0] = mfma32(a[sa], b[sb], accW[0]);
                  accH = mfma32(a2[sa], b2[sb], accH);
                }
          }
        }
      }
      }  // t < ntl
      if (TRAIN) {
        f32x16 w0 = accW[0];
        float b0 = db_acc[0], s0 = dfs_acc[0], h0 = dth_acc[0];
#pragma unroll
        for (int q = 0; q + 1 < NPT; q++) {
          accW[q] = accW[q + 1];
          db_acc[q] = db_acc[q + 1];
          dfs_acc[q] = dfs_acc[q + 1];
          dth_acc[q] = dth_acc[q + 1];
        }
        accW[NPT - 1] = w0;
        db_acc[NPT - 1] = b0; dfs_acc[NPT - 1] = s0; dth_acc[NPT - 1] = h0;
      }
      PROF_MARK(10);
#ifdef DEC_PROF
      prof_it++;
#endif
    }  // feature tiles

    // ---- per-cell results of this block ---------------------------------------------------------
    if (TRAIN) {
      const int mi = wave & 1, nj = wave >> 1;
      if (nj < NTK) {
        int k = nj * 32 + r;
        if (k < g.K) {
          // slab rows are padded to whole cell blocks: one base address, compile-time row offsets, no predicates
          float* dst = g.dh_slab + ((long)blockIdx.x * (ncb * CB) + c0 + mi * 32 + 4 * hq) * KP + k;
#pragma unroll
          for (int e = 0; e < 16; e++) dst[((e & 3) + 8 * (e >> 2)) * KP] = accH[e] * inv_n;
        }
      }
    }
    if (MODE == MODE_BER_TRAIN || NB_LOSS || MODE == MODE_NB_STATS) {
      // Row reductions over the 64 feature columns of the chunk's tiles: the per-lane partials go through an LDS
      // tile rt[cell][65] (odd stride: conflict-free both ways; it reuses the dZ images, idle here) and are summed
      // by 4 threads per cell in a fixed order.
      float* rt = (float*)dzT;
      const int rrow = tid >> 2, rq = tid & 3;
      auto row_sum = [&]() {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 16; j++) a += rt[rrow * 65 + rq * 16 + j];
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        return a;
      };
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 16; e++) rt[(ti * 32 + acc_row(e, lane)) * 65 + tj * 32 + r] = racc0[e];
      __syncthreads();
      float* dst = g.row_slab + ((long)blockIdx.x * g.B + c0 + rrow) * 2;
      const bool wr = (rq == 0) && (c0 + rrow < g.B);
      if (MODE == MODE_NB_STATS) {
        // log-sum-exp pieces (m, s) per lane -> row maximum first, then the sums rescaled to it
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < 16; j++) m = fmaxf(m, rt[rrow * 65 + rq * 16 + j]);
        m = fmaxf(m, __shfl_xor(m, 1, 64));
        m = fmaxf(m, __shfl_xor(m, 2, 64));
        if (rq == 0) rred[rrow] = m;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int cl = ti * 32 + acc_row(e, lane);
          float sc = (racc0[e] == -INFINITY) ? 0.f : racc1[e] * expf(racc0[e] - rred[cl]);
          rt[cl * 65 + tj * 32 + r] = sc;
        }
        __syncthreads();
        float ssum = row_sum();
        if (wr) {
          dst[0] = m;
          dst[1] = ssum;
        }
      } else {
        float ssum = row_sum();
        if (MODE == MODE_BER_TRAIN) ssum *= (1.f - rowv[rrow]) * inv_n;     // the factors the epilogue left out
        if (wr) {
          dst[0] = ssum;
          dst[1] = 0.f;
        }
      }
    }
    __syncthreads();
  }  // cell blocks

  // ---- chunk-level outputs ------------------------------------------------------------------------
  if (MODE != MODE_NB_STATS && !NB_GRAD) {
    double s = block_sum_d(loss_acc, dred);
    if (tid == 0) g.loss_part[bid] = s;
  }
  if (TRAIN) {
    const int mi = wave & 1, nj = wave >> 1;
#pragma unroll
    for (int t = 0; t < NPT; t++) {
      const int f0 = f_chunk0 + t * FT;
      if (f0 >= g.F) break;
      if (nj < NTK) {
        int k = nj * 32 + r;
        if (k < g.K) {
#pragma unroll
          for (int e = 0; e < 16; e++) {
            int f = f0 + mi * 32 + acc_row(e, lane);
            if (f < g.F) g.dW[(long)blockIdx.y * g.cg_stride_w + (long)f * g.lddw + k] = accW[t][e] * inv_n;
          }
        }
      }
      // per-feature sums: lanes (r, hq=0/1) of waves (ti=0/1, tj) hold partials of feature tj*32+r
      float v0 = (db_acc[t] + __shfl_xor(db_acc[t], 32, 64)) * inv_n;
      float v1 = (dfs_acc[t] + __shfl_xor(dfs_acc[t], 32, 64)) * inv_n;
      float v2 = (dth_acc[t] + __shfl_xor(dth_acc[t], 32, 64)) * inv_n;
      if (hq == 0) {
        cred[((ti * NPT + t) * 3 + 0) * FT + tj * 32 + r] = v0;
        cred[((ti * NPT + t) * 3 + 1) * FT + tj * 32 + r] = v1;
        cred[((ti * NPT + t) * 3 + 2) * FT + tj * 32 + r] = v2;
      }
    }
    __syncthreads();
    static_assert(2 * NPT * 3 * FT <= CB * FT, "cred must fit inside the label tile it aliases");
    for (int idx = tid; idx < NPT * FT; idx += 256) {
      int t = idx / FT, fl = idx % FT;
      int f = f_chunk0 + t * FT + fl;
      if (f < g.F) {
        float v0 = cred[((0 * NPT + t) * 3 + 0) * FT + fl] + cred[((1 * NPT + t) * 3 + 0) * FT + fl];
        float v1 = cred[((0 * NPT + t) * 3 + 1) * FT + fl] + cred[((1 * NPT + t) * 3 + 1) * FT + fl];
        float v2 = cred[((0 * NPT + t) * 3 + 2) * FT + fl] + cred[((1 * NPT + t) * 3 + 2) * FT + fl];
        const long fo = (long)blockIdx.y * g.F + f;
        if (g.db) g.db[fo] = v0;
        if (g.dfs) g.dfs[fo] = v1;
        if (g.dtheta) g.dtheta[fo] = v2;
      }
    }
  }
}


#define DEC_NPT 2        // feature tiles per chunk, training passes (a dW accumulator tile per feature tile stays in registers)
#define DEC_NPT_EVAL 4   // passes without gradients: the h block is staged and the row reductions run once per 4 tiles
template <int MODE> constexpr int dec_npt() { return (MODE == MODE_BER_TRAIN || MODE == MODE_NB_GRAD || MODE == MODE_NB_GRAD_TH) ? DEC_NPT : DEC_NPT_EVAL; }

template <int NS, int KP>
static size_t dec_smem() {
  size_t bf = (size_t)NS * ((size_t)CB * (KP + 8) * 2) + (size_t)dec_dz_elems<NS>();
  return bf * sizeof(__bf16) + sizeof(float) * (CB * FT + 3 * CB + 4 * CB);
}

template <int NS, int KP, int MODE, bool VEC>
static int dec_launch_v(const DecArgs& g, int grid, hipStream_t stream) {
  const int ncb_ = (g.B + CB - 1) / CB;
  const int ncg_ = (ncb_ + g.cpg - 1) / g.cpg;
  static uint64_t attr_done = 0;
  size_t smem = dec_smem<NS, KP>();
  int rc_ = scp_set_smem((const void*)dec_kernel<NS, KP, dec_npt<MODE>(), MODE, VEC>, smem, &attr_done);
  if (rc_) return rc_;
  hipLaunchKernelGGL((dec_kernel<NS, KP, dec_npt<MODE>(), MODE, VEC>), dim3(grid, ncg_), dim3(256), smem, stream, g);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

template <int NS, int KP, int MODE>
static int dec_launch(const DecArgs& g, int grid, hipStream_t stream) {
  // buffer-load path of the operand blocks: 16-byte rows and 32-bit byte offsets
  const bool vec = ((g.K & 3) == 0) && ((g.ldw & 3) == 0) && ((g.ldh & 3) == 0) && ((((uintptr_t)g.W) & 15) == 0) &&
                   ((((uintptr_t)g.h) & 15) == 0) && ((long)(g.F + 2 * FT) * g.ldw * 4 < (1L << 30)) &&
                   ((long)(g.B + 2 * CB) * g.ldh * 4 < (1L << 30));
  return vec ? dec_launch_v<NS, KP, MODE, true>(g, grid, stream) : dec_launch_v<NS, KP, MODE, false>(g, grid, stream);
}

// one translation unit per (NS, family) keeps the build parallel
template <int NS, int MODE>
static int dec_run_mode(const DecArgs& g, int kp, int grid, hipStream_t stream) {
  if (kp == 16) return dec_launch<NS, 16, MODE>(g, grid, stream);
  if (kp == 48) return dec_launch<NS, 48, MODE>(g, grid, stream);
  if constexpr (NS == 1) {
    if (kp == 64) return dec_launch<NS, 64, MODE>(g, grid, stream);
  }
  scp_set_error("decoder: unsupported (ns=%d, KP=%d)", NS, kp);
  return SCP_ERR_SHAPE;
}
