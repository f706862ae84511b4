/* scpair_hip.h -- C ABI of the MI355X-native scPair training hot path (libscpair_hip.so).
 *
 * The reference (quon-titative-biology/scPair) is pure Python on PyTorch and has no FFI of its own;
 * the seam it exposes is Python level.  Each entry point below replaces the
 * ATen-op chain behind one reference construct, cited as  <reference file>:<lines>.
 *
 * Conventions
 *   - plain pointers to DEVICE memory (HBM) unless a parameter says "host"; fp32 unless stated
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work (no sync, no malloc)
 *   - caller owns all buffers, including workspaces whose sizes the *_len / *_ws_* helpers return
 *   - return 0 on success, negative on error (SCP_ERR_*); scp_last_error() gives the message
 *   - `ns` selects MFMA input precision: 1 = bf16 inputs, 3 = bf16 hi+mid+lo split of every fp32
 *     operand (6 MFMAs per product, fp32-equivalent); accumulation is always fp32
 *   - a "tile list" (t_ptr, t_key, t_val) is the minibatch re-bucketing built by scp_tiles_build
 *   - multi-GPU: there is no communication entry point.  The gradient exchange (SURVEY.md 8b sketched an
 *     scp_allreduce_bucket) is torch.distributed on RCCL, called from Python (scpair_amd/distributed.py): a C wrapper
 *     would only forward to the same RCCL call and put communicator handles into this ABI.  The kernels' side of it is
 *     here: scp_spmm_dw_bf16_g16 writes gradients straight into the exchange buffer, scp_adamw_multi reads them (g_bf16).
 */
#ifndef SCPAIR_HIP_H
#define SCPAIR_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* scp_last_error(void);
int scp_version(void);

/* ---- minibatch assembly: replaces LazyGPUDataset.__getitem__ + default_collate + .to(device)
 *      (model.py:26-38, 704-717, 731-732) and the dense `lib = mtx.sum(1)` (model.py:509-510) ---- */
long scp_tiles_ptr_len(int B, int F);   /* int32 elements needed for `ptr` */
long scp_tiles_ws_len(int B, int F);    /* int32 elements needed for `ws`  */
int scp_tiles_build(const int64_t* indptr, const int32_t* indices, const float* values /* null: implicit 1 */,
                    const int64_t* rows, int B, int F, int32_t* ptr, int32_t* key, float* val /* nullable */,
                    uint32_t* pk /* nullable; written when values and val are given: the packed entries
                                    (bf16 bits of the value) << 16 | (key & 0xffff) the bf16 first-layer kernels read */,
                    float* lib /* [B] row sums, nullable */, int32_t* nnan /* [1], nullable */, int32_t* ws,
                    void* stream);
/* tile list -> dense [B][ld] fp32 labels (zeros elsewhere) for the likelihoods evaluated on materialised tensors
 * (ZINB, Gaussian: loss.py:22-27, 46-51); replaces the dense gather of model.py:26-38 */
int scp_tiles_to_dense(int B, int F, const int32_t* t_ptr, const int32_t* t_key, const float* t_val, float* out, long ld,
                       void* stream);
/* Optional input pre-transform on a minibatch tile list, in place (no counterpart in the reference, which feeds raw
 * counts: model.py:492-510; opt-in, default off): mode 1: v <- log1p(v); mode 2: v <- log1p(v * scale / lib[cell]). */
int scp_tiles_transform(int B, int F, const int32_t* ptr, const int32_t* key, float* val, uint32_t* pk /* nullable, kept in step */,
                        const float* lib, int mode, float scale, void* stream);

/* ---- first encoder layer on sparse counts: nn.Linear(F_in, H) of FCNN (utils.py:57) applied to
 *      the dense x in Input_Module (model.py:128-130, 136-140), forward and weight gradient ---- */
int scp_gemm_split_count(int K, int nsplit);   /* slabs actually produced for a requested nsplit */
int scp_spmm_fwd(int ncells, int N, int F, const int32_t* t_ptr, const int32_t* t_key, const float* t_val,
                 const float* W /* [N][w_sr] */, long w_sr, float* Y /* [nsplit][ncells][ldy] */, long ldy,
                 const float* bias /* nullable, added to slab 0 */, int nsplit, int ns, void* stream);
int scp_spmm_dw(int ncells, int N, int F, const int32_t* t_ptr, const int32_t* t_key, const float* t_val,
                const float* dY /* [ncells][ldy] */, long ldy, float* dW /* [N][ldw] */, long ldw, int ns,
                void* stream);

/* bf16 throughput path of the same layer (precision mode 'bf16'): the weight operand is a bf16 shadow copy
 * [N][ldw] (ldw % 8 == 0, zero padded; kept current by scp_adamw_multi / scp_cast_bf16) and dY comes transposed
 * as bf16 [N][ldt] (scp_transpose_bf16) */
int scp_cast_bf16(const float* src, long rows, long cols, long ld_src, void* dst_bf16, long ld_dst, void* stream);
/* dst = bf16(relu(src)): the input of the ZINB zero-inflation layer pi = Linear(relu(z)) (model.py:240,255-256) */
int scp_relu_cast_bf16(const float* src, long rows, long cols, long ld_src, void* dst_bf16, long ld_dst, void* stream);
int scp_transpose_bf16(const float* src, int rows, int cols, long ld_src, void* dst_bf16, long ld_dst, void* stream);
/* The same layer in the fp32-equivalent mode when the sparse operand is exact in bf16 (implicit ones: binarised peaks;
 * utils.py:57 / model.py:128-140 with fp32 weights): the fp32 weights split into three bf16 planes hi | mid | lo,
 * `plane` elements apart (scp_split3_bf16), X W^T = three launches of scp_spmm_fwd_bf16 into consecutive slabs; dY split
 * and transposed as bf16 [N][ld_dst] with the three terms side by side, columns t * rows + r (scp_split3_transpose_bf16),
 * dW = one launch of scp_spmm_dw_bf16 over a tile list of the minibatch repeated three times */
int scp_split3_bf16(const float* src, long rows, long cols, long ld_src, void* dst_bf16, long ld_dst, long plane, void* stream);
int scp_split3_transpose_bf16(const float* src, int rows, int cols, long ld_src, void* dst_bf16, long ld_dst, void* stream);
int scp_spmm_fwd_bf16_min_split(int F);   /* smallest nsplit scp_spmm_fwd_bf16 accepts for F features */
/* t_pk: the packed entries of scp_tiles_build; required when t_val is given (the kernels read them instead of key / val) */
int scp_spmm_fwd_bf16(int ncells, int N, int F, const int32_t* t_ptr, const int32_t* t_key, const float* t_val,
                      const uint32_t* t_pk, const void* Wbf, long ldw, float* Y, long ldy, int nsplit, void* stream);
int scp_spmm_dw_bf16(int ncells, int N, int F, const int32_t* t_ptr, const int32_t* t_key, const float* t_val,
                     const uint32_t* t_pk, const void* dYT, long ldt, float* dW, long ldw, void* stream);
/* The same weight gradient written as bf16 [N][ldw] (data-parallel training in precision mode 'bf16': the kernel writes
 * straight into the exchange buffer that the all-reduce sums and scp_adamw_multi then reads, g_bf16 = 1) */
int scp_spmm_dw_bf16_g16(int ncells, int N, int F, const int32_t* t_ptr, const int32_t* t_key, const float* t_val,
                         const uint32_t* t_pk, const void* dYT, long ldt, void* dW_bf16, long ldw, void* stream);
/* The same weight gradient with the AdamW step of that weight (model.py:698-701, torch.optim.AdamW semantics as in
 * scp_adamw_multi) applied in the kernel's epilogue: W [N][ldw] fp32 master weights, exp_avg / exp_avg_sq (same
 * layout) and the bf16 shadow are updated in place and the gradient never reaches HBM.  For single-process training
 * (no gradient exchange between backward and the optimiser step).  Only columns 0 .. F-1 of W are touched (ldw > F: a
 * covariate block's weights or row padding behind them).  Needs ldw % 4 == 0, ld_bf16 % 4 == 0 and 16-byte aligned
 * W / moments (SCP_ERR_SHAPE otherwise: use scp_spmm_dw_bf16 + scp_adamw_multi); rows that start on 128-byte boundaries
 * (ldw % 32 == 0) stream at the full rate, others re-read the lines that feature tiles share. */
int scp_spmm_dw_adamw_bf16(int ncells, int N, int F, const int32_t* t_ptr, const int32_t* t_key, const float* t_val,
                           const uint32_t* t_pk, const void* dYT, long ldt, float* W, long ldw, float* exp_avg, float* exp_avg_sq,
                           void* W_bf16, long ld_bf16, float lr, float weight_decay, float beta1, float beta2, float eps,
                           float bias_corr1, float bias_corr2, float grad_scale,
                           const float* bc_dev /* nullable: device {bias_corr1, bias_corr2} used instead of the values */,
                           void* stream);

/* ---- dense layers: nn.Linear forward / dgrad / wgrad (utils.py:57; model.py:84, 92, 163, 217, 228,
 *      280, 331; pairedVAE.py:49-50, 111-119) as one strided GEMM  C[M,N] (+)= A[M,K] B[N,K]^T + bias ---- */
int scp_gemm_nt(int M, int N, int K, const float* A, long a_row_stride, long a_k_stride, const float* B,
                long b_row_stride, long b_k_stride, float* C, long ldc, const float* bias /* [N] nullable */,
                int accumulate, int nsplit, int ns, void* stream);
/* The wide layers (ZINB G x G model.py:228, VAE 900 <-> G pairedVAE.py:111-119) on the hand-written MFMA kernel
 * (csrc/wgemm.hip), replacing torch's addmm behind nn.Linear
 * (model.py:228,255-256; pairedVAE.py:111-119) forward, input gradient and weight gradient:
 *   C[m][n] = sum_k A(m,k) B(n,k) (+ bias[n]) (set to zero where mask[m][n] <= 0: the backward of a relu on the input)
 * bf16 operands, fp32 accumulate/output.  a_col = 0: A is stored [M][lda] (k contiguous); 1: stored [K][lda] (m
 * contiguous); b_col likewise for B over n.  Leading dimensions are multiples of 8 >= the column count rounded up to 8,
 * pointers 16-byte aligned, the whole [rows][ld] extent readable, and the pad columns of a k-contiguous operand zero
 * (scp_cast_bf16 and the AdamW shadow write them so).  Shapes that do not fill the CUs evenly run as
 * scp_gemm_bf16_nsplit K-ranges into the caller's workspace ws (scp_gemm_bf16_ws_floats floats, 0 when not needed),
 * added in a fixed order: results are bitwise reproducible.  bias / mask may be null. */
int scp_gemm_bf16_nsplit(int M, int N, int K);
long scp_gemm_bf16_ws_floats(int M, int N, int K, long ldc);
int scp_gemm_bf16(int a_col, int b_col, int M, int N, int K, const void* A_bf16, long lda, const void* B_bf16, long ldb,
                  float* C, long ldc, const float* bias, const void* mask_bf16, long ldm, float* ws, void* stream);
/* Weight gradient of a wide nn.Linear, dW[n][j] = sum_m dy[m][n] x[m][j] (A = dy stored [K][lda], B = x stored [K][ldb],
 * both "col" operands of scp_gemm_bf16; here M = out features, N = in features, K = batch), with the AdamW step of that
 * weight (torch.optim.AdamW semantics as in scp_adamw_multi; reference utils.py:160-176 optimizer.step()) applied in the
 * kernel's epilogue: W [M][ldw] fp32 master weights, exp_avg / exp_avg_sq (same layout) and the bf16 shadow are updated
 * in place and the gradient never reaches HBM.  Single-process training only.  Needs N % 4 == 0, ldw % 4 == 0,
 * ld_bf16 % 4 == 0 and 16-byte aligned W / moments (SCP_ERR_SHAPE otherwise: use scp_gemm_bf16 + scp_adamw_multi). */
int scp_gemm_bf16_adamw(int M, int N, int K, const void* A_bf16, long lda, const void* B_bf16, long ldb, float* W, long ldw,
                        float* exp_avg, float* exp_avg_sq, void* W_bf16, long ld_bf16, float lr, float weight_decay,
                        float beta1, float beta2, float eps, float bias_corr1, float bias_corr2, float grad_scale,
                        const float* bc_dev /* nullable */, void* stream);
int scp_slab_reduce(const float* slabs, long slab_stride, int nslab, int rows, int cols, long ld,
                    const float* bias, float* out, long ldo, void* stream);
int scp_colsum(const float* X, long ld, int rows, int cols, float* out, float* partials_ws /* [32*cols] */,
               void* stream);

/* scp_colsum without a caller workspace and without any allocation.  No counterpart in the reference: it is for callers
 * that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  The
 * [32][cols] partial sums live in a device array inside the library (2 MiB per GPU, held until the process ends), so
 * cols <= scp_colsum_noalloc_max_cols() (16384; SCP_ERR_SHAPE beyond).  Calls that overlap on one GPU race on it. */
int scp_colsum_noalloc_max_cols(void);
int scp_colsum_noalloc(const float* X, long ld, int rows, int cols, float* out, void* stream);

/* ---- FCNN layer epilogue: [LayerNorm] -> LeakyReLU(slope)/ReLU(0) -> Dropout (utils.py:58-63),
 *      fused with the split-K slab reduction and the bias add; gamma/beta null = no LayerNorm ---- */
int scp_ln_act_fwd(const float* Y, long ldy, long slab_stride, int nslab, const float* bias, const float* gamma,
                   const float* beta, int B, int H, float eps, float slope, float p_drop, uint64_t seed,
                   const uint64_t* seed_dev /* nullable: device word added to seed (scp_step_state_set) */,
                   float* xhat /* [B][ldx] saved for backward, nullable */, long ldx, float* rstd /* [B] */,
                   float* out, long ldo, void* stream);
/* BatchNorm1d (placed after the activation, utils.py:61-62) + the layer's Dropout: batch statistics (training:
 * biased variance for the output, unbiased for the running estimate, momentum update in place) or running statistics
 * (eval); save_mean / save_rstd [H] feed the backward */
int scp_bn_fwd(const float* x, long ldx, int B, int H, const float* gamma, const float* beta, float* run_mean,
               float* run_var, int training, float momentum, float eps, float p_drop, uint64_t seed,
               const uint64_t* seed_dev, float* save_mean,
               float* save_rstd, float* y, long ldy, void* stream);
int scp_bn_bwd(const float* dy, long lddy, const float* x, long ldx, int B, int H, const float* gamma,
               const float* save_mean, const float* save_rstd, int training, float p_drop, uint64_t seed,
               const uint64_t* seed_dev, float* dx,
               long lddx, float* dgamma, float* dbeta, void* stream);
int scp_ln_act_bwd_grid(int B);   /* partials_ws needs grid*3*H floats */
int scp_ln_act_bwd(const float* dOut, long ldo, const float* xhat, long ldx, const float* rstd, const float* gamma,
                   const float* beta, int B, int H, float slope, float p_drop, uint64_t seed, const uint64_t* seed_dev,
                   float* dY, long ldy,
                   float* dgamma, float* dbeta, float* dbias /* each [H], nullable */, float* partials_ws,
                   void* stream);

/* ---- encoder tail: what Input_Module (model.py:60-143) does between the wide first-layer product and its output,
 *      i.e. the rest of a two-layer FCNN (utils.py:54-64: + bias, LayerNorm, LeakyReLU/ReLU, Dropout; Linear(H1, H2),
 *      LayerNorm, activation, Dropout) and the Linear behind it (extra_linear_rep model.py:83-84, or the library head
 *      Linear(H2, 1) model.py:92; N3 = 0: none), as one forward call and one backward call for up to
 *      SCP_ENC_TAIL_MAX_BRANCHES branches of one encoder (hidden_rep and hidden_lib, model.py:124-143).  Exact fp32.
 *      Needs H1 % 4 == 0, H1 <= 1024, H2 <= 64, N3 <= 64, LayerNorm on both layers, 16-byte aligned H1-wide tensors;
 *      the branches of a call share B.
 *      fwd: slabs [nslab][B][ld_slab] = split-K partials of x W1^T (scp_spmm_fwd*, scp_gemm_nt; a branch may be a
 *           column block of a wider product: ld_slab >= H1); out [B][N3] (N3 = 0: [B][H2]); xhat1 [B][H1], rstd1 [B],
 *           xhat2 [B][H2], rstd2 [B] are written for the backward (all null: inference).
 *      bwd: dOut [B][N3 or H2] -> the gradient of the first-layer product as fp32 dY1 [B][ldy] and / or as the
 *           transposed bf16 dYT [H1][ldt] scp_spmm_dw_*bf16 takes (ldt >= B rounded up to 64; either may be null), and
 *           the gradients of every tensor of the tail (each nullable); ws: scp_enc_tail_ws_floats(B, H1, H2, N3, C) floats. ---- */
#define SCP_ENC_TAIL_MAX_BRANCHES 4
typedef struct {
  int B, H1, H2, N3;
  const float *b1, *g1, *be1;          /* first layer: bias (nullable), LayerNorm weight / bias   [H1] */
  const float *W2, *b2, *g2, *be2;     /* second layer: weight [H2][H1], bias, LayerNorm weight / bias [H2] */
  const float *W3, *b3;                /* final Linear: weight [N3][H2], bias [N3] */
  float eps1, eps2, slope, p1, p2;     /* LayerNorm eps, activation slope, dropout probabilities */
  uint64_t seed1, seed2;               /* dropout seeds of the two layers */
  const uint64_t* seed_dev;            /* nullable: device word added to both seeds (scp_step_state_set) */
  float *xhat1, *rstd1, *xhat2, *rstd2;
  /* forward */
  const float* slabs; long slab_stride, ld_slab; int nslab;
  float* out; long ldo;
  /* backward */
  const float* dOut; long lddo;
  float* dY1; long ldy;
  void* dYT; long ldt;
  float* ws;
  float *dg1, *dbe1, *db1, *dW2, *dg2, *dbe2, *db2, *dW3, *db3;
  /* covariate block appended to the encoder input (model.py:128: torch.cat([input, input_batch])): cov [B][ldcov], C <= 64
   * columns (0: none); W1c = the first-layer weight's columns F .. F+C-1 (pointer W1 + F, row stride ldw1c = F + C).
   * backward: the gradient of those columns goes to dW1c [H1][lddw1c] (the same columns of the layer's gradient;
   * nullable), or -- adam != 0, single-process training with the first layer's AdamW step fused into its weight-gradient
   * kernel -- is consumed by the AdamW step of W1c in place (adam_m / adam_v / adam_sh point at the same columns of the
   * moments and of the bf16 shadow; hyper-parameters as in scp_spmm_dw_adamw_bf16) */
  const float* cov; long ldcov; int C;
  const float* W1c; long ldw1c;
  float* dW1c; long lddw1c;
  int adam; float *adam_m, *adam_v; void* adam_sh; long adam_ldsh;
  float lr, wd, beta1, beta2, adam_eps, bc1, bc2, gscale; const float* bc_dev;
  /* BatchNorm1d behind each activation (FCNN built with batchnorm=True, utils.py:61-62: Linear -> LayerNorm -> act ->
   * BatchNorm1d -> Dropout); bn = 0: none.  bn_train != 0: statistics of the batch (biased variance), running mean / var
   * updated in place with momentum bn_mom (unbiased variance), as nn.BatchNorm1d in training mode; else the running
   * statistics normalise.  Affine weight / bias and the running tensors are nullable (affine=False,
   * track_running_stats=False).  Batch statistics need the whole batch between the activation and the dropout of a
   * layer, so the calls run the tail in three segments with a column-statistics kernel between them (forward) and
   * three segments with a column-sum kernel between them (backward); the segments hand over through xhat1 / xhat2
   * (required whenever bn_train != 0, also without a backward) and through bn_dy1 [B][H1].
   * bn_ws: scp_enc_tail_bn_ws_floats(H1, H2) floats (scale, shift, mean, rstd and the backward's column sums of both
   * layers), kept between the forward and the backward call of a step.  All branches of a call share bn / bn_train. */
  int bn, bn_train;
  const float *bn1_w, *bn1_b; float *bn1_rm, *bn1_rv;      /* [H1] */
  const float *bn2_w, *bn2_b; float *bn2_rm, *bn2_rv;      /* [H2] */
  float bn_eps1, bn_eps2, bn_mom1, bn_mom2;
  float *bn_ws, *bn_dy1;
  float *dbn1_w, *dbn1_b, *dbn2_w, *dbn2_b;                /* backward: gradients of the affine parameters (nullable) */
} ScpEncTail;
int scp_enc_tail_fwd(const ScpEncTail* branches, int nbranch, void* stream);
long scp_enc_tail_ws_floats(int B, int H1, int H2, int N3, int C);
long scp_enc_tail_bn_ws_floats(int H1, int H2);
int scp_enc_tail_bwd(const ScpEncTail* branches, int nbranch, void* stream);

/* ---- Bernoulli decoder + BCE, fused forward(+backward): Output_Module_Ber.forward (model.py:164-190)
 *      + binarized_loss (loss.py:36-41, BCELoss log clamp -100, NaN labels masked) ---- */
int scp_dec_chunks(int F);
int scp_dec_kp(int K);
long scp_dec_ws_floats(int B, int F, int K);
long scp_dec_ws_doubles(int B, int F);
int scp_ber_dec_bce(int train, int B, int F, int K, const float* h, long ldh, const float* W, long ldw,
                    const float* bias, const float* fs /* feature_scalar, nullable */,
                    const float* lrow /* lib or latent_lib logits, nullable */, int lrow_n /* 0, 1 or B */,
                    const int32_t* t_ptr, const int32_t* t_key, const float* t_val, const int32_t* nnan,
                    const float* gout /* nullable = 1 */, float* loss /* [1] */, float* dW, long lddw, float* db,
                    float* dfs, float* dh, long lddh, float* dlrow, float* ws_f, double* ws_d, int ns, void* stream);

/* ---- NB decoder + log-likelihood, fused: Output_Module_Raw.forward without zero inflation
 *      (model.py:229-259) + raw_loss('nb') (loss.py:46-48) + log_nb_positive (scVI_distribution.py:74-105) ---- */
int scp_nb_dec_ll(int train, int B, int F, int K, const float* h, long ldh, const float* W, long ldw,
                  const float* bias, const float* fs, const float* theta_param /* dispersion='feature', nullable */,
                  const float* lrow /* [B] library size L */, const float* lat /* latent_lib or null */,
                  const int32_t* t_ptr, const int32_t* t_key, const float* t_val, const float* gout, float* loss,
                  float* dW, long lddw, float* db, float* dfs, float* dtheta, float* dh, long lddh, float* dlat,
                  float* rowbuf /* [2*B] */, float* ws_f, double* ws_d, int ns, void* stream);

/* ---- masked MSE: scaled_loss (loss.py:22-27) ---- */
int scp_mse_masked_fwd(const float* x, long ldx, const float* mu, long ldm, int rows, int cols,
                       float* loss_cnt /* [2]: mean, count */, double* ws /* >= 520 */, void* stream);
int scp_mse_masked_bwd(const float* x, long ldx, const float* mu, long ldm, int rows, int cols,
                       const float* loss_cnt, const float* gout, float* dmu, long ldd, void* stream);

/* ---- stand-alone likelihoods on dense tensors (the reference's own call form) ----
 *      binarized_loss(x, p) (loss.py:36-41); log_nb_positive / log_zinb_positive element-wise
 *      (scVI_distribution.py:29-105; pi == NULL selects NB; theta_vec != 0: theta is a [cols] vector) ---- */
int scp_bce_masked_fwd(const float* x, const float* p, long n, float* loss_cnt /* [2] */, double* ws /* >= 520 */,
                       void* stream);
int scp_bce_masked_bwd(const float* x, const float* p, long n, const float* loss_cnt, const float* gout, float* dp,
                       void* stream);
int scp_nbll_fwd(const float* x, const float* mu, const float* theta, int theta_vec, int cols, const float* pi,
                 long n, float* out, void* stream);
int scp_nbll_bwd(const float* x, const float* mu, const float* theta, int theta_vec, int cols, const float* pi,
                 const float* g, long n, float* dmu, float* dtheta, float* dpi, void* stream);

/* ---- VAE bottleneck: softplus + 1e-7, reparameterise (pairedVAE.py:51-58), KL mean (utils.py:186) ---- */
int scp_reparam_kl_fwd(const float* mu, const float* sraw, const float* eps, long n, float* sigma, float* z,
                       float* kl /* [1] */, double* ws /* >= 520 */, void* stream);
int scp_reparam_kl_bwd(const float* mu, const float* sraw, const float* eps, const float* sigma, const float* dz,
                       const float* dmu_in, const float* gkl, long n, float* dmu, float* dsraw, void* stream);

/* ---- decoder heads for predict() (model.py:558-591): Output_Module_Ber.forward's sigmoid * sample factor * feature
 *      factor (model.py:173-189) and Output_Module_Raw.forward's relu -> [feature factor] -> row softmax, exp dispersion
 *      (model.py:238-252) on the logits z.  out_bf16 != 0: `out` / `scale` are bf16 [rows][ld] (halves the bytes that
 *      predict()'s device-to-host copy moves; the reference returns fp32) ---- */
int scp_ber_prob(const float* z, long ldz, int rows, int cols, const float* lrow, int lrow_n, const float* fs,
                 void* out, long ldo, int out_bf16, void* stream);
int scp_relu_softmax_rows(const float* z, long ldz, int rows, int cols, const float* fs, int apply_relu,
                          void* scale, long lds, float* theta /* nullable, fp32 */, long ldt, int out_bf16, void* stream);

/* ---- (ZI)NB decoder heads on materialised pre-activations (loss.py:46-51; scVI_distribution.py:29-105), no feature
 *      factor.  z [B][G] = output of the mean Linear, pi [B][G] zero-inflation logits or null (plain NB), y dense
 *      labels, L [B] library term (observed library size, exp(LeakyReLU(latent_lib)), or ones).
 *        tz == null: the ZINB predictor with its G x G zero-inflation layer (model.py:228-259, dispersion
 *                    'feature-cell'): a = relu(z), scale = softmax(a), theta = exp(a);
 *        tz [B][G] : the VAE decoder heads (pairedVAE.py:147-187): scale = softmax(z), theta = softplus(tz) + 1e-8.
 *      fast != 0: hardware exp / log / reciprocal (precision mode 'bf16').
 *      fwd: loss, lse [B], T [B] (kept for bwd), part: B doubles.  bwd: dz, dtz (with tz), dpi (with pi), dL (nullable) ---- */
int scp_raw_rows_fwd(const float* z, long ldz, const float* tz, long ldt, const float* pi, long ldp, const float* y,
                     long ldy, const float* L, int B, int G, int fast, float* lse, float* T, double* part, float* loss,
                     void* stream);
int scp_raw_rows_bwd(const float* z, long ldz, const float* tz, long ldt, const float* pi, long ldp, const float* y,
                     long ldy, const float* L, const float* lse, const float* T, const float* gout, int B, int G, int fast,
                     float* dz, long lddz, float* dtz, long lddt, float* dpi, long lddp, float* dL, void* stream);

/* The predictor's head with per-feature parameters (model.py:239-252): fs [G] = feature_scalar of feature_factor=True
 * (scale = softmax(relu(z) * sigmoid(fs))), thp [G] = output_inverse_dispersion of dispersion='feature'
 * (theta = exp(thp) instead of exp(relu(z))); either may be null.  bwd also writes dfs [G] / dthp [G] (nullable): column
 * sums over the cells, taken over fixed chunks of 16 rows into ws (scp_raw_rows_feat_ws_floats floats) and added in
 * chunk order (bitwise reproducible). */
long scp_raw_rows_feat_ws_floats(int B, int G);
int scp_raw_rows_feat_fwd(const float* z, long ldz, const float* pi, long ldp, const float* y, long ldy, const float* L,
                          const float* fs, const float* thp, int B, int G, int fast, float* lse, float* T, double* part,
                          float* loss, void* stream);
int scp_raw_rows_feat_bwd(const float* z, long ldz, const float* pi, long ldp, const float* y, long ldy, const float* L,
                          const float* fs, const float* thp, const float* lse, const float* T, const float* gout, int B,
                          int G, int fast, float* dz, long lddz, float* dpi, long lddp, float* dL, float* dfs, float* dthp,
                          float* ws, void* stream);

/* ---- evaluation metrics (evaluator.py:14-71): neighbour counts behind FOSCTTM / FOSFTTM for paired embeddings
 *      x, y [n][d]: lt12[i] = #{j: D[i][j] < D[i][i]}, lt21[i] = #{j: D[j][i] < D[i][i]},
 *      gt12[i] = #{j: D[i][j] > D[i][i]} with D the direct-form squared distances; diag_ws: n floats ---- */
int scp_fos_counts(const float* x, long ldx, const float* y, long ldy, int n, int d, float* diag_ws, int32_t* lt12,
                   int32_t* lt21, int32_t* gt12, void* stream);

/* ---- optimiser: torch.optim.AdamW step as the trainer configures it (model.py:698-701, 849-850, 982) ---- */
#define SCP_ADAM_MAX_TENSORS 24
typedef struct {
  float* p;
  const float* g;
  float* m;
  float* v;
  long n;
  float lr, weight_decay, bias_corr1, bias_corr2, grad_scale;
  void* p_bf16;      /* optional bf16 shadow of p laid out [n / cols][ld_bf16], rewritten with the new weights */
  long cols, ld_bf16;
  long g_bf16;       /* != 0: g points to bf16 gradients (n contiguous elements: the data-parallel exchange buffer
                        scp_spmm_dw_bf16_g16 writes and the all-reduce sums) instead of fp32 */
  const float* bc_dev; /* nullable: device words {bias_corr1, bias_corr2} read instead of the two values above */
  long ld_p;         /* 0 or == cols: p, m, v are n contiguous elements; > cols: they are [n / cols][ld_p] (rows padded to
                        128-byte boundaries) while g stays n contiguous elements */
} ScpAdamTensor;
int scp_adamw_multi(const ScpAdamTensor* tensors /* host array */, int n, float beta1, float beta2, float eps,
                    void* stream);
/* Per-step state of a training step captured in a HIP graph (no counterpart in the reference, whose step is eager:
 * utils.py:118-148).  A graph bakes by-value kernel arguments; what changes per step -- the dropout seed and AdamW's
 * bias corrections 1 - beta^t -- is read from 16 bytes of device memory {uint64 seed_offset; float bc1; float bc2}
 * (the seed_dev / bc_dev arguments above point into it) that this call rewrites before each replay. */
int scp_step_state_set(void* state, uint64_t seed_offset, float bias_corr1, float bias_corr2, void* stream);

#ifdef __cplusplus
}
#endif
#endif
