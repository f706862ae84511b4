// bf16-MFMA GEMM core with pluggable operand loaders (gfx950).
//
//   C[M,N] (+)= A[M,K] * B[N,K]^T (+ bias[N])          fp32 in HBM, bf16 MFMA, fp32 accumulate
//
// Operand kinds
//   dense  : fp32 in HBM with arbitrary (row, k) strides, converted to bf16 while staging into LDS
//   sparse : a cell-block tile list (see tiles.hip) densified INTO LDS -- the sparse counts are read
//            straight from their CSR-derived entry list in HBM, never materialised densely in HBM
//              A sparse : A[m = cell   ][k = feature]   first encoder layer forward  (model.py:128-130)
//              B sparse : B[n = feature][k = cell   ]   first encoder layer dW = dY^T X
//
// Precision: template NS = number of bf16 terms each fp32 operand is split into.
//   NS=1 plain bf16 inputs;  NS=3 hi+mid+lo split, 6 MFMAs per product, fp32-equivalent result.
//
// Tiling: 128x128x64 per 256-thread workgroup, 2x2 waves, each wave 2x2 MFMA 32x32x16 tiles.
// LDS rows are padded to 72 bf16 (144 B) so the 16-B fragment reads are bank-conflict free.
// Workgroup ids are mapped so that the m-tiles sharing one (n-tile, k-slice) of the weight/sparse
// operand run on the same XCD (blocks b and b+8 share an XCD) and hit in its L2.
#include "common.h"
#include "scpair_hip.h"

#define TM 128
#define TN 128
#define TK 64
#define LDK 72

struct GemmArgs {
  int M, N, K;
  // dense A: element (m,k) = A[m*a_sr + k*a_sk]
  const float* A;
  long a_sr, a_sk;
  int a_vec;
  // dense B: element (n,k) = B[n*b_sr + k*b_sk]
  const float* B;
  long b_sr, b_sk;
  int b_vec;
  // sparse operand (A or B)
  const int32_t* sp_ptr;
  const int32_t* sp_key;
  const float* sp_val;
  int sp_F;     // feature count of the tile list
  int sp_ncb;   // number of 64-cell blocks
  // output
  float* C;
  long ldc;
  const float* bias;
  int accumulate;
  int nsplit, ksplit_len;   // ksplit_len multiple of TK
  int tiles_m, tiles_n;
};

template <int NS>
__device__ __forceinline__ void store_split8(__bf16* S, int row, int kcol, const float* v) {
  __bf16 t[8][3];
#pragma unroll
  for (int j = 0; j < 8; j++) bf16_split<NS>(v[j], t[j]);
#pragma unroll
  for (int s = 0; s < NS; s++) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = t[j][s];
    *(bf16x8*)&S[(s * TM + row) * LDK + kcol] = o;
  }
}

// Dense operand staging is split in two so the global loads of a whole 128 x 64 tile (4 chunks of 8 values per
// thread) are in flight together and, for the next K-step, overlap the MFMAs of the current one.
__device__ __forceinline__ void fetch_dense_tile(float (&v)[4][8], const float* __restrict__ p, long sr, long sk, int vec,
                                                 int r0, int rows, int k0, int kend) {
#pragma unroll
  for (int it = 0; it < 4; it++) {
    int c = threadIdx.x + 256 * it;
    int row, kc;
    if (sk == 1) {
      row = c >> 3;
      kc = c & 7;
    } else {
      row = c & (TM - 1);
      kc = c >> 7;
    }
    int gr = r0 + row, gk = k0 + kc * 8;
    if (gr < rows && gk < kend) {
      const float* q = p + (long)gr * sr + (long)gk * sk;
      if (vec && gk + 8 <= kend) {
        float4 x = *(const float4*)q, y = *(const float4*)(q + 4);
        v[it][0] = x.x; v[it][1] = x.y; v[it][2] = x.z; v[it][3] = x.w;
        v[it][4] = y.x; v[it][5] = y.y; v[it][6] = y.z; v[it][7] = y.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++) v[it][j] = (gk + j < kend) ? q[(long)j * sk] : 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) v[it][j] = 0.f;
    }
  }
}

template <int NS>
__device__ __forceinline__ void put_dense_tile(__bf16* S, const float (&v)[4][8], long sk) {
#pragma unroll
  for (int it = 0; it < 4; it++) {
    int c = threadIdx.x + 256 * it;
    int row, kc;
    if (sk == 1) {
      row = c >> 3;
      kc = c & 7;
    } else {
      row = c & (TM - 1);
      kc = c >> 7;
    }
    store_split8<NS>(S, row, kc * 8, v[it]);
  }
}

template <int NS>
__device__ __forceinline__ void zero_tile(__bf16* S) {
  uint4 z = {0u, 0u, 0u, 0u};
  uint4* q = (uint4*)S;
  for (int i = threadIdx.x; i < NS * TM * LDK * 2 / 16; i += 256) q[i] = z;
}

// A[m = cell][k = feature]; tile rows = 2 cell blocks of 64, tile k range = features [k0, kend)
template <int NS>
__device__ __forceinline__ void scatter_sparse_A(__bf16* S, const GemmArgs& g, int m0, int k0, int kend) {
  for (int half = 0; half < 2; half++) {
    int cb = (m0 >> 6) + half;
    if (cb >= g.sp_ncb) break;
    long base = (long)cb * g.sp_F;
    int p0 = g.sp_ptr[base + k0], p1 = g.sp_ptr[base + kend];
    for (int e = p0 + threadIdx.x; e < p1; e += 256) {
      int key = g.sp_key[e];
      float v = g.sp_val ? g.sp_val[e] : 1.f;
      int col = key >> 6, rl = key & 63;
      __bf16 t[3];
      bf16_split<NS>(v, t);
#pragma unroll
      for (int s = 0; s < NS; s++) S[(s * TM + half * 64 + rl) * LDK + (col - k0)] = t[s];
    }
  }
}

// B[n = feature][k = cell]; tile rows = features [n0, n0+128), tile k range = cell block k0/64
template <int NS>
__device__ __forceinline__ void scatter_sparse_B(__bf16* S, const GemmArgs& g, int n0, int k0) {
  int cb = k0 >> 6;
  long base = (long)cb * g.sp_F;
  int nend = min(n0 + TN, g.sp_F);
  int p0 = g.sp_ptr[base + n0], p1 = g.sp_ptr[base + nend];
  for (int e = p0 + threadIdx.x; e < p1; e += 256) {
    int key = g.sp_key[e];
    float v = g.sp_val ? g.sp_val[e] : 1.f;
    int col = key >> 6, rl = key & 63;
    __bf16 t[3];
    bf16_split<NS>(v, t);
#pragma unroll
    for (int s = 0; s < NS; s++) S[(s * TM + (col - n0)) * LDK + rl] = t[s];
  }
}

template <int NS, int AKIND, int BKIND>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* As = (__bf16*)smem_raw;
  __bf16* Bs = As + NS * TM * LDK;

  // XCD-aware id map: the tiles_m blocks that share one (n-tile, k-slice) sit on one XCD
  int b = blockIdx.x;
  int xcd = b & 7, j = b >> 3;
  int m_tile = j % g.tiles_m;
  int unit = (j / g.tiles_m) * 8 + xcd;
  if (unit >= g.tiles_n * g.nsplit) return;
  int n_tile = unit % g.tiles_n;
  int split = unit / g.tiles_n;
  int m0 = m_tile * TM, n0 = n_tile * TN;
  int kbeg = split * g.ksplit_len;
  int kend_all = min(g.K, kbeg + g.ksplit_len);

  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int wm = wave >> 1, wn = wave & 1;
  int r = lane & 31, hq = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int jj = 0; jj < 2; jj++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][jj][e] = 0.f;

  float ra[AKIND == 0 ? 4 : 1][8], rb[BKIND == 0 ? 4 : 1][8];
  auto fetch = [&](int k0) {
    int kend = min(k0 + TK, kend_all);
    if constexpr (AKIND == 0) fetch_dense_tile(ra, g.A, g.a_sr, g.a_sk, g.a_vec, m0, g.M, k0, kend);
    if constexpr (BKIND == 0) fetch_dense_tile(rb, g.B, g.b_sr, g.b_sk, g.b_vec, n0, g.N, k0, kend);
  };
  if (kbeg < kend_all) fetch(kbeg);
  for (int k0 = kbeg; k0 < kend_all; k0 += TK) {
    int kend = min(k0 + TK, kend_all);
    if constexpr (AKIND == 0)
      put_dense_tile<NS>(As, ra, g.a_sk);
    else
      zero_tile<NS>(As);
    if constexpr (BKIND == 0)
      put_dense_tile<NS>(Bs, rb, g.b_sk);
    else
      zero_tile<NS>(Bs);
    if (AKIND == 1 || BKIND == 1) {
      __syncthreads();
      if (AKIND == 1) scatter_sparse_A<NS>(As, g, m0, k0, kend);
      if (BKIND == 1) scatter_sparse_B<NS>(Bs, g, n0, k0);
    }
    __syncthreads();
    if (k0 + TK < kend_all) fetch(k0 + TK);       // in flight during the MFMAs below
#pragma unroll
    for (int kk = 0; kk < TK / 16; kk++) {
      bf16x8 af[NS][2], bfr[NS][2];
#pragma unroll
      for (int s = 0; s < NS; s++) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
          af[s][i] = *(const bf16x8*)&As[(s * TM + wm * 64 + i * 32 + r) * LDK + kk * 16 + hq * 8];
          bfr[s][i] = *(const bf16x8*)&Bs[(s * TM + wn * 64 + i * 32 + r) * LDK + kk * 16 + hq * 8];
        }
      }
#pragma unroll
      for (int sa = 0; sa < NS; sa++)
#pragma unroll
        for (int sb = 0; sb < NS; sb++)
          if (sa + sb < NS) {
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
              for (int jj = 0; jj < 2; jj++) acc[i][jj] = mfma32(af[sa][i], bfr[sb][jj], acc[i][jj]);
          }
    }
    __syncthreads();
  }

  float* C = g.C + (long)split * g.M * g.ldc;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int jj = 0; jj < 2; jj++) {
      int n = n0 + wn * 64 + jj * 32 + r;
      if (n >= g.N) continue;
      float bv = (g.bias && split == 0) ? g.bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; e++) {
        int m = m0 + wm * 64 + i * 32 + acc_row(e, lane);
        if (m < g.M) {
          float* dst = C + (long)m * g.ldc + n;
          float v = acc[i][jj][e] + bv;
          if (g.accumulate) v += *dst;
          *dst = v;
        }
      }
    }
}

template <int NS, int AK, int BK_>
static int launch_gemm(const GemmArgs& g, hipStream_t stream) {
  static uint64_t attr_done = 0;
  size_t smem = (size_t)NS * 2 * TM * LDK * sizeof(__bf16);
  int rc_ = scp_set_smem((const void*)gemm_kernel<NS, AK, BK_>, smem, &attr_done);
  if (rc_) return rc_;
  int units = g.tiles_n * g.nsplit;
  int grid = 8 * g.tiles_m * ((units + 7) / 8);
  hipLaunchKernelGGL((gemm_kernel<NS, AK, BK_>), dim3(grid), dim3(256), smem, stream, g);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

template <int AK, int BK_>
static int dispatch_ns(const GemmArgs& g, int ns, hipStream_t stream) {
  switch (ns) {
    case 1: return launch_gemm<1, AK, BK_>(g, stream);
    case 3: return launch_gemm<3, AK, BK_>(g, stream);
  }
  scp_set_error("ns must be 1 or 3 (got %d)", ns);
  return SCP_ERR_ARG;
}

static void fill_common(GemmArgs& g, int M, int N, int K, float* C, long ldc, const float* bias, int accumulate,
                        int nsplit) {
  g.M = M; g.N = N; g.K = K;
  g.C = C; g.ldc = ldc; g.bias = bias; g.accumulate = accumulate;
  g.tiles_m = (M + TM - 1) / TM;
  g.tiles_n = (N + TN - 1) / TN;
  if (nsplit < 1) nsplit = 1;
  int ksteps = (K + TK - 1) / TK;
  if (nsplit > ksteps) nsplit = ksteps > 0 ? ksteps : 1;
  int per = (ksteps + nsplit - 1) / nsplit;
  g.ksplit_len = per * TK;
  g.nsplit = (ksteps + per - 1) / per;
  if (g.nsplit < 1) g.nsplit = 1;
}

static int vec_ok(const float* p, long sr, long sk) { return sk == 1 && (sr % 4 == 0) && (((uintptr_t)p) % 16 == 0); }

extern "C" int scp_gemm_split_count(int K, int nsplit) {
  int ksteps = (K + TK - 1) / TK;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > ksteps) nsplit = ksteps > 0 ? ksteps : 1;
  int per = (ksteps + nsplit - 1) / nsplit;
  int n = (ksteps + per - 1) / per;
  return n < 1 ? 1 : n;
}

extern "C" int scp_gemm_nt(int M, int N, int K, const float* A, long a_sr, long a_sk, const float* B, long b_sr,
                           long b_sk, float* C, long ldc, const float* bias, int accumulate, int nsplit, int ns,
                           void* stream) {
  SCP_CHECK(M > 0 && N > 0 && K > 0, SCP_ERR_ARG, "scp_gemm_nt: M,N,K must be positive (%d,%d,%d)", M, N, K);
  SCP_CHECK(A && B && C, SCP_ERR_ARG, "scp_gemm_nt: null pointer");
  SCP_CHECK(!(accumulate && nsplit > 1), SCP_ERR_ARG, "scp_gemm_nt: accumulate with split-K slabs is not supported");
  GemmArgs g = {};
  fill_common(g, M, N, K, C, ldc, bias, accumulate, nsplit);
  g.A = A; g.a_sr = a_sr; g.a_sk = a_sk; g.a_vec = vec_ok(A, a_sr, a_sk);
  g.B = B; g.b_sr = b_sr; g.b_sk = b_sk; g.b_vec = vec_ok(B, b_sr, b_sk);
  return dispatch_ns<0, 0>(g, ns, (hipStream_t)stream);
}

// Y[ncells, N] (+slabs) = X_tiles[ncells, F] * W[N, F(+..)]^T ; W row stride w_sr (>= F), unit k stride
extern "C" int scp_spmm_fwd(int ncells, int N, int F, const int32_t* t_ptr, const int32_t* t_key, const float* t_val,
                            const float* W, long w_sr, float* Y, long ldy, const float* bias, int nsplit, int ns,
                            void* stream) {
  SCP_CHECK(ncells > 0 && N > 0 && F > 0, SCP_ERR_ARG, "scp_spmm_fwd: sizes must be positive");
  SCP_CHECK(t_ptr && t_key && W && Y, SCP_ERR_ARG, "scp_spmm_fwd: null pointer");
  SCP_CHECK(F < (1 << 25), SCP_ERR_SHAPE, "scp_spmm_fwd: F=%d exceeds the 2^25 key range", F);
  GemmArgs g = {};
  fill_common(g, ncells, N, F, Y, ldy, bias, 0, nsplit);
  g.sp_ptr = t_ptr; g.sp_key = t_key; g.sp_val = t_val; g.sp_F = F; g.sp_ncb = (ncells + 63) / 64;
  g.B = W; g.b_sr = w_sr; g.b_sk = 1; g.b_vec = vec_ok(W, w_sr, 1);
  return dispatch_ns<1, 0>(g, ns, (hipStream_t)stream);
}

// dW[N, F] = dY[ncells, N]^T * X_tiles[ncells, F]; dY row stride ldy (unit column stride); dW row stride ldw
extern "C" int scp_spmm_dw(int ncells, int N, int F, const int32_t* t_ptr, const int32_t* t_key, const float* t_val,
                           const float* dY, long ldy, float* dW, long ldw, int ns, void* stream) {
  SCP_CHECK(ncells > 0 && N > 0 && F > 0, SCP_ERR_ARG, "scp_spmm_dw: sizes must be positive");
  SCP_CHECK(t_ptr && t_key && dY && dW, SCP_ERR_ARG, "scp_spmm_dw: null pointer");
  SCP_CHECK(F < (1 << 25), SCP_ERR_SHAPE, "scp_spmm_dw: F=%d exceeds the 2^25 key range", F);
  GemmArgs g = {};
  // M = N(hidden), N = F(features), K = cells
  fill_common(g, N, F, ncells, dW, ldw, nullptr, 0, 1);
  g.A = dY; g.a_sr = 1; g.a_sk = ldy; g.a_vec = 0;
  g.sp_ptr = t_ptr; g.sp_key = t_key; g.sp_val = t_val; g.sp_F = F; g.sp_ncb = (ncells + 63) / 64;
  return dispatch_ns<0, 1>(g, ns, (hipStream_t)stream);
}
