// Wide dense GEMM of the plain nn.Linear layers whose two dimensions are both large: the ZINB zero-inflation layer
// G x G (reference model.py:228,255-256) and the VAE's 900 <-> G layers (pairedVAE.py:111-119), precision mode 'bf16'.
//
//   C[m][n] = sum_k A(m,k) B(n,k)  (+ bias[n])  (zero where mask[m][n] <= 0)       bf16 operands, fp32 accumulate / output
//
// Each operand is either stored "row" ([rows][ld], k contiguous) or "col" ([K][ld], rows contiguous), so the three
// products of a Linear layer run on the operands as they lie in memory, without a transposed copy:
//   y  = x W^T   : A = x  row, B = W row        dx = dy W : A = dy row, B = W col        dW = dy^T x : A = dy col, B = x col
//
// Workgroup: tile 256 x 256 x 64 on 512 threads = 2 x 4 waves of 128 x 64 (4 x 2 accumulators of the 32x32x16 bf16
// MFMA, two waves per SIMD) or on 256 threads = 2 x 2 waves of 128 x 128 (one wave per SIMD).  Operand tiles go
// global -> registers (16-byte buffer loads, the range check supplies the zeros of ragged edges) -> LDS, two LDS
// stages of 64 KB, one barrier per K-step: the next step's global loads are in flight while this step's 64 MFMAs run.
//   row image  [256][64]  : 128-byte rows, 16-byte chunk c of row r at chunk c ^ ((r >> 1) & 7); read with ds_read_b128
//   col image  [64][256]  : 512-byte rows, 64-byte segment s of row k at segment s ^ (k & 3); read with the
//                           transposing ds_read_b64_tr_b16 (4 k-rows x 16 columns per 16 lanes), conflict-free
// Work split: tiles x nsplit K-ranges so that the launch fills the 256 CUs evenly (1024 x 20 k is 316 tiles = 1.23
// rounds; four K-ranges make it 4.94 rounds of a quarter each); the K-ranges write fp32 slabs which a second kernel
// adds in a fixed order, together with bias and mask (bitwise reproducible: no atomics).  Consecutive workgroup ids
// go round-robin over the 8 XCDs, so the id is remapped to give each XCD a contiguous run of tiles (the tiles that
// share a B tile sit next to each other in one L2).
#include <stdlib.h>
#include "common.h"
#include "scpair_hip.h"

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int STAGE_BYTES = (TM + TN) * TK * 2;   // 64 KB

struct WgArgs {
  const __bf16* A;
  const __bf16* B;
  float* C;            // output, or slabs [nsplit][M][ldc] when nsplit > 1
  const float* bias;   // direct mode only
  const __bf16* mask;  // direct mode only
  long lda, ldb, ldc, ldm;
  int M, N, K;
  int tiles_m, tiles_n, nsplit, ksteps_per_split;
  unsigned a_bytes, b_bytes;
  unsigned c_bytes, m_bytes;   // extent of C (one slab) and of the mask when both are below 2 GiB, else 0 (guarded global stores)
  // weight gradient with the AdamW step applied in the epilogue (EPI_ADAMW): C = fp32 master weights, updated in place
  float* adam_m; float* adam_v; __bf16* adam_sh; long adam_ldsh;
  float lr, wd, b1, b2, eps, bc1, bc2, gscale;
  const float* bc_dev;       // optional device words {bias_corr1, bias_corr2} (HIP-graph replays), else bc1 / bc2
  unsigned dbg_oor;    // experiments only (SCPAIR_WGEMM_DBG=1): every load in the K loop out of range -> no memory traffic
};

typedef __attribute__((ext_vector_type(4))) short short4_;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ u32x4 ld128(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 0);
}

// LDS byte offsets inside an operand image
// (rows r and r + 1 lie in the two halves of a 256-byte bank row, so the swizzle runs over r >> 1: the 16 lanes that
// a ds_read_b128 serves together then touch 16 distinct 16-byte slots)
__device__ __forceinline__ int row_off(int r, int ch) { return r * 128 + ((ch ^ ((r >> 1) & 7)) << 4); }
__device__ __forceinline__ int col_off(int k, int m) { return k * 512 + ((((m >> 5) ^ (k & 3))) << 6) + ((m & 31) << 1); }

template <bool COL>
__device__ __forceinline__ bf16x8 read_frag(const unsigned char* img, int r0, int s, int lane) {
  if (!COL) {
    const int r = r0 + (lane & 31), ch = 2 * s + (lane >> 5);
    return *(const bf16x8*)(img + row_off(r, ch));
  } else {
    // lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a 4 x 16 block; lane i receives column i
    const int g = lane >> 5, mb = ((lane >> 4) & 1) * 16, q = (lane & 15) >> 2, p = lane & 3;
    const int k = 16 * s + 8 * g + q, m = r0 + mb + 4 * p;
    typedef __attribute__((address_space(3))) short4_ lds_s4;
    short4_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(img + col_off(k, m)));
    short4_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(img + col_off(k + 4, m)));
    union { short4_ h[2]; bf16x8 v; } u;
    u.h[0] = lo;
    u.h[1] = hi;
    return u.v;
  }
}

enum { EPI_SLAB = 0, EPI_DIRECT = 1, EPI_ADAMW = 2 };

// NW = 8 (default for the plain products): two waves per SIMD, wave tile 128 x 64, one register set of prefetched tiles
// (256 registers per lane): one wave's MFMA groups cover the other's waits.  NW = 4 (the AdamW form; SCPAIR_WGEMM_NW=4
// elsewhere): one wave per SIMD, wave tile 128 x 128, two register sets (loads two K-steps ahead).
template <bool A_COL, bool B_COL, int EPI, int NW = 4>
__global__ __launch_bounds__(64 * NW) void wgemm_kernel(WgArgs g) {
  constexpr bool DIRECT = EPI == EPI_DIRECT;
  constexpr int NJ = NW == 4 ? 4 : 2, WN = 32 * NJ, CH = 32 / NW, LSTEP = 1024 * NW, NSET = NW == 4 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = NW == 4 ? wave >> 1 : wave >> 2, wn = NW == 4 ? wave & 1 : wave & 3;

  // XCD-aware id: ids b, b+8, b+16.. run on one XCD -> give them consecutive tiles
  const int total = g.tiles_m * g.tiles_n * g.nsplit;
  const int per_xcd = (total + 7) >> 3;
  const int lid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (lid >= total) return;
  const int tm = lid % g.tiles_m;
  const int tn = (lid / g.tiles_m) % g.tiles_n;
  const int z = lid / (g.tiles_m * g.tiles_n);
  const int m0 = tm * TM, n0 = tn * TN;
  const int nk_all = (g.K + TK - 1) / TK;
  const int kt0 = z * g.ksteps_per_split;
  const int kt1 = min(nk_all, kt0 + g.ksteps_per_split);

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)g.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, (int)g.b_bytes, 0x00020000);
  constexpr unsigned OOR = 0x80000000u;     // beyond any operand (host checks bytes < 2 GiB): the load returns zeros

  // chunk c = tid + 256 i.  row image: row (tid >> 3) + 32 i, chunk tid & 7; col image: k row (tid >> 5) + 8 i,
  // chunk tid & 31: the global offset and the LDS offset of chunk i are those of chunk 0 plus i times a constant
  const int Kp = (g.K + 7) & ~7;
  const int rA = A_COL ? (tid >> 5) : (tid >> 3), cA = A_COL ? (tid & 31) : (tid & 7);
  const int rB = B_COL ? (tid >> 5) : (tid >> 3), cB = B_COL ? (tid & 31) : (tid & 7);
  const unsigned baseA = A_COL ? (unsigned)(((long)rA * g.lda + m0 + 8 * cA) * 2) : (unsigned)(((long)(m0 + rA) * g.lda + 8 * cA) * 2);
  const unsigned baseB = B_COL ? (unsigned)(((long)rB * g.ldb + n0 + 8 * cB) * 2) : (unsigned)(((long)(n0 + rB) * g.ldb + 8 * cB) * 2);
  const unsigned stepA = (unsigned)((A_COL ? 8 : 32) * (NW / 4) * g.lda * 2), stepB = (unsigned)((B_COL ? 8 : 32) * (NW / 4) * g.ldb * 2);
  const int ldsA0 = A_COL ? col_off(rA, 8 * cA) : row_off(rA, cA);
  const int ldsB0 = B_COL ? col_off(rB, 8 * cB) : row_off(rB, cB);
  // Ragged edges come from the buffer's range check: a row image's rows >= M and a col image's k rows >= K start at
  // or beyond the operand's last byte and load as zeros.  Only the k tail of a row image needs a test (it would run
  // into the next row): those chunks get the top offset bit.  A col image's columns beyond ld wrap into the next k
  // row; they only feed output rows >= M, which are not stored.
  unsigned offA[CH], offB[CH];
#pragma unroll
  for (int i = 0; i < CH; i++) {
    offA[i] = baseA + stepA * i;
    offB[i] = baseB + stepB * i;
  }

  // two register sets: the loads of K-step kt + 2 are issued during step kt and stored to LDS during step kt + 1,
  // so a load has more than a whole K-step (64 MFMAs) to arrive
  u32x4 ra[NSET][CH], rb[NSET][CH];
  auto load_tiles = [&](int kt, int set) {       // kt >= kt1: out of range, zeros
    const int k0 = kt * TK;
    const unsigned dead = (kt < kt1 ? 0u : OOR) | g.dbg_oor;
    const unsigned kA = (A_COL ? (unsigned)((long)k0 * g.lda * 2) : ((k0 + 8 * cA < Kp) ? (unsigned)k0 * 2u : OOR)) | dead;
    const unsigned kB = (B_COL ? (unsigned)((long)k0 * g.ldb * 2) : ((k0 + 8 * cB < Kp) ? (unsigned)k0 * 2u : OOR)) | dead;
#pragma unroll
    for (int i = 0; i < CH; i++) {
      ra[set][i] = ld128(rsA, offA[i] + kA);
      rb[set][i] = ld128(rsB, offB[i] + kB);
    }
  };
  auto store_tiles = [&](int stage, int set) {
    unsigned char* ia = smem + stage * STAGE_BYTES;
    unsigned char* ib = ia + TM * TK * 2;
#pragma unroll
    for (int i = 0; i < CH; i++) {
      *(u32x4*)(ia + ldsA0 + LSTEP * i) = ra[set][i];
      *(u32x4*)(ib + ldsB0 + LSTEP * i) = rb[set][i];
    }
  };

  f32x16 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  bf16x8 fa[2][4], fb[2][NJ];
  auto read_frags = [&](int stage, int s, int set) {
    const unsigned char* ia = smem + stage * STAGE_BYTES;
    const unsigned char* ib = ia + TM * TK * 2;
#pragma unroll
    for (int i = 0; i < 4; i++) fa[set][i] = read_frag<A_COL>(ia, wm * 128 + 32 * i, s, lane);
#pragma unroll
    for (int j = 0; j < NJ; j++) fb[set][j] = read_frag<B_COL>(ib, wn * WN + 32 * j, s, lane);
  };
  auto products = [&](int set) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < NJ; j++) acc[i][j] = mfma32(fa[set][i], fb[set][j], acc[i][j]);
  };
  // instruction classes of __builtin_amdgcn_sched_group_barrier
  constexpr int MFMA = 0x008, VMEM_RD = 0x020, DS_RD = 0x100, DS_WR = 0x200;
  constexpr int RD_A = A_COL ? 2 : 1, RD_B = B_COL ? 2 : 1;     // LDS reads per fragment

  load_tiles(kt0, 0);
  store_tiles(0, 0);
  if (NSET == 2) load_tiles(kt0 + 1, 1);
  __syncthreads();
  // One K-step = four MFMA groups of 4 x NJ products.  The fragments of group s + 1 are read while group s multiplies,
  // the global loads of a later step (kt + 2 with two register sets, kt + 1 with one) are spread over group 0 and the
  // LDS stores of step kt + 1's tiles over group 3 (the other stage: nobody reads it before the barrier), so the
  // matrix pipe always has the next group's operands in registers.  Two steps per trip (static register sets); an
  // odd range runs one step on zero tiles.
#define WG_SCHED(VM)                                                                 \
  _Pragma("unroll") for (int t = 0; t < 4; t++) {                                    \
    if (VM) __builtin_amdgcn_sched_group_barrier(VMEM_RD, 2, 0);                     \
    __builtin_amdgcn_sched_group_barrier(DS_RD, RD_A, 0);                            \
    __builtin_amdgcn_sched_group_barrier(MFMA, NJ == 4 ? 2 : 1, 0);                  \
  }                                                                                  \
  _Pragma("unroll") for (int t = 0; t < NJ; t++) {                                   \
    if (VM && NJ == 4) __builtin_amdgcn_sched_group_barrier(VMEM_RD, 2, 0);          \
    __builtin_amdgcn_sched_group_barrier(DS_RD, RD_B, 0);                            \
    __builtin_amdgcn_sched_group_barrier(MFMA, 2, 0);                                \
  }                                                                                  \
  __builtin_amdgcn_sched_barrier(0);
  auto k_step = [&](int kt, int stage, int set_ld, int set_st) {
    read_frags(stage, 0, 0);
    load_tiles(kt + NSET, set_ld);
    read_frags(stage, 1, 1);
    products(0);
    WG_SCHED(true)
    read_frags(stage, 2, 0);
    products(1);
    WG_SCHED(false)
    read_frags(stage, 3, 1);
    products(0);
    WG_SCHED(false)
    store_tiles(stage ^ 1, set_st);
    products(1);
#pragma unroll
    for (int t = 0; t < 2 * CH; t++) {
      __builtin_amdgcn_sched_group_barrier(DS_WR, 1, 0);
      __builtin_amdgcn_sched_group_barrier(MFMA, 1, 0);
    }
    __syncthreads();
  };
#undef WG_SCHED
  for (int kt = kt0; kt < kt1; kt += 2) {
    k_step(kt, 0, 0, NSET - 1);
    k_step(kt + 1, 1, NSET - 1, 0);
  }

  // epilogue: accumulator element e of lane l is row acc_row(e, l), column l & 31 of its 32 x 32 tile
  if constexpr (EPI == EPI_ADAMW) {
    static_assert(NW == 4, "the AdamW epilogue is written for the four-wave tile");
    // The tile is a block of the weight gradient: apply torch.optim.AdamW's update to the master weights, the moments
    // and the bf16 shadow right here; the gradient never reaches HBM.  Every wave passes its 32-row strips through a
    // private LDS buffer [32][132] (the operand stages are free after the loop's last barrier; same-wave LDS accesses
    // need no barrier) and walks them row-wise: half a wave per 512-byte row segment, float4 loads and stores.
    typedef float f4 __attribute__((ext_vector_type(4)));
    float* wt = (float*)smem + wave * (32 * 132);
    const float bc1 = g.bc_dev ? g.bc_dev[0] : g.bc1, bc2 = g.bc_dev ? g.bc_dev[1] : g.bc2;
    const AdamHyper hy = adam_hyper(g.lr, g.wd, g.b1, g.b2, g.eps, bc1, bc2, g.gscale);
    const int c4 = (lane & 31) * 4;
    const int n = n0 + wn * 128 + c4;
    const bool col_ok = n + 4 <= g.N;             // N % 4 == 0 (host-checked): a float4 is all in or all out
    const bool interior = m0 + TM <= g.M && n0 + TN <= g.N;
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) wt[acc_row(e, lane) * 132 + 32 * j + (lane & 31)] = acc[i][j][e];
      __builtin_amdgcn_wave_barrier();
      const int mrow = m0 + wm * 128 + 32 * i + (lane >> 5);
      if (interior) {
        // no tests: the loads of four row pairs are issued before the first update, twelve float4 in flight per lane
#pragma unroll
        for (int it0 = 0; it0 < 16; it0 += 4) {
          f4 pv[4], mv[4], vv[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const long idx = (long)(mrow + 2 * (it0 + u)) * g.ldc + n;
            // read once and written once per step, 10 GB in all: streaming accesses, nothing to keep in the caches
            pv[u] = __builtin_nontemporal_load((const f4*)(g.C + idx));
            mv[u] = __builtin_nontemporal_load((const f4*)(g.adam_m + idx));
            vv[u] = __builtin_nontemporal_load((const f4*)(g.adam_v + idx));
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int row = 2 * (it0 + u) + (lane >> 5);
            const f4 gv = *(const f4*)(wt + row * 132 + c4);
            const long idx = (long)(mrow + 2 * (it0 + u)) * g.ldc + n;
            bf16x4 sh4;
#pragma unroll
            for (int t = 0; t < 4; t++) {
              float p_ = pv[u][t], m_ = mv[u][t], v_ = vv[u][t];
              adamw_update_fast(hy, p_, gv[t], m_, v_);
              pv[u][t] = p_;
              mv[u][t] = m_;
              vv[u][t] = v_;
              sh4[t] = (__bf16)p_;
            }
            __builtin_nontemporal_store(pv[u], (f4*)(g.C + idx));
            __builtin_nontemporal_store(mv[u], (f4*)(g.adam_m + idx));
            __builtin_nontemporal_store(vv[u], (f4*)(g.adam_v + idx));
            *(bf16x4*)(g.adam_sh + (long)(mrow + 2 * (it0 + u)) * g.adam_ldsh + n) = sh4;
          }
        }
      } else {
#pragma unroll 2
        for (int it = 0; it < 16; it++) {
          const int row = 2 * it + (lane >> 5);
          const f4 gv = *(const f4*)(wt + row * 132 + c4);
          const int m = m0 + wm * 128 + 32 * i + row;
          if (col_ok && m < g.M) {
            const long idx = (long)m * g.ldc + n;
            f4 pv = *(const f4*)(g.C + idx), mv = *(const f4*)(g.adam_m + idx), vv = *(const f4*)(g.adam_v + idx);
            bf16x4 sh4;
#pragma unroll
            for (int t = 0; t < 4; t++) {
              float p_ = pv[t], m_ = mv[t], v_ = vv[t];
              adamw_update_fast(hy, p_, gv[t], m_, v_);
              pv[t] = p_;
              mv[t] = m_;
              vv[t] = v_;
              sh4[t] = (__bf16)p_;
            }
            *(f4*)(g.C + idx) = pv;
            *(f4*)(g.adam_m + idx) = mv;
            *(f4*)(g.adam_v + idx) = vv;
            *(bf16x4*)(g.adam_sh + (long)m * g.adam_ldsh + n) = sh4;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  float* Cz = g.C + (DIRECT ? 0 : (long)z * g.M * g.ldc);
  if (m0 + TM <= g.M && n0 + TN <= g.N && !(DIRECT && g.mask)) {
    // interior tile: no tests; one store instruction writes two 128-byte row segments
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const int n = n0 + wn * WN + 32 * j + (lane & 31);
      const float bn = (DIRECT && g.bias) ? g.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        float* p = Cz + (long)(m0 + wm * 128 + 32 * i + 4 * (lane >> 5)) * g.ldc + n;
#pragma unroll
        for (int e = 0; e < 16; e++) p[(long)((e & 3) + 8 * (e >> 2)) * g.ldc] = acc[i][j][e] + bn;
      }
    }
    return;
  }
  if (g.c_bytes) {
    // edge tile or masked output, C (and mask) below 2 GiB: buffer stores, no branches.  Rows >= M start at or beyond
    // the buffer's last byte, so the range check drops them; columns >= N get the top offset bit.
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)Cz, 0, (int)g.c_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsM =
        __builtin_amdgcn_make_buffer_rsrc((void*)(DIRECT && g.mask ? g.mask : g.A), 0, (int)g.m_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const int n = n0 + wn * WN + 32 * j + (lane & 31);
      const unsigned colbad = n < g.N ? 0u : OOR;
      const float bn = (DIRECT && g.bias && n < g.N) ? g.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int mb = m0 + wm * 128 + 32 * i + 4 * (lane >> 5);
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int r = (e & 3) + 8 * (e >> 2);
          // The out-of-range mark REPLACES the offset (a top bit OR-ed into a base would be carried out again by the
          // row term when C ends just below 2 GiB); offsets are formed in 64 bits and compared with the extent.
          const long co = ((long)(mb + r) * g.ldc + n) * 4, mo = ((long)(mb + r) * g.ldm + n) * 2;
          const unsigned coff = (colbad || co >= (long)g.c_bytes) ? OOR : (unsigned)co;
          float v = acc[i][j][e] + bn;
          if (DIRECT && g.mask) {
            const unsigned moff = (colbad || mo >= (long)g.m_bytes) ? OOR : (unsigned)mo;
            const unsigned short mk = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsM, (int)moff, 0, 0);
            const float mf = __uint_as_float((unsigned)mk << 16);
            if (!(mf > 0.f)) v = 0.f;
          }
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsC, (int)coff, 0, 0);
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const int n = n0 + wn * WN + 32 * j + (lane & 31);
    if (n >= g.N) continue;
    const float bn = (DIRECT && g.bias) ? g.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int m = m0 + wm * 128 + 32 * i + acc_row(e, lane);
        if (m < g.M) {
          float v = acc[i][j][e] + bn;
          if (DIRECT && g.mask && !((float)g.mask[(long)m * g.ldm + n] > 0.f)) v = 0.f;
          Cz[(long)m * g.ldc + n] = v;
        }
      }
    }
  }
}

// C = sum of the K-range slabs in slab order, + bias, masked
__global__ __launch_bounds__(256) void wgemm_reduce_kernel(const float* __restrict__ slabs, int nsplit, int M, int N, long ldc,
                                                            const float* __restrict__ bias, const __bf16* __restrict__ mask,
                                                            long ldm, float* __restrict__ C) {
  const long n4 = (N + 3) / 4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)M * n4) return;
  const int m = (int)(i / n4), c = (int)(i % n4) * 4;
  const long slab = (long)M * ldc;
  const float* p = slabs + (long)m * ldc + c;
  if (c + 3 < N && (ldc & 3) == 0) {
    f32x4 s = *(const f32x4*)p;
    for (int zz = 1; zz < nsplit; zz++) s += *(const f32x4*)(p + zz * slab);
#pragma unroll
    for (int t = 0; t < 4; t++) {
      if (bias) s[t] += bias[c + t];
      if (mask && !((float)mask[(long)m * ldm + c + t] > 0.f)) s[t] = 0.f;
    }
    *(f32x4*)(C + (long)m * ldc + c) = s;
  } else {
    for (int t = 0; t < 4 && c + t < N; t++) {
      float s = p[t];
      for (int zz = 1; zz < nsplit; zz++) s += p[t + zz * slab];
      if (bias) s += bias[c + t];
      if (mask && !((float)mask[(long)m * ldm + c + t] > 0.f)) s = 0.f;
      C[(long)m * ldc + c + t] = s;
    }
  }
}

// K-ranges so that tiles x ranges fill the 256 CUs evenly; a function of the shape only (results do not depend on
// the machine's state).  Cost model in units of one full-K tile: rounds / nsplit for the product, plus the slab
// traffic of the reduction priced against the tile's MFMA time.
int pick_nsplit(int M, int N, int K) {
  if (const char* e = getenv("SCPAIR_WGEMM_NSPLIT")) {      // experiments only
    const int v = atoi(e);
    if (v >= 1 && v <= 16 && v <= (K + TK - 1) / TK) return v;
  }
  const long tiles = (long)((M + TM - 1) / TM) * ((N + TN - 1) / TN);
  const int nk = (K + TK - 1) / TK;
  const double tile_s = 2.0 * TM * TN * (double)K / (2.5e15 / 256 * 0.5);       // one tile at half of the MFMA peak
  double best = 1e30;
  int best_ns = 1;
  for (int ns = 1; ns <= 16; ns++) {
    if (ns > 1 && nk / ns < 8) break;
    const double rounds = (double)((tiles * ns + 255) / 256) / ns;
    double cost = rounds * tile_s;
    if (ns > 1) cost += (double)(ns + 2) * M * N * 4.0 / 4.0e12 + 4e-6;
    if (cost < best * 0.97) {
      best = cost;
      best_ns = ns;
    }
  }
  return best_ns;
}

template <bool AC, bool BC>
int launch(const WgArgs& g, hipStream_t st) {
  static uint64_t done_d = 0, done_s = 0;
  const int total = g.tiles_m * g.tiles_n * g.nsplit;
  const unsigned grid = (unsigned)(((total + 7) / 8) * 8);
  static uint64_t done_d8 = 0, done_s8 = 0;
  // eight waves by default (measured 2-10 % faster than four on every shape of the wide layers, same box, back to
  // back); SCPAIR_WGEMM_NW=4 selects the four-wave form for comparison
  const char* e8 = getenv("SCPAIR_WGEMM_NW");
  const bool nw8 = !(e8 && atoi(e8) == 4);
  if (g.nsplit == 1 && !nw8) {
    int rc = scp_set_smem((const void*)wgemm_kernel<AC, BC, EPI_DIRECT>, 2 * STAGE_BYTES, &done_d);
    if (rc) return rc;
    hipLaunchKernelGGL((wgemm_kernel<AC, BC, EPI_DIRECT>), dim3(grid), dim3(256), 2 * STAGE_BYTES, st, g);
  } else if (!nw8) {
    int rc = scp_set_smem((const void*)wgemm_kernel<AC, BC, EPI_SLAB>, 2 * STAGE_BYTES, &done_s);
    if (rc) return rc;
    hipLaunchKernelGGL((wgemm_kernel<AC, BC, EPI_SLAB>), dim3(grid), dim3(256), 2 * STAGE_BYTES, st, g);
  } else if (g.nsplit == 1) {
    int rc = scp_set_smem((const void*)wgemm_kernel<AC, BC, EPI_DIRECT, 8>, 2 * STAGE_BYTES, &done_d8);
    if (rc) return rc;
    hipLaunchKernelGGL((wgemm_kernel<AC, BC, EPI_DIRECT, 8>), dim3(grid), dim3(512), 2 * STAGE_BYTES, st, g);
  } else {
    int rc = scp_set_smem((const void*)wgemm_kernel<AC, BC, EPI_SLAB, 8>, 2 * STAGE_BYTES, &done_s8);
    if (rc) return rc;
    hipLaunchKernelGGL((wgemm_kernel<AC, BC, EPI_SLAB, 8>), dim3(grid), dim3(512), 2 * STAGE_BYTES, st, g);
  }
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

}  // namespace

extern "C" int scp_gemm_bf16_nsplit(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 1;
  return pick_nsplit(M, N, K);
}

extern "C" long scp_gemm_bf16_ws_floats(int M, int N, int K, long ldc) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int ns = pick_nsplit(M, N, K);
  return ns > 1 ? (long)ns * M * ldc : 0;
}

// dW = A^T-style product of two k-major operands with the AdamW step of W in the epilogue (see the header)
extern "C" int scp_gemm_bf16_adamw(int M, int N, int K, const void* A, long lda, const void* B, long ldb, float* W, long ldw,
                                   float* exp_avg, float* exp_avg_sq, void* W_bf16, long ld_bf16, float lr, float weight_decay,
                                   float beta1, float beta2, float eps, float bias_corr1, float bias_corr2, float grad_scale,
                                   const float* bc_dev, void* stream_) {
  SCP_CHECK(M > 0 && N > 0 && K > 0, SCP_ERR_ARG, "scp_gemm_bf16_adamw: M, N, K must be positive (%d, %d, %d)", M, N, K);
  SCP_CHECK(A && B && W && exp_avg && exp_avg_sq && W_bf16, SCP_ERR_ARG, "scp_gemm_bf16_adamw: null pointer");
  SCP_CHECK(lda >= ((M + 7) & ~7L) && ldb >= ((N + 7) & ~7L) && ldw >= N && ld_bf16 >= N, SCP_ERR_ARG,
            "scp_gemm_bf16_adamw: leading dimension too small");
  SCP_CHECK(lda % 8 == 0 && ldb % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, SCP_ERR_ARG,
            "scp_gemm_bf16_adamw: operands must be 16-byte aligned with leading dimensions that are multiples of 8");
  SCP_CHECK(((long)K + 3 * TK) * lda * 2 + 4096 < (1L << 31) && ((long)K + 3 * TK) * ldb * 2 + 4096 < (1L << 31), SCP_ERR_SHAPE,
            "scp_gemm_bf16_adamw: an operand within 3 K-steps of 2 GiB (or larger) is not supported (see scp_gemm_bf16)");
  SCP_CHECK(N % 4 == 0 && ldw % 4 == 0 && ld_bf16 % 4 == 0 &&
                (((uintptr_t)W | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16) == 0 && ((uintptr_t)W_bf16 % 8) == 0,
            SCP_ERR_SHAPE, "scp_gemm_bf16_adamw: needs N %% 4 == 0, leading dimensions %% 4 == 0 and 16-byte aligned W / moments");
  SCP_CHECK(bc_dev || (bias_corr1 > 0.f && bias_corr2 > 0.f), SCP_ERR_ARG, "scp_gemm_bf16_adamw: bias corrections must be > 0");
  WgArgs g = {};
  g.A = (const __bf16*)A;
  g.B = (const __bf16*)B;
  g.C = W;
  g.lda = lda;
  g.ldb = ldb;
  g.ldc = ldw;
  g.M = M;
  g.N = N;
  g.K = K;
  g.tiles_m = (M + TM - 1) / TM;
  g.tiles_n = (N + TN - 1) / TN;
  g.nsplit = 1;                                   // the update needs the complete sum: one K-range
  g.ksteps_per_split = (K + TK - 1) / TK;
  g.a_bytes = (unsigned)((long)K * lda * 2);
  g.b_bytes = (unsigned)((long)K * ldb * 2);
  g.adam_m = exp_avg; g.adam_v = exp_avg_sq; g.adam_sh = (__bf16*)W_bf16; g.adam_ldsh = ld_bf16;
  g.lr = lr; g.wd = weight_decay; g.b1 = beta1; g.b2 = beta2; g.eps = eps; g.bc1 = bias_corr1; g.bc2 = bias_corr2;
  g.gscale = grad_scale; g.bc_dev = bc_dev;
  SCP_CHECK((long)g.tiles_m * g.tiles_n < (1L << 30), SCP_ERR_SHAPE, "scp_gemm_bf16_adamw: too many tiles");
  static uint64_t done_a = 0;
  int rc = scp_set_smem((const void*)wgemm_kernel<true, true, EPI_ADAMW>, 2 * STAGE_BYTES, &done_a);
  if (rc) return rc;
  const int total = g.tiles_m * g.tiles_n;
  hipLaunchKernelGGL((wgemm_kernel<true, true, EPI_ADAMW>), dim3((unsigned)(((total + 7) / 8) * 8)), dim3(256), 2 * STAGE_BYTES,
                     (hipStream_t)stream_, g);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

extern "C" int scp_gemm_bf16(int a_col, int b_col, int M, int N, int K, const void* A, long lda, const void* B, long ldb,
                             float* C, long ldc, const float* bias, const void* mask, long ldm, float* ws, void* stream_) {
  SCP_CHECK(M > 0 && N > 0 && K > 0, SCP_ERR_ARG, "scp_gemm_bf16: M, N, K must be positive (%d, %d, %d)", M, N, K);
  SCP_CHECK(A && B && C, SCP_ERR_ARG, "scp_gemm_bf16: null pointer");
  SCP_CHECK((a_col == 0 || a_col == 1) && (b_col == 0 || b_col == 1), SCP_ERR_ARG, "scp_gemm_bf16: a_col / b_col must be 0 or 1");
  const long a_rows = a_col ? K : M, a_cols = a_col ? M : K, b_rows = b_col ? K : N, b_cols = b_col ? N : K;
  SCP_CHECK(lda >= ((a_cols + 7) & ~7L) && ldb >= ((b_cols + 7) & ~7L) && ldc >= N, SCP_ERR_ARG,
            "scp_gemm_bf16: leading dimension too small (operands are padded to a multiple of 8 columns)");
  SCP_CHECK(lda % 8 == 0 && ldb % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, SCP_ERR_ARG,
            "scp_gemm_bf16: operands must be 16-byte aligned with leading dimensions that are multiples of 8");
  // The K loop marks dead loads by OR-ing the top offset bit into the per-step term and then adds the per-chunk offset:
  // the sum must stay below 2^31 for every (step <= last + 2, chunk), or the mark would be carried out and real data
  // read.  Hence the slack of two K-steps (k-major operand: whole k rows) on top of the operand's own extent.
  SCP_CHECK((a_rows + (a_col ? 3 * TK : 0)) * lda * 2 + 4096 < (1L << 31) &&
                (b_rows + (b_col ? 3 * TK : 0)) * ldb * 2 + 4096 < (1L << 31), SCP_ERR_SHAPE,
            "scp_gemm_bf16: an operand within 3 K-steps of 2 GiB (or larger) is not supported");
  SCP_CHECK(!mask || ldm >= N, SCP_ERR_ARG, "scp_gemm_bf16: mask leading dimension too small");
  WgArgs g = {};
  g.A = (const __bf16*)A;
  g.B = (const __bf16*)B;
  g.lda = lda;
  g.ldb = ldb;
  g.ldc = ldc;
  g.ldm = ldm;
  g.M = M;
  g.N = N;
  g.K = K;
  g.tiles_m = (M + TM - 1) / TM;
  g.tiles_n = (N + TN - 1) / TN;
  g.nsplit = pick_nsplit(M, N, K);
  const int nk = (K + TK - 1) / TK;
  g.ksteps_per_split = (nk + g.nsplit - 1) / g.nsplit;
  g.a_bytes = (unsigned)(a_rows * lda * 2);
  g.b_bytes = (unsigned)(b_rows * ldb * 2);
  g.dbg_oor = (getenv("SCPAIR_WGEMM_DBG") && atoi(getenv("SCPAIR_WGEMM_DBG")) == 1) ? 0x80000000u : 0u;
  SCP_CHECK((long)g.tiles_m * g.tiles_n * g.nsplit < (1L << 30), SCP_ERR_SHAPE, "scp_gemm_bf16: too many tiles");
  {
    const long cb = (long)M * ldc * 4, mb = mask ? (long)M * ldm * 2 : 2;
    const bool small = cb < (1L << 31) && mb < (1L << 31);
    g.c_bytes = small ? (unsigned)cb : 0u;
    g.m_bytes = small ? (unsigned)mb : 0u;
  }
  hipStream_t st = (hipStream_t)stream_;
  if (g.nsplit == 1) {
    g.C = C;
    g.bias = bias;
    g.mask = (const __bf16*)mask;
  } else {
    SCP_CHECK(ws, SCP_ERR_ARG, "scp_gemm_bf16: this shape runs as %d K-ranges and needs the workspace of scp_gemm_bf16_ws_floats",
              g.nsplit);
    g.C = ws;
    g.bias = nullptr;
    g.mask = nullptr;
  }
  int rc = a_col ? (b_col ? launch<true, true>(g, st) : launch<true, false>(g, st))
                 : (b_col ? launch<false, true>(g, st) : launch<false, false>(g, st));
  if (rc) return rc;
  if (g.nsplit > 1) {
    const long n = (long)M * ((N + 3) / 4);
    hipLaunchKernelGGL(wgemm_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ws, g.nsplit, M, N, ldc, bias,
                       (const __bf16*)mask, ldm, C);
    SCP_LAUNCH_CHECK();
  }
  return SCP_OK;
}
