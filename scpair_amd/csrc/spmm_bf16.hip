// First encoder layer on sparse counts, bf16 throughput path (precision mode 'bf16').
//
//   forward : Y[cells, H]  = X_tiles[cells, F] * Wbf[H, F]^T          (split-K slabs, fp32)
//   wgrad   : dW[H, F]     = dYT[H, cells] * X_tiles[cells, F]        (fp32 output)
//
// Differences from the generic kernel in gemm.hip (which stays the fp32-equivalent path):
//   * the weight operand is a bf16 SHADOW copy of the fp32 master weights (kept in sync by the fused AdamW
//     kernel), the dY operand a bf16 transposed copy: both are loaded with 16-byte vector loads and go to LDS
//     without conversion, halving the L2->LDS bytes and removing the per-element VALU work;
//   * 256-row tiles (512 threads, 8 waves): the streamed dense operand is re-read half as often;
//   * the next K-step's dense tile and sparse entries are prefetched into registers while the MFMAs of the
//     current step run.
// The sparse operand is densified into LDS from the minibatch tile list exactly as in gemm.hip.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "scpair_hip.h"

#define LDK 72
#define PCH 255 // forward: K-steps whose entry offsets are staged in LDS at a time
#define PCW 511 // weight gradient: cell blocks per call (entry ranges of a feature tile in every cell block, staged in LDS):
                // 32 768 cells, the global batch of 8 ranks x 4096 under feature-sharded data parallelism

struct SpArgs {
  int ncells, N, F;          // cells, hidden width, features
  const int32_t* ptr; const int32_t* key; const float* val;
  const uint32_t* pk;        // value-carrying data: packed entries (bf16 bits of the value) << 16 | (key & 0xffff), written by
                             // scp_tiles_build next to key / val: one load and one register per entry instead of two, and the
                             // value goes to LDS as it is (the unpacked form rounded it to bf16 at the same place)
  int ncb;
  const __bf16* D; long ldd; // dense bf16 operand: forward Wbf[N][ldd] ; wgrad dYT[N][ldd]
  float* C; long ldc;        // forward: slabs [nsplit][ncells][ldc] ; wgrad: dW[N][ldc]
  int nsplit, ksplit_len, tiles_m, tiles_n;
  // wgrad with the AdamW update applied in the epilogue (C = the fp32 master weights, updated in place)
  float* adam_m; float* adam_v; __bf16* adam_sh; long adam_ldsh;
  float lr, wd, b1, b2, eps, bc1, bc2, gscale;
  const float* bc_dev;       // optional device words {bias_corr1, bias_corr2} (HIP-graph replays), else bc1 / bc2
#ifdef SPMM_PROF
  unsigned long long* prof;  // diagnostic build only: per-phase timestamps of one wave of the forward kernel
#endif
};
#ifdef SPMM_PROF
#define SP_MARK(m) do { if (prof_on) g.prof[(long)i * 8 + (m)] = clock64(); } while (0)
static unsigned long long* sp_prof_dev = nullptr;
extern "C" int scp_spmm_prof_read(unsigned long long* out, int n) {
  if (!sp_prof_dev) return -1;
  hipDeviceSynchronize();
  return (int)hipMemcpy(out, sp_prof_dev, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost);
}
#else
#define SP_MARK(m) do { } while (0)
#endif

__device__ __forceinline__ uint4 ld16(const __bf16* p, bool ok) {
  uint4 z = {0u, 0u, 0u, 0u};
  return ok ? *(const uint4*)p : z;
}

// A load the compiler's memory-wait bookkeeping does not see: issued and awaited inside one asm statement.
// The K loop of the forward kernel keeps three K-steps of compiler-tracked loads in flight; the waitcnt pass turns
// any *visible* load on a rare side path (a loop of unknown trip count) into a full drain (vmcnt(0)) at the next
// control-flow merge, which would serialise the prefetch.  The rare overflow path therefore uses this helper.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t gload_b32_sync(const void* p) {
  uint32_t v;
  asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}

// ------------------------------------------------------------------------------------------------------
// forward: tile 256 cells x 128 hidden x 64 features, 8 waves as 4 (cells) x 2 (hidden)
//
// One barrier per K-step.  LDS holds three A (densified cells x features) buffers and two B (weights) buffers:
// while the MFMAs of step i read A[i%3] / B[i%2], the same step writes the operands of step i+1 (B tile from
// registers, sparse entries scattered into A[(i+1)%3]) and re-zeroes just the positions step i-1 had scattered
// into A[(i-1)%3] (its keys are kept in registers), so the A buffers are never cleared wholesale inside the loop.
// The global loads run three steps ahead of their use in registers (stage r holds steps r, r+3, ...).
// ------------------------------------------------------------------------------------------------------
template <bool HASVAL>
__global__ __launch_bounds__(512) void spmm_fwd_bf16_kernel(SpArgs g) {
  constexpr int EPT = HASVAL ? 6 : 2;             // sparse entries held per thread and stage (C2 RNA: up to 673 of 768 slots)
  constexpr int ASZ = 256 * LDK, BSZ = 128 * LDK;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* As = (__bf16*)smem_raw;        // [3][256][LDK]
  __bf16* Bs = As + 3 * ASZ;             // [2][128][LDK]
  int* ptrs = (int*)(Bs + 2 * BSZ);      // [4][PCH + 1] entry offsets of the next PCH K-steps, per cell block
  __shared__ int ovf[3];                 // step+1 of a step whose A buffer holds entries beyond the per-thread slots
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, hq = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  int b = blockIdx.x;
  int xcd = b & 7, j = b >> 3;
  int m_tile = j % g.tiles_m;
  int unit = (j / g.tiles_m) * 8 + xcd;
  if (unit >= g.tiles_n * g.nsplit) return;
  int n_tile = unit % g.tiles_n, split = unit / g.tiles_n;
  const int m0 = m_tile * 256, n0 = n_tile * 128;
  const int kbeg = split * g.ksplit_len;
  const int kend_all = min(g.F, kbeg + g.ksplit_len);
  const int nstep = (kend_all - kbeg + 63) >> 6;

  // sparse side: 4 cell blocks, 128 threads each
  const int cbl = tid >> 7, tl = tid & 127;
#ifdef SPMM_PROF
  const bool prof_on = (tid == 0) && (blockIdx.x == 64) && g.prof;
#endif

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int jj = 0; jj < 2; jj++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][jj][e] = 0.f;

  u32x4 breg[3][2];
  uint32_t ekey[3][EPT];                          // key words, or packed (value, key) words when HASVAL
  int p0[3], p1[3];
  int skey[3][EPT];                               // LDS offsets this thread scattered for step j (slot j%3), -1 = none
  const int brow = tid >> 3, bkc = tid & 7;       // B tile chunk of this thread (and +64 rows for the second one)

  auto prefetch = [&](u32x4 (&br)[2], uint32_t (&ek)[EPT], int& q0, int& q1,
                      int step) {
    const int k0 = kbeg + 64 * step;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      int n = min(n0 + brow + 64 * i, g.N - 1);
      long k = min((long)k0 + bkc * 8, g.ldd - 8);
      br[i] = *(const u32x4*)(g.D + (long)n * g.ldd + k);
    }
    {
      int si = step % PCH;
      q0 = ptrs[cbl * (PCH + 1) + si];
      q1 = ptrs[cbl * (PCH + 1) + si + 1];
    }
    const int elast = max(q1 - 1, 0);
#pragma unroll
    for (int i = 0; i < EPT; i++) {
      int e = min(q0 + tl + 128 * i, elast);
      if constexpr (HASVAL) ek[i] = g.pk[e];
      else ek[i] = (uint32_t)g.key[e];
    }
  };

  // entry offsets of steps [first, first + PCH] of this slice -> LDS (one global latency per PCH steps instead of a
  // dependent ptr -> key chain in every step)
  auto load_ptrs = [&](int first) {
    for (int i = tid; i < 4 * (PCH + 1); i += 512) {
      int c = i / (PCH + 1), sidx = i % (PCH + 1);
      int cbx = (m0 >> 6) + c;
      int k = min(kbeg + (first + sidx) * 64, kend_all);
      ptrs[i] = (cbx < g.ncb) ? g.ptr[(long)cbx * g.F + k] : 0;
    }
  };
  // registers of one stage -> LDS operands of `step`
  auto stage_out = [&](const u32x4 (&br)[2], const uint32_t (&ek)[EPT], int q0,
                       int q1, int step, int (&sk)[EPT]) {
    const int k0 = kbeg + 64 * step;
    __bf16* Ad = As + (step % 3) * ASZ;
    __bf16* Bd = Bs + (step & 1) * BSZ;
    const bool kin = (long)k0 + bkc * 8 < g.ldd;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      u32x4 v = br[i];
      if (!(kin && n0 + brow + 64 * i < g.N)) v = u32x4{0u, 0u, 0u, 0u};
      *(u32x4*)&Bd[(brow + 64 * i) * LDK + bkc * 8] = v;
    }
#pragma unroll
    for (int i = 0; i < EPT; i++) {
      sk[i] = -1;
      if (q0 + tl + 128 * i < q1) {
        const uint32_t w = ek[i];
        if constexpr (HASVAL) {                   // k0 is a multiple of 64: the low key bits are the position in the tile
          int off = (cbl * 64 + (w & 63)) * LDK + ((w >> 6) & 63);
          ((uint16_t*)Ad)[off] = (uint16_t)(w >> 16);
          sk[i] = off;
        } else {
          int off = (cbl * 64 + (w & 63)) * LDK + (((int)w >> 6) - k0);
          Ad[off] = (__bf16)1.f;
          sk[i] = off;
        }
      }
    }
    if (q0 + 128 * EPT < q1) {                    // overflow of a dense (cell block, K-step): rare, uniform per wave
      if (tl == 0) ovf[step % 3] = step + 1;
      for (int e = q0 + tl + 128 * EPT; e < q1; e += 128) {
        if constexpr (HASVAL) {
          const uint32_t w = gload_b32_sync(g.pk + e);
          ((uint16_t*)Ad)[(cbl * 64 + (w & 63)) * LDK + ((w >> 6) & 63)] = (uint16_t)(w >> 16);
        } else {
          int key = (int)gload_b32_sync(g.key + e);
          Ad[(cbl * 64 + (key & 63)) * LDK + ((key >> 6) - k0)] = (__bf16)1.f;
        }
      }
    }
  };
  auto zero_A = [&](int buf) {
    u32x4 z = {0u, 0u, 0u, 0u};
    u32x4* q = (u32x4*)(As + buf * ASZ);
    for (int i = tid; i < ASZ * 2 / 16; i += 512) q[i] = z;
  };
  // undo the scatter of `step` (whose MFMAs are done): its buffer is all zero again afterwards
  auto unscatter = [&](int step, const int (&sk)[EPT]) {
    __bf16* Ad = As + (step % 3) * ASZ;
    // the tag was written two barriers ago (when the step was staged) and the slot is next written one barrier
    // from now, so every thread reads the same value; a stale tag of step-3 never matches
    if (ovf[step % 3] == step + 1) {
      zero_A(step % 3);
    } else {
#pragma unroll
      for (int i = 0; i < EPT; i++)
        if (sk[i] >= 0) Ad[sk[i]] = (__bf16)0.f;
    }
  };
  if (nstep > 0) {
    load_ptrs(0);
    zero_A(0);
    zero_A(1);
    zero_A(2);
    if (tid < 3) ovf[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 3; s++)
      if (s < nstep) prefetch(breg[s], ekey[s], p0[s], p1[s], s);
    stage_out(breg[0], ekey[0], p0[0], p1[0], 0, skey[0]);
    if (3 < nstep) prefetch(breg[0], ekey[0], p0[0], p1[0], 3);
    __syncthreads();
  }
  // One K-step.  FULL = every neighbour the step touches exists (steps i-1 .. i+4), so the body is straight-line:
  // the hot loop below consists of FULL steps only, which lets the compiler wait for exactly the oldest of the
  // three load stages in flight instead of draining all of them.
  auto k_step = [&](auto full_c, auto s_c, int i) {
    constexpr bool FULL = decltype(full_c)::value;
    constexpr int s = decltype(s_c)::value;       // i % 3
    constexpr int s1 = (s + 1) % 3, s2 = (s + 2) % 3;
    const __bf16* Ar = As + s * ASZ;
    const __bf16* Br = Bs + (i & 1) * BSZ;
    SP_MARK(0);
    bf16x8 af[4][2], bfr[4][2];
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
#pragma unroll
      for (int q = 0; q < 2; q++) {
        af[kk][q] = *(const bf16x8*)&Ar[(wm * 64 + q * 32 + r) * LDK + kk * 16 + hq * 8];
        bfr[kk][q] = *(const bf16x8*)&Br[(wn * 64 + q * 32 + r) * LDK + kk * 16 + hq * 8];
      }
    auto mfma_group = [&](int kk) {
#pragma unroll
      for (int q = 0; q < 2; q++)
#pragma unroll
        for (int jj = 0; jj < 2; jj++) acc[q][jj] = mfma32(af[kk][q], bfr[kk][jj], acc[q][jj]);
    };
    // ---- operands of step i+1 (stage s1) -----------------------------------------------------------------------
    SP_MARK(1);
    if (FULL || i + 1 < nstep) stage_out(breg[s1], ekey[s1], p0[s1], p1[s1], i + 1, skey[s1]);
    SP_MARK(2);
    mfma_group(0);
    mfma_group(1);
    SP_MARK(3);
    // ---- loads of step i+4 into the stage just drained; restore the buffer of step i-1 for step i+2 -----------
    if (FULL || i + 4 < nstep) prefetch(breg[s1], ekey[s1], p0[s1], p1[s1], i + 4);
    SP_MARK(4);
    mfma_group(2);
    if (FULL || (i >= 1 && i + 2 < nstep)) unscatter(i - 1, skey[s2]);
    mfma_group(3);
    SP_MARK(5);
    __syncthreads();
    SP_MARK(6);
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  int i = 0;
  // head: steps 0..2 (no step i-1 to restore at i = 0)
  if (i < nstep) k_step(F_{}, S0{}, i++);
  if (i < nstep) k_step(F_{}, S1{}, i++);
  if (i < nstep) k_step(F_{}, S2{}, i++);
  // hot loop: three FULL steps per trip
  for (; i + 6 < nstep; i += 3) {
    k_step(T_{}, S0{}, i);
    k_step(T_{}, S1{}, i + 1);
    k_step(T_{}, S2{}, i + 2);
  }
  // tail
  for (; i < nstep; i += 3) {
    k_step(F_{}, S0{}, i);
    if (i + 1 < nstep) k_step(F_{}, S1{}, i + 1);
    if (i + 2 < nstep) k_step(F_{}, S2{}, i + 2);
  }

  float* C = g.C + (long)split * g.ncells * g.ldc;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int jj = 0; jj < 2; jj++) {
      int n = n0 + wn * 64 + jj * 32 + r;
      if (n >= g.N) continue;
#pragma unroll
      for (int e = 0; e < 16; e++) {
        int m = m0 + wm * 64 + i * 32 + acc_row(e, lane);
        if (m < g.ncells) C[(long)m * g.ldc + n] = acc[i][jj][e];
      }
    }
}

// ------------------------------------------------------------------------------------------------------
// wgrad: tile 128 hidden x 256 features x 64 cells (one cell block per K-step), 8 waves as 2 (hidden) x 4 (features)
//
// Same pipeline as the forward kernel with the operand roles swapped: the sparse X tile [feature][cell] lives in
// three LDS buffers (scatter for step i+1, MFMAs of step i, un-scatter of step i-1), the dense dY^T tile in two,
// one barrier per step, three steps of loads in flight in registers.
//
// ADAM: the gradient tile never goes to HBM.  The accumulators are laid out row-major in LDS (the operand buffers
// are free after the K loop) and every wave then streams whole 1 KB rows of the master weights and the two AdamW
// moments through the update, writing them and the bf16 shadow back: 26 B/parameter of HBM traffic instead of
// 4 (dW written) + 30 (optimiser kernel), and that traffic overlaps the MFMA phases of the other workgroups.
// Measured at C2 (900 x 100 k weights, 1024 cells): 556 us against 274 + 420 us for the two kernels; with a single
// K-step the launch takes 457 us, i.e. the epilogue streams at 5.1 TB/s and most of the K loop hides behind it.
// ------------------------------------------------------------------------------------------------------
template <bool HASVAL, bool ADAM, bool OUT16 = false, int EPB = 16>
__global__ __launch_bounds__(512) void spmm_dw_bf16_kernel(SpArgs g) {
  constexpr int EPT = HASVAL ? 5 : 2;             // sparse entries held per thread and stage
  constexpr int ASZ = 128 * LDK, BSZ = 256 * LDK;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* Bs = (__bf16*)smem_raw;        // [3][256][LDK]   X tile    [feature][cell]
  __bf16* As = Bs + 3 * BSZ;             // [2][128][LDK]   dYT tile  [hidden][cell]
  int* ptrs = (int*)(As + 2 * ASZ);      // [2][PCW + 1]: entry range [start, end) of this feature tile per cell block
  __shared__ int ovf[3];                 // step+1 of a step whose X buffer holds entries beyond the per-thread slots
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, hq = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;

  int b = blockIdx.x;
  int xcd = b & 7, j = b >> 3;
  int m_tile = j % g.tiles_m;
  int unit = (j / g.tiles_m) * 8 + xcd;
  if (unit >= g.tiles_n) return;
  const int m0 = m_tile * 128, n0 = unit * 256;
  const int nend = min(n0 + 256, g.F);
  const int nstep = g.ncb;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int jj = 0; jj < 2; jj++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][jj][e] = 0.f;

  u32x4 areg[3][2];
  uint32_t ekey[3][EPT];                          // key words, or packed (value, key) words when HASVAL
  int p0[3], p1[3];
  int skey[3][EPT];                               // LDS offsets this thread scattered for step j (slot j%3), -1 = none
  const int arow = tid >> 3, akc = tid & 7;       // dYT tile chunk of this thread (and +64 rows for the second one)

  auto prefetch = [&](u32x4 (&ar)[2], uint32_t (&ek)[EPT], int& q0, int& q1,
                      int step) {
    const int k0 = step * 64;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      int m = min(m0 + arow + 64 * i, g.N - 1);
      long k = min((long)k0 + akc * 8, g.ldd - 8);
      ar[i] = *(const u32x4*)(g.D + (long)m * g.ldd + k);
    }
    q0 = ptrs[step];
    q1 = ptrs[(PCW + 1) + step];
    const int elast = max(q1 - 1, 0);
#pragma unroll
    for (int i = 0; i < EPT; i++) {
      int e = min(q0 + tid + 512 * i, elast);
      if constexpr (HASVAL) ek[i] = g.pk[e];
      else ek[i] = (uint32_t)g.key[e];
    }
  };
  // registers of one stage -> LDS operands of `step`
  auto stage_out = [&](const u32x4 (&ar)[2], const uint32_t (&ek)[EPT], int q0,
                       int q1, int step, int (&sk)[EPT]) {
    const int k0 = step * 64;
    __bf16* Ad = As + (step & 1) * ASZ;
    __bf16* Bd = Bs + (step % 3) * BSZ;
    const bool kin = (long)k0 + akc * 8 < g.ldd;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      u32x4 v = ar[i];
      if (!(kin && m0 + arow + 64 * i < g.N)) v = u32x4{0u, 0u, 0u, 0u};
      *(u32x4*)&Ad[(arow + 64 * i) * LDK + akc * 8] = v;
    }
#pragma unroll
    for (int i = 0; i < EPT; i++) {
      sk[i] = -1;
      if (q0 + tid + 512 * i < q1) {
        const uint32_t w = ek[i];
        if constexpr (HASVAL) {                   // n0 is a multiple of 256: the low key bits are the position in the tile
          int off = ((w >> 6) & 255) * LDK + (w & 63);
          ((uint16_t*)Bd)[off] = (uint16_t)(w >> 16);
          sk[i] = off;
        } else {
          int off = (((int)w >> 6) - n0) * LDK + (w & 63);
          Bd[off] = (__bf16)1.f;
          sk[i] = off;
        }
      }
    }
    if (q0 + 512 * EPT < q1) {                    // overflow of a dense (cell block, feature tile): rare, uniform
      if (tid == 0) ovf[step % 3] = step + 1;
      for (int e = q0 + tid + 512 * EPT; e < q1; e += 512) {
        if constexpr (HASVAL) {
          const uint32_t w = gload_b32_sync(g.pk + e);
          ((uint16_t*)Bd)[((w >> 6) & 255) * LDK + (w & 63)] = (uint16_t)(w >> 16);
        } else {
          int key = (int)gload_b32_sync(g.key + e);
          Bd[((key >> 6) - n0) * LDK + (key & 63)] = (__bf16)1.f;
        }
      }
    }
  };
  auto zero_B = [&](int buf) {
    u32x4 z = {0u, 0u, 0u, 0u};
    u32x4* q = (u32x4*)(Bs + buf * BSZ);
    for (int i = tid; i < BSZ * 2 / 16; i += 512) q[i] = z;
  };
  // undo the scatter of `step` (whose MFMAs are done): its buffer is all zero again afterwards
  auto unscatter = [&](int step, const int (&sk)[EPT]) {
    __bf16* Bd = Bs + (step % 3) * BSZ;
    // the tag was written two barriers ago (when the step was staged) and the slot is next written one barrier
    // from now, so every thread reads the same value; a stale tag of step-3 never matches
    if (ovf[step % 3] == step + 1) {
      zero_B(step % 3);
    } else {
#pragma unroll
      for (int i = 0; i < EPT; i++)
        if (sk[i] >= 0) Bd[sk[i]] = (__bf16)0.f;
    }
  };

  // entry ranges of this feature tile in every cell block (ncb <= PCW + 1, checked by the host)
  for (int i = tid; i < 2 * (PCW + 1); i += 512) {
    int which = i / (PCW + 1), cbx = i % (PCW + 1);
    ptrs[i] = (cbx < g.ncb) ? g.ptr[(long)cbx * g.F + (which ? nend : n0)] : 0;
  }
  zero_B(0);
  zero_B(1);
  zero_B(2);
  if (tid < 3) ovf[tid] = 0;
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 3; s++)
    if (s < nstep) prefetch(areg[s], ekey[s], p0[s], p1[s], s);
  stage_out(areg[0], ekey[0], p0[0], p1[0], 0, skey[0]);
  if (3 < nstep) prefetch(areg[0], ekey[0], p0[0], p1[0], 3);
  __syncthreads();

  // One K-step; FULL = steps i-1 .. i+4 all exist (straight-line body, see the forward kernel)
  auto k_step = [&](auto full_c, auto s_c, int i) {
    constexpr bool FULL = decltype(full_c)::value;
    constexpr int s = decltype(s_c)::value;       // i % 3
    constexpr int s1 = (s + 1) % 3, s2 = (s + 2) % 3;
    const __bf16* Ar = As + (i & 1) * ASZ;
    const __bf16* Br = Bs + s * BSZ;
    bf16x8 af[4][2], bfr[4][2];
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
#pragma unroll
      for (int q = 0; q < 2; q++) {
        af[kk][q] = *(const bf16x8*)&Ar[(wm * 64 + q * 32 + r) * LDK + kk * 16 + hq * 8];
        bfr[kk][q] = *(const bf16x8*)&Br[(wn * 64 + q * 32 + r) * LDK + kk * 16 + hq * 8];
      }
    auto mfma_group = [&](int kk) {
#pragma unroll
      for (int q = 0; q < 2; q++)
#pragma unroll
        for (int jj = 0; jj < 2; jj++) acc[q][jj] = mfma32(af[kk][q], bfr[kk][jj], acc[q][jj]);
    };
    if (FULL || i + 1 < nstep) stage_out(areg[s1], ekey[s1], p0[s1], p1[s1], i + 1, skey[s1]);
    mfma_group(0);
    mfma_group(1);
    if (FULL || i + 4 < nstep) prefetch(areg[s1], ekey[s1], p0[s1], p1[s1], i + 4);
    mfma_group(2);
    if (FULL || (i >= 1 && i + 2 < nstep)) unscatter(i - 1, skey[s2]);
    mfma_group(3);
    __syncthreads();
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  int i = 0;
  if (i < nstep) k_step(F_{}, S0{}, i++);
  if (i < nstep) k_step(F_{}, S1{}, i++);
  if (i < nstep) k_step(F_{}, S2{}, i++);
  for (; i + 6 < nstep; i += 3) {
    k_step(T_{}, S0{}, i);
    k_step(T_{}, S1{}, i + 1);
    k_step(T_{}, S2{}, i + 2);
  }
  for (; i < nstep; i += 3) {
    k_step(F_{}, S0{}, i);
    if (i + 1 < nstep) k_step(F_{}, S1{}, i + 1);
    if (i + 2 < nstep) k_step(F_{}, S2{}, i + 2);
  }

  if constexpr (!ADAM) {
#pragma unroll
    for (int i2 = 0; i2 < 2; i2++)
#pragma unroll
      for (int jj = 0; jj < 2; jj++) {
        int n = n0 + wn * 64 + jj * 32 + r;
        if (n >= g.F) continue;
#pragma unroll
        for (int e = 0; e < 16; e++) {
          int m = m0 + wm * 64 + i2 * 32 + acc_row(e, lane);
          if (m < g.N) {
            if constexpr (OUT16) ((__bf16*)g.C)[(long)m * g.ldc + n] = (__bf16)acc[i2][jj][e];
            else g.C[(long)m * g.ldc + n] = acc[i2][jj][e];
          }
        }
      }
  } else {
    // gradient tile -> LDS [128][GLD] (all waves passed the last barrier of the K loop: the operand buffers are dead)
    constexpr int GLD = 260;
    float* gt = (float*)smem_raw;
#pragma unroll
    for (int i2 = 0; i2 < 2; i2++)
#pragma unroll
      for (int jj = 0; jj < 2; jj++)
#pragma unroll
        for (int e = 0; e < 16; e++)
          gt[(wm * 64 + i2 * 32 + acc_row(e, lane)) * GLD + wn * 64 + jj * 32 + r] = acc[i2][jj][e];
    __syncthreads();
    typedef float f4 __attribute__((ext_vector_type(4)));
    const float bc1 = g.bc_dev ? g.bc_dev[0] : g.bc1, bc2 = g.bc_dev ? g.bc_dev[1] : g.bc2;
    const AdamHyper hy = adam_hyper(g.lr, g.wd, g.b1, g.b2, g.eps, bc1, bc2, g.gscale);
    const int c4 = (tid & 63) * 4, rw = tid >> 6;   // one wave per 1 KB row segment, 8 rows per pass
    const int n = n0 + c4;
    const bool col_ok = n + 4 <= g.F;               // a float4 that straddles column F (F % 4 != 0) takes the scalar path below
#pragma unroll 1
    for (int pass = 0; pass < 16; pass += EPB) {
      // EPB row segments of the three streams in flight per lane before the first use: with one workgroup per CU the
      // epilogue's HBM rate is set by the bytes each CU keeps outstanding
      f4 pp[EPB], mm[EPB], vv[EPB];
      // unpredicated (rows / columns past the end read a clamped, valid, 16-byte aligned address; only the stores are
      // predicated): the loads and the updates below form one straight line, so each row is processed when ITS loads
      // have landed (counted waits) instead of after all of them
      const int nc = col_ok ? n : max((g.F & ~3) - 4, 0);
#pragma unroll
      for (int u = 0; u < EPB; u++) {
        const int m = min(m0 + (pass + u) * 8 + rw, g.N - 1);
        const long o = (long)m * g.ldc + nc;
        pp[u] = __builtin_nontemporal_load((const f4*)(g.C + o));
        mm[u] = __builtin_nontemporal_load((const f4*)(g.adam_m + o));
        vv[u] = __builtin_nontemporal_load((const f4*)(g.adam_v + o));
      }
#pragma unroll
      for (int u = 0; u < EPB; u++) {
        int row = (pass + u) * 8 + rw;
        int m = m0 + row;
        f4 gg = *(const f4*)&gt[row * GLD + c4];
        float pa[4], ma[4], va[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          pa[k] = pp[u][k]; ma[k] = mm[u][k]; va[k] = vv[u][k];
          adamw_update_fast(hy, pa[k], gg[k], ma[k], va[k]);
        }
        if (col_ok && m < g.N) {
          long o = (long)m * g.ldc + n;
          f4 po = {pa[0], pa[1], pa[2], pa[3]}, mo = {ma[0], ma[1], ma[2], ma[3]}, vo = {va[0], va[1], va[2], va[3]};
          __builtin_nontemporal_store(po, (f4*)(g.C + o));
          __builtin_nontemporal_store(mo, (f4*)(g.adam_m + o));
          __builtin_nontemporal_store(vo, (f4*)(g.adam_v + o));
          bf16x4 sh;
#pragma unroll
          for (int k = 0; k < 4; k++) sh[k] = (__bf16)pa[k];
          *(bf16x4*)(g.adam_sh + (long)m * g.adam_ldsh + n) = sh;
        } else if (n < g.F && m < g.N) {
          // the last columns of a feature count that is no multiple of 4: element by element (one lane per row); the
          // columns behind F are not this kernel's (a covariate block's weights, or padding)
          for (int k = 0; n + k < g.F; k++) {
            const long o = (long)m * g.ldc + n + k;
            float p_ = g.C[o], m_ = g.adam_m[o], v_ = g.adam_v[o];
            adamw_update_fast(hy, p_, gg[k], m_, v_);
            g.C[o] = p_;
            g.adam_m[o] = m_;
            g.adam_v[o] = v_;
            g.adam_sh[(long)m * g.adam_ldsh + n + k] = (__bf16)p_;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// wgrad, narrow variant: tile 128 hidden x 128 features x 64 cells, 4 waves as 2 (hidden) x 2 (features), 74 KB of
// LDS -> TWO workgroups per CU, so the HBM-bound AdamW epilogue of one overlaps the MFMA-bound K loop of the other
// (the 256-feature kernel above fills the LDS with one workgroup and alternates the two phases).  Two X buffers
// instead of three: the buffer step i+1 scatters into still holds step i-1, so a step is
//     un-scatter(i-1) | barrier | stage(i+1), MFMAs(i), prefetch(i+4) | barrier.
// ------------------------------------------------------------------------------------------------------
template <bool HASVAL, bool ADAM, bool OUT16 = false, int EPB = 16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void spmm_dw128_bf16_kernel(SpArgs g) {
  constexpr int EPT = HASVAL ? 5 : 2;             // sparse entries held per thread and stage
  constexpr int ASZ = 128 * LDK, BSZ = 128 * LDK;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* Bs = (__bf16*)smem_raw;        // [2][128][LDK]   X tile    [feature][cell]
  __bf16* As = Bs + 2 * BSZ;             // [2][128][LDK]   dYT tile  [hidden][cell]
  int* ptrs = (int*)(As + 2 * ASZ);      // [2][PCW + 1]: entry range [start, end) of this feature tile per cell block
  __shared__ int ovf[3];                 // step+1 of a step whose X buffer holds entries beyond the per-thread slots
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, hq = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;

  int b = blockIdx.x;
#ifdef SPMM_PROF
  const bool prof_on = (threadIdx.x == 0) && g.prof && (blockIdx.x % 97 == 5);
  const long prof_slot = (blockIdx.x / 97) % 500;
#define DW_MARK(m) do { if (prof_on) g.prof[prof_slot * 8 + (m)] = clock64(); } while (0)
  DW_MARK(0);
#else
#define DW_MARK(m) do { } while (0)
#endif
  int xcd = b & 7, j = b >> 3;
  int m_tile = j % g.tiles_m;
  int unit = (j / g.tiles_m) * 8 + xcd;
  if (unit >= g.tiles_n) return;
  const int m0 = m_tile * 128, n0 = unit * 128;
  const int nend = min(n0 + 128, g.F);
  const int nstep = g.ncb;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int jj = 0; jj < 2; jj++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][jj][e] = 0.f;

  u32x4 areg[3][4];
  uint32_t ekey[3][EPT];                          // key words, or packed (value, key) words when HASVAL
  int p0[3], p1[3];
  int skey[3][EPT];                               // LDS offsets this thread scattered for step j (slot j%3), -1 = none
  const int arow = tid >> 3, akc = tid & 7;       // dYT tile chunks of this thread: rows arow, +32, +64, +96

  auto prefetch = [&](u32x4 (&ar)[4], uint32_t (&ek)[EPT], int& q0, int& q1,
                      int step) {
    const int k0 = step * 64;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int m = min(m0 + arow + 32 * i, g.N - 1);
      long k = min((long)k0 + akc * 8, g.ldd - 8);
      ar[i] = *(const u32x4*)(g.D + (long)m * g.ldd + k);
    }
    q0 = ptrs[step];
    q1 = ptrs[(PCW + 1) + step];
    const int elast = max(q1 - 1, 0);
#pragma unroll
    for (int i = 0; i < EPT; i++) {
      int e = min(q0 + tid + 256 * i, elast);
      if constexpr (HASVAL) ek[i] = g.pk[e];
      else ek[i] = (uint32_t)g.key[e];
    }
  };
  // registers of one stage -> LDS operands of `step`
  auto stage_out = [&](const u32x4 (&ar)[4], const uint32_t (&ek)[EPT], int q0,
                       int q1, int step, int (&sk)[EPT]) {
    const int k0 = step * 64;
    __bf16* Ad = As + (step & 1) * ASZ;
    __bf16* Bd = Bs + (step & 1) * BSZ;
    const bool kin = (long)k0 + akc * 8 < g.ldd;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      u32x4 v = ar[i];
      if (!(kin && m0 + arow + 32 * i < g.N)) v = u32x4{0u, 0u, 0u, 0u};
      *(u32x4*)&Ad[(arow + 32 * i) * LDK + akc * 8] = v;
    }
#pragma unroll
    for (int i = 0; i < EPT; i++) {
      sk[i] = -1;
      if (q0 + tid + 256 * i < q1) {
        const uint32_t w = ek[i];
        if constexpr (HASVAL) {                   // n0 is a multiple of 128: the low key bits are the position in the tile
          int off = ((w >> 6) & 127) * LDK + (w & 63);
          ((uint16_t*)Bd)[off] = (uint16_t)(w >> 16);
          sk[i] = off;
        } else {
          int off = (((int)w >> 6) - n0) * LDK + (w & 63);
          Bd[off] = (__bf16)1.f;
          sk[i] = off;
        }
      }
    }
    if (q0 + 256 * EPT < q1) {                    // overflow of a dense (cell block, feature tile): rare, uniform
      if (tid == 0) ovf[step % 3] = step + 1;
      for (int e = q0 + tid + 256 * EPT; e < q1; e += 256) {
        if constexpr (HASVAL) {
          const uint32_t w = gload_b32_sync(g.pk + e);
          ((uint16_t*)Bd)[((w >> 6) & 127) * LDK + (w & 63)] = (uint16_t)(w >> 16);
        } else {
          int key = (int)gload_b32_sync(g.key + e);
          Bd[((key >> 6) - n0) * LDK + (key & 63)] = (__bf16)1.f;
        }
      }
    }
  };
  auto zero_B = [&](int buf) {
    u32x4 z = {0u, 0u, 0u, 0u};
    u32x4* q = (u32x4*)(Bs + buf * BSZ);
    for (int i = tid; i < BSZ * 2 / 16; i += 256) q[i] = z;
  };
  // undo the scatter of `step` (whose MFMAs are done): its buffer is all zero again afterwards
  auto unscatter = [&](int step, const int (&sk)[EPT]) {
    __bf16* Bd = Bs + (step & 1) * BSZ;
    // the tag was written barriers ago (when the step was staged) and its slot is next written by step + 3, so every
    // thread reads the same value; a stale tag of step-3 never matches
    if (ovf[step % 3] == step + 1) {
      zero_B(step & 1);
    } else {
#pragma unroll
      for (int i = 0; i < EPT; i++)
        if (sk[i] >= 0) Bd[sk[i]] = (__bf16)0.f;
    }
  };

  // entry ranges of this feature tile in every cell block (ncb <= PCW + 1, checked by the host)
  for (int i = tid; i < 2 * (PCW + 1); i += 256) {
    int which = i / (PCW + 1), cbx = i % (PCW + 1);
    ptrs[i] = (cbx < g.ncb) ? g.ptr[(long)cbx * g.F + (which ? nend : n0)] : 0;
  }
  zero_B(0);
  zero_B(1);
  if (tid < 3) ovf[tid] = 0;
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 3; s++)
    if (s < nstep) prefetch(areg[s], ekey[s], p0[s], p1[s], s);
  stage_out(areg[0], ekey[0], p0[0], p1[0], 0, skey[0]);
  if (3 < nstep) prefetch(areg[0], ekey[0], p0[0], p1[0], 3);
  __syncthreads();

  // One K-step; FULL = steps i-1 .. i+4 all exist (straight-line body, see the forward kernel)
  auto k_step = [&](auto full_c, auto s_c, int i) {
    constexpr bool FULL = decltype(full_c)::value;
    constexpr int s = decltype(s_c)::value;       // i % 3
    constexpr int s1 = (s + 1) % 3, s2 = (s + 2) % 3;
    // phase 1: the X buffer step i+1 is about to use still holds the entries of step i-1 (whose MFMAs are done)
    if (FULL || (i >= 1 && i + 1 < nstep)) unscatter(i - 1, skey[s2]);
    __syncthreads();
    const __bf16* Ar = As + (i & 1) * ASZ;
    const __bf16* Br = Bs + (i & 1) * BSZ;
    bf16x8 af[4][2], bfr[4][2];
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
#pragma unroll
      for (int q = 0; q < 2; q++) {
        af[kk][q] = *(const bf16x8*)&Ar[(wm * 64 + q * 32 + r) * LDK + kk * 16 + hq * 8];
        bfr[kk][q] = *(const bf16x8*)&Br[(wn * 64 + q * 32 + r) * LDK + kk * 16 + hq * 8];
      }
    auto mfma_group = [&](int kk) {
#pragma unroll
      for (int q = 0; q < 2; q++)
#pragma unroll
        for (int jj = 0; jj < 2; jj++) acc[q][jj] = mfma32(af[kk][q], bfr[kk][jj], acc[q][jj]);
    };
    if (FULL || i + 1 < nstep) stage_out(areg[s1], ekey[s1], p0[s1], p1[s1], i + 1, skey[s1]);
    mfma_group(0);
    mfma_group(1);
    if (FULL || i + 4 < nstep) prefetch(areg[s1], ekey[s1], p0[s1], p1[s1], i + 4);
    mfma_group(2);
    mfma_group(3);
    __syncthreads();
  };
  DW_MARK(1);
  using T_ = std::true_type;
  using F_ = std::false_type;
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  int i = 0;
  if (i < nstep) k_step(F_{}, S0{}, i++);
  if (i < nstep) k_step(F_{}, S1{}, i++);
  if (i < nstep) k_step(F_{}, S2{}, i++);
  for (; i + 6 < nstep; i += 3) {
    k_step(T_{}, S0{}, i);
    k_step(T_{}, S1{}, i + 1);
    k_step(T_{}, S2{}, i + 2);
  }
  for (; i < nstep; i += 3) {
    k_step(F_{}, S0{}, i);
    if (i + 1 < nstep) k_step(F_{}, S1{}, i + 1);
    if (i + 2 < nstep) k_step(F_{}, S2{}, i + 2);
  }

  DW_MARK(2);
  if constexpr (!ADAM) {
#pragma unroll
    for (int i2 = 0; i2 < 2; i2++)
#pragma unroll
      for (int jj = 0; jj < 2; jj++) {
        int n = n0 + wn * 64 + jj * 32 + r;
        if (n >= g.F) continue;
#pragma unroll
        for (int e = 0; e < 16; e++) {
          int m = m0 + wm * 64 + i2 * 32 + acc_row(e, lane);
          if (m < g.N) {
            if constexpr (OUT16) ((__bf16*)g.C)[(long)m * g.ldc + n] = (__bf16)acc[i2][jj][e];
            else g.C[(long)m * g.ldc + n] = acc[i2][jj][e];
          }
        }
      }
  } else {
    // gradient tile -> LDS [128][GLD] (all waves passed the last barrier of the K loop: the operand buffers are dead)
    constexpr int GLD = 132;
    float* gt = (float*)smem_raw;
#pragma unroll
    for (int i2 = 0; i2 < 2; i2++)
#pragma unroll
      for (int jj = 0; jj < 2; jj++)
#pragma unroll
        for (int e = 0; e < 16; e++)
          gt[(wm * 64 + i2 * 32 + acc_row(e, lane)) * GLD + wn * 64 + jj * 32 + r] = acc[i2][jj][e];
    __syncthreads();
    DW_MARK(3);
    typedef float f4 __attribute__((ext_vector_type(4)));
    const float bc1 = g.bc_dev ? g.bc_dev[0] : g.bc1, bc2 = g.bc_dev ? g.bc_dev[1] : g.bc2;
    const AdamHyper hy = adam_hyper(g.lr, g.wd, g.b1, g.b2, g.eps, bc1, bc2, g.gscale);
    const int c4 = (tid & 31) * 4, rw = tid >> 5;   // half a wave per 512 B row segment, 8 rows per pass
    const int n = n0 + c4;
    const bool col_ok = n + 4 <= g.F;               // a float4 that straddles column F (F % 4 != 0) takes the scalar path below
#pragma unroll 1
    for (int pass = 0; pass < 16; pass += EPB) {
      // EPB row segments of the three streams in flight per lane before the first use: with one workgroup per CU the
      // epilogue's HBM rate is set by the bytes each CU keeps outstanding
      f4 pp[EPB], mm[EPB], vv[EPB];
      // unpredicated (rows / columns past the end read a clamped, valid, 16-byte aligned address; only the stores are
      // predicated): the loads and the updates below form one straight line, so each row is processed when ITS loads
      // have landed (counted waits) instead of after all of them
      const int nc = col_ok ? n : max((g.F & ~3) - 4, 0);
#pragma unroll
      for (int u = 0; u < EPB; u++) {
        const int m = min(m0 + (pass + u) * 8 + rw, g.N - 1);
        const long o = (long)m * g.ldc + nc;
        pp[u] = __builtin_nontemporal_load((const f4*)(g.C + o));
        mm[u] = __builtin_nontemporal_load((const f4*)(g.adam_m + o));
        vv[u] = __builtin_nontemporal_load((const f4*)(g.adam_v + o));
      }
#pragma unroll
      for (int u = 0; u < EPB; u++) {
        int row = (pass + u) * 8 + rw;
        int m = m0 + row;
        f4 gg = *(const f4*)&gt[row * GLD + c4];
        float pa[4], ma[4], va[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          pa[k] = pp[u][k]; ma[k] = mm[u][k]; va[k] = vv[u][k];
          adamw_update_fast(hy, pa[k], gg[k], ma[k], va[k]);
        }
        if (col_ok && m < g.N) {
          long o = (long)m * g.ldc + n;
          f4 po = {pa[0], pa[1], pa[2], pa[3]}, mo = {ma[0], ma[1], ma[2], ma[3]}, vo = {va[0], va[1], va[2], va[3]};
          __builtin_nontemporal_store(po, (f4*)(g.C + o));
          __builtin_nontemporal_store(mo, (f4*)(g.adam_m + o));
          __builtin_nontemporal_store(vo, (f4*)(g.adam_v + o));
          bf16x4 sh;
#pragma unroll
          for (int k = 0; k < 4; k++) sh[k] = (__bf16)pa[k];
          *(bf16x4*)(g.adam_sh + (long)m * g.adam_ldsh + n) = sh;
        } else if (n < g.F && m < g.N) {
          // the last columns of a feature count that is no multiple of 4: element by element (one lane per row); the
          // columns behind F are not this kernel's (a covariate block's weights, or padding)
          for (int k = 0; n + k < g.F; k++) {
            const long o = (long)m * g.ldc + n + k;
            float p_ = g.C[o], m_ = g.adam_m[o], v_ = g.adam_v[o];
            adamw_update_fast(hy, p_, gg[k], m_, v_);
            g.C[o] = p_;
            g.adam_m[o] = m_;
            g.adam_v[o] = v_;
            g.adam_sh[(long)m * g.adam_ldsh + n + k] = (__bf16)p_;
          }
        }
      }
    }
  }
  DW_MARK(5);
}

// ------------------------------------------------------------------------------------------------------
// helpers: fp32 -> bf16 cast into a padded shadow, and fp32 [rows][cols] -> bf16 transposed [cols][ldt]
// ------------------------------------------------------------------------------------------------------
template <bool RELU>
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, long rows, long cols, long lds_,
                                                        __bf16* __restrict__ dst, long ldd) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  long n = rows * ldd;
  if (i >= n) return;
  long rr = i / ldd, c = i % ldd;
  float v = (c < cols) ? src[rr * lds_ + c] : 0.f;
  if (RELU && v < 0.f) v = 0.f;                  // a NaN stays a NaN, as in torch.relu
  dst[i] = (__bf16)v;
}

__global__ __launch_bounds__(256) void transpose_bf16_kernel(const float* __restrict__ src, int rows, int cols, long lds_,
                                                             __bf16* __restrict__ dst, long ldt) {
  __shared__ float tile[64][65];
  int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    int rr = i >> 6, cc = i & 63;
    tile[rr][cc] = (r0 + rr < rows && c0 + cc < cols) ? src[(long)(r0 + rr) * lds_ + c0 + cc] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    int cc = i >> 6, rr = i & 63;
    if (c0 + cc < cols && r0 + rr < ldt) dst[(long)(c0 + cc) * ldt + r0 + rr] = (__bf16)tile[rr][cc];
  }
}

// fp32 -> three bf16 planes hi | mid | lo (bf16_split<3>: hi + mid + lo carries 24 significant bits), each [rows][ldd],
// `plane` elements apart: the operands of the fp32-equivalent first layer on the bf16 kernels when the sparse operand is
// exact in bf16 (implicit ones): X W^T = X hi^T + X mid^T + X lo^T, three launches of the one-term kernel
__global__ __launch_bounds__(256) void split3_bf16_kernel(const float* __restrict__ src, long rows, long cols, long lds_,
                                                          __bf16* __restrict__ dst, long ldd, long plane) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  long n = rows * ldd;
  if (i >= n) return;
  long rr = i / ldd, c = i % ldd;
  float v = (c < cols) ? src[rr * lds_ + c] : 0.f;
  __bf16 t[3];
  bf16_split<3>(v, t);
  dst[i] = t[0];
  dst[plane + i] = t[1];
  dst[2 * plane + i] = t[2];
}

// fp32 [rows][cols] -> bf16 [cols][ldt] holding the three terms side by side: dst[c][t * rows + r] = term_t(src[r][c]);
// columns 3 * rows .. ldt - 1 are zeroed.  With a tile list of the same minibatch repeated three times this makes
// dW = dY^T X of the fp32-equivalent mode one launch of the one-term weight-gradient kernel over 3 * rows cells
__global__ __launch_bounds__(256) void split3_transpose_bf16_kernel(const float* __restrict__ src, int rows, int cols,
                                                                    long lds_, __bf16* __restrict__ dst, long ldt) {
  __shared__ float tile[64][65];
  int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    int rr = i >> 6, cc = i & 63;
    tile[rr][cc] = (r0 + rr < rows && c0 + cc < cols) ? src[(long)(r0 + rr) * lds_ + c0 + cc] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    int cc = i >> 6, rr = i & 63;
    if (c0 + cc < cols && r0 + rr < rows) {
      __bf16 t[3];
      bf16_split<3>(tile[rr][cc], t);
      __bf16* d = dst + (long)(c0 + cc) * ldt + r0 + rr;
      d[0] = t[0];
      d[rows] = t[1];
      d[2 * (long)rows] = t[2];
    }
  }
  // padding behind the three terms (the weight-gradient kernel reads whole 64-cell blocks)
  if (blockIdx.y == 0) {
    int pad = (int)(ldt - 3L * rows);
    for (int i = threadIdx.x; i < 64 * pad; i += 256) {
      int cc = i / pad, k = i % pad;
      if (c0 + cc < cols) dst[(long)(c0 + cc) * ldt + 3L * rows + k] = (__bf16)0.f;
    }
  }
}

extern "C" int scp_split3_bf16(const float* src, long rows, long cols, long ld_src, void* dst, long ld_dst, long plane,
                               void* stream_) {
  SCP_CHECK(src && dst && rows > 0 && cols > 0 && ld_dst >= cols && plane >= rows * ld_dst, SCP_ERR_ARG,
            "scp_split3_bf16: bad argument");
  long n = rows * ld_dst;
  hipLaunchKernelGGL(split3_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, src, rows,
                     cols, ld_src, (__bf16*)dst, ld_dst, plane);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

extern "C" int scp_split3_transpose_bf16(const float* src, int rows, int cols, long ld_src, void* dst, long ld_dst,
                                         void* stream_) {
  SCP_CHECK(src && dst && rows > 0 && cols > 0 && ld_dst >= 3L * rows, SCP_ERR_ARG, "scp_split3_transpose_bf16: bad argument");
  dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  hipLaunchKernelGGL(split3_transpose_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream_, src, rows, cols, ld_src,
                     (__bf16*)dst, ld_dst);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

extern "C" int scp_cast_bf16(const float* src, long rows, long cols, long ld_src, void* dst, long ld_dst, void* stream_) {
  SCP_CHECK(src && dst && rows > 0 && cols > 0 && ld_dst >= cols, SCP_ERR_ARG, "scp_cast_bf16: bad argument");
  long n = rows * ld_dst;
  hipLaunchKernelGGL(cast_bf16_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, src,
                     rows, cols, ld_src, (__bf16*)dst, ld_dst);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

// dst = bf16(relu(src)): the input of the ZINB zero-inflation layer, pi = Linear(relu(z)) (model.py:240,255-256), cast
// for the wide GEMM in the same pass that applies the relu
extern "C" int scp_relu_cast_bf16(const float* src, long rows, long cols, long ld_src, void* dst, long ld_dst, void* stream_) {
  SCP_CHECK(src && dst && rows > 0 && cols > 0 && ld_dst >= cols, SCP_ERR_ARG, "scp_relu_cast_bf16: bad argument");
  long n = rows * ld_dst;
  hipLaunchKernelGGL(cast_bf16_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, src,
                     rows, cols, ld_src, (__bf16*)dst, ld_dst);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

extern "C" int scp_transpose_bf16(const float* src, int rows, int cols, long ld_src, void* dst, long ld_dst,
                                  void* stream_) {
  SCP_CHECK(src && dst && rows > 0 && cols > 0 && ld_dst >= rows, SCP_ERR_ARG, "scp_transpose_bf16: bad argument");
  dim3 grid((cols + 63) / 64, (int)((ld_dst + 63) / 64));
  hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream_, src, rows, cols, ld_src,
                     (__bf16*)dst, ld_dst);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

static int set_smem(const void* fn, size_t smem, uint64_t* done) { return scp_set_smem(fn, smem, done); }

// Feature-tile width of the fused weight-gradient + AdamW launch: 128 (two workgroups per CU: the AdamW epilogue of one
// overlaps the K loop of the other) -- measured at C2: 519 against 555 us for the 900 x 100 k ATAC weights; for the
// 900 x 20 k RNA weights 150 against 159 us since value-carrying entries travel as one packed word (with separate key
// and value registers the narrow variant spilled and lost, 175 against 160 us).  SCPAIR_DW_TILE = 128 / 256 overrides.
static int dw_tile_features(bool has_val) {
  static int w = -1;
  if (w < 0) {
    const char* e = getenv("SCPAIR_DW_TILE");
    w = e ? atoi(e) : 0;
  }
  if (w == 128 || w == 256) return w;
  (void)has_val;
  return 128;
}

// smallest split-K factor scp_spmm_fwd_bf16 accepts for F features (its per-slab offset table holds PCH K-steps)
extern "C" int scp_spmm_fwd_bf16_min_split(int F) {
  int ksteps = (F + 63) / 64;
  return (ksteps + PCH - 1) / PCH;
}

extern "C" int scp_spmm_fwd_bf16(int ncells, int N, int F, const int32_t* t_ptr, const int32_t* t_key, const float* t_val,
                                 const uint32_t* t_pk, const void* Wbf, long ldw, float* Y, long ldy, int nsplit, void* stream_) {
  SCP_CHECK(ncells > 0 && N > 0 && F > 0, SCP_ERR_ARG, "scp_spmm_fwd_bf16: sizes must be positive");
  SCP_CHECK(t_ptr && t_key && Wbf && Y, SCP_ERR_ARG, "scp_spmm_fwd_bf16: null pointer");
  SCP_CHECK(!t_val || t_pk, SCP_ERR_ARG, "scp_spmm_fwd_bf16: value-carrying tile lists need the packed entries (t_pk) scp_tiles_build writes");
  SCP_CHECK(ldw % 8 == 0 && ldw >= F && ((uintptr_t)Wbf % 16) == 0, SCP_ERR_ARG,
            "scp_spmm_fwd_bf16: shadow weights need a 16-byte aligned base and ld %% 8 == 0");
  static uint64_t done_v = 0, done_n = 0;
  size_t smem = (size_t)(3 * 256 + 2 * 128) * LDK * sizeof(__bf16) + 4 * (PCH + 1) * sizeof(int);
  int rc = t_val ? set_smem((const void*)spmm_fwd_bf16_kernel<true>, smem, &done_v)
                 : set_smem((const void*)spmm_fwd_bf16_kernel<false>, smem, &done_n);
  if (rc) return rc;
  SpArgs g = {};
  g.ncells = ncells; g.N = N; g.F = F; g.ptr = t_ptr; g.key = t_key; g.val = t_val; g.pk = t_pk; g.ncb = (ncells + 63) / 64;
  g.D = (const __bf16*)Wbf; g.ldd = ldw; g.C = Y; g.ldc = ldy;
  g.tiles_m = (ncells + 255) / 256; g.tiles_n = (N + 127) / 128;
  int ksteps = (F + 63) / 64;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > ksteps) nsplit = ksteps;
  int per = (ksteps + nsplit - 1) / nsplit;
  g.ksplit_len = per * 64;
  g.nsplit = (ksteps + per - 1) / per;
  SCP_CHECK(per <= PCH, SCP_ERR_SHAPE,
            "scp_spmm_fwd_bf16: %d K-steps per slab exceed %d; use nsplit >= scp_spmm_fwd_bf16_min_split(F)", per, PCH);
  int units = g.tiles_n * g.nsplit;
  int grid = 8 * g.tiles_m * ((units + 7) / 8);
#ifdef SPMM_PROF
  if (!sp_prof_dev) hipMalloc(&sp_prof_dev, sizeof(unsigned long long) * 8 * 4096);
  hipMemsetAsync(sp_prof_dev, 0, sizeof(unsigned long long) * 8 * 4096, (hipStream_t)stream_);
  g.prof = sp_prof_dev;
#endif
  if (t_val)
    hipLaunchKernelGGL(spmm_fwd_bf16_kernel<true>, dim3(grid), dim3(512), smem, (hipStream_t)stream_, g);
  else
    hipLaunchKernelGGL(spmm_fwd_bf16_kernel<false>, dim3(grid), dim3(512), smem, (hipStream_t)stream_, g);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

extern "C" int scp_spmm_dw_bf16(int ncells, int N, int F, const int32_t* t_ptr, const int32_t* t_key, const float* t_val,
                                const uint32_t* t_pk, const void* dYT, long ldt, float* dW, long ldw, void* stream_) {
  SCP_CHECK(ncells > 0 && N > 0 && F > 0, SCP_ERR_ARG, "scp_spmm_dw_bf16: sizes must be positive");
  SCP_CHECK(t_ptr && t_key && dYT && dW, SCP_ERR_ARG, "scp_spmm_dw_bf16: null pointer");
  SCP_CHECK(!t_val || t_pk, SCP_ERR_ARG, "scp_spmm_dw_bf16: value-carrying tile lists need the packed entries (t_pk) scp_tiles_build writes");
  SCP_CHECK(ldt % 8 == 0 && ldt >= ((ncells + 63) / 64) * 64 && ((uintptr_t)dYT % 16) == 0, SCP_ERR_ARG,
            "scp_spmm_dw_bf16: dYT needs a 16-byte aligned base, ld %% 8 == 0 and ld >= cells rounded up to 64");
  SCP_CHECK((ncells + 63) / 64 <= PCW + 1, SCP_ERR_SHAPE, "scp_spmm_dw_bf16: more than %d cells per call unsupported",
            64 * (PCW + 1));
  static uint64_t done_v = 0, done_n = 0;
  size_t smem = (size_t)(3 * 256 + 2 * 128) * LDK * sizeof(__bf16) + 2 * (PCW + 1) * sizeof(int);
  int rc = t_val ? set_smem((const void*)spmm_dw_bf16_kernel<true, false>, smem, &done_v)
                 : set_smem((const void*)spmm_dw_bf16_kernel<false, false>, smem, &done_n);
  if (rc) return rc;
  SpArgs g = {};
  g.ncells = ncells; g.N = N; g.F = F; g.ptr = t_ptr; g.key = t_key; g.val = t_val; g.pk = t_pk; g.ncb = (ncells + 63) / 64;
  g.D = (const __bf16*)dYT; g.ldd = ldt; g.C = dW; g.ldc = ldw;
  g.tiles_m = (N + 127) / 128; g.tiles_n = (F + 255) / 256; g.nsplit = 1;
  int grid = 8 * g.tiles_m * ((g.tiles_n + 7) / 8);
  if (t_val)
    hipLaunchKernelGGL((spmm_dw_bf16_kernel<true, false>), dim3(grid), dim3(512), smem, (hipStream_t)stream_, g);
  else
    hipLaunchKernelGGL((spmm_dw_bf16_kernel<false, false>), dim3(grid), dim3(512), smem, (hipStream_t)stream_, g);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

extern "C" int scp_spmm_dw_adamw_bf16(int ncells, int N, int F, const int32_t* t_ptr, const int32_t* t_key,
                                      const float* t_val, const uint32_t* t_pk, const void* dYT, long ldt, float* W, long ldw,
                                      float* exp_avg,
                                      float* exp_avg_sq, void* W_bf16, long ld_bf16, float lr, float weight_decay,
                                      float beta1, float beta2, float eps, float bias_corr1, float bias_corr2,
                                      float grad_scale, const float* bc_dev, void* stream_) {
  SCP_CHECK(ncells > 0 && N > 0 && F > 0, SCP_ERR_ARG, "scp_spmm_dw_adamw_bf16: sizes must be positive");
  SCP_CHECK(t_ptr && t_key && dYT && W && exp_avg && exp_avg_sq && W_bf16, SCP_ERR_ARG,
            "scp_spmm_dw_adamw_bf16: null pointer");
  SCP_CHECK(!t_val || t_pk, SCP_ERR_ARG,
            "scp_spmm_dw_adamw_bf16: value-carrying tile lists need the packed entries (t_pk) scp_tiles_build writes");
  SCP_CHECK(ldt % 8 == 0 && ldt >= ((ncells + 63) / 64) * 64 && ((uintptr_t)dYT % 16) == 0, SCP_ERR_ARG,
            "scp_spmm_dw_adamw_bf16: dYT needs a 16-byte aligned base, ld %% 8 == 0 and ld >= cells rounded up to 64");
  SCP_CHECK((ncells + 63) / 64 <= PCW + 1, SCP_ERR_SHAPE, "scp_spmm_dw_adamw_bf16: more than %d cells per call unsupported",
            64 * (PCW + 1));
  SCP_CHECK(ldw % 4 == 0 && ldw >= F && ldw >= 4 && ld_bf16 % 4 == 0 && ld_bf16 >= F &&
                (((uintptr_t)W | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16) == 0 && ((uintptr_t)W_bf16 % 8) == 0,
            SCP_ERR_SHAPE, "scp_spmm_dw_adamw_bf16: needs leading dimensions %% 4 == 0 and 16-byte aligned W / moments");
  SCP_CHECK(bc_dev || (bias_corr1 > 0.f && bias_corr2 > 0.f), SCP_ERR_ARG,
            "scp_spmm_dw_adamw_bf16: bias corrections must be > 0");
  static uint64_t done_v = 0, done_n = 0;
  size_t smem = (size_t)(3 * 256 + 2 * 128) * LDK * sizeof(__bf16) + 2 * (PCW + 1) * sizeof(int);
  static_assert((size_t)128 * 260 * sizeof(float) <= (size_t)(3 * 256 + 2 * 128) * LDK * sizeof(__bf16),
                "gradient tile must fit in the operand buffers");
  int rc = t_val ? set_smem((const void*)spmm_dw_bf16_kernel<true, true>, smem, &done_v)
                 : set_smem((const void*)spmm_dw_bf16_kernel<false, true>, smem, &done_n);
  if (rc) return rc;
  SpArgs g = {};
  g.ncells = ncells; g.N = N; g.F = F; g.ptr = t_ptr; g.key = t_key; g.val = t_val; g.pk = t_pk; g.ncb = (ncells + 63) / 64;
  g.D = (const __bf16*)dYT; g.ldd = ldt; g.C = W; g.ldc = ldw;
  g.tiles_m = (N + 127) / 128; g.tiles_n = (F + 255) / 256; g.nsplit = 1;
  g.adam_m = exp_avg; g.adam_v = exp_avg_sq; g.adam_sh = (__bf16*)W_bf16; g.adam_ldsh = ld_bf16;
  g.lr = lr; g.wd = weight_decay; g.b1 = beta1; g.b2 = beta2; g.eps = eps; g.bc1 = bias_corr1; g.bc2 = bias_corr2;
  g.gscale = grad_scale; g.bc_dev = bc_dev;
#ifdef SPMM_PROF
  if (!sp_prof_dev) hipMalloc(&sp_prof_dev, sizeof(unsigned long long) * 8 * 4096);
  hipMemsetAsync(sp_prof_dev, 0, sizeof(unsigned long long) * 8 * 4096, (hipStream_t)stream_);
  g.prof = sp_prof_dev;
#endif
  if (dw_tile_features(t_val != nullptr) == 128) {
    // two workgroups of 128 x 128 tiles per CU: the AdamW epilogue of one overlaps the K loop of the other
    static uint64_t d128_v = 0, d128_n = 0;
    const size_t smem128 = (size_t)(2 * 128 + 2 * 128) * LDK * sizeof(__bf16) + 2 * (PCW + 1) * sizeof(int);
    static_assert((size_t)128 * 132 * sizeof(float) <= (size_t)(2 * 128 + 2 * 128) * LDK * sizeof(__bf16),
                  "gradient tile must fit in the operand buffers");
    rc = t_val ? set_smem((const void*)spmm_dw128_bf16_kernel<true, true>, smem128, &d128_v)
               : set_smem((const void*)spmm_dw128_bf16_kernel<false, true>, smem128, &d128_n);
    if (rc) return rc;
    g.tiles_n = (F + 127) / 128;
    const int grid128 = 8 * g.tiles_m * ((g.tiles_n + 7) / 8);
    if (t_val)
      hipLaunchKernelGGL((spmm_dw128_bf16_kernel<true, true>), dim3(grid128), dim3(256), smem128, (hipStream_t)stream_, g);
    else
      hipLaunchKernelGGL((spmm_dw128_bf16_kernel<false, true>), dim3(grid128), dim3(256), smem128, (hipStream_t)stream_, g);
    SCP_LAUNCH_CHECK();
    return SCP_OK;
  }
  int grid = 8 * g.tiles_m * ((g.tiles_n + 7) / 8);
  if (t_val)
    hipLaunchKernelGGL((spmm_dw_bf16_kernel<true, true>), dim3(grid), dim3(512), smem, (hipStream_t)stream_, g);
  else
    hipLaunchKernelGGL((spmm_dw_bf16_kernel<false, true>), dim3(grid), dim3(512), smem, (hipStream_t)stream_, g);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

extern "C" int scp_spmm_dw_bf16_g16(int ncells, int N, int F, const int32_t* t_ptr, const int32_t* t_key,
                                    const float* t_val, const uint32_t* t_pk, const void* dYT, long ldt, void* dW, long ldw,
                                    void* stream_) {
  SCP_CHECK(ncells > 0 && N > 0 && F > 0, SCP_ERR_ARG, "scp_spmm_dw_bf16_g16: sizes must be positive");
  SCP_CHECK(t_ptr && t_key && dYT && dW && ldw >= F, SCP_ERR_ARG, "scp_spmm_dw_bf16_g16: bad argument");
  SCP_CHECK(!t_val || t_pk, SCP_ERR_ARG, "scp_spmm_dw_bf16_g16: value-carrying tile lists need the packed entries (t_pk) scp_tiles_build writes");
  SCP_CHECK(ldt % 8 == 0 && ldt >= ((ncells + 63) / 64) * 64 && ((uintptr_t)dYT % 16) == 0, SCP_ERR_ARG,
            "scp_spmm_dw_bf16_g16: dYT needs a 16-byte aligned base, ld %% 8 == 0 and ld >= cells rounded up to 64");
  SCP_CHECK((ncells + 63) / 64 <= PCW + 1, SCP_ERR_SHAPE, "scp_spmm_dw_bf16_g16: more than %d cells per call unsupported",
            64 * (PCW + 1));
  static uint64_t done_v = 0, done_n = 0;
  size_t smem = (size_t)(3 * 256 + 2 * 128) * LDK * sizeof(__bf16) + 2 * (PCW + 1) * sizeof(int);
  int rc = t_val ? set_smem((const void*)spmm_dw_bf16_kernel<true, false, true>, smem, &done_v)
                 : set_smem((const void*)spmm_dw_bf16_kernel<false, false, true>, smem, &done_n);
  if (rc) return rc;
  SpArgs g = {};
  g.ncells = ncells; g.N = N; g.F = F; g.ptr = t_ptr; g.key = t_key; g.val = t_val; g.pk = t_pk; g.ncb = (ncells + 63) / 64;
  g.D = (const __bf16*)dYT; g.ldd = ldt; g.C = (float*)dW; g.ldc = ldw;
  g.tiles_m = (N + 127) / 128; g.tiles_n = (F + 255) / 256; g.nsplit = 1;
  int grid = 8 * g.tiles_m * ((g.tiles_n + 7) / 8);
  if (t_val)
    hipLaunchKernelGGL((spmm_dw_bf16_kernel<true, false, true>), dim3(grid), dim3(512), smem, (hipStream_t)stream_, g);
  else
    hipLaunchKernelGGL((spmm_dw_bf16_kernel<false, false, true>), dim3(grid), dim3(512), smem, (hipStream_t)stream_, g);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}
