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
#include "common.h"
#include "scpair_hip.h"

#define LDK 72
#define PF 4   // sparse entries prefetched per thread per K-step
#define PCH 255 // K-steps whose entry offsets are staged in LDS at a time

struct SpArgs {
  int ncells, N, F;          // cells, hidden width, features
  const int32_t* ptr; const int32_t* key; const float* val;
  int ncb;
  const __bf16* D; long ldd; // dense bf16 operand: forward Wbf[N][ldd] ; wgrad dYT[N][ldd]
  float* C; long ldc;        // forward: slabs [nsplit][ncells][ldc] ; wgrad: dW[N][ldc]
  int nsplit, ksplit_len, tiles_m, tiles_n;
};

__device__ __forceinline__ uint4 ld16(const __bf16* p, bool ok) {
  uint4 z = {0u, 0u, 0u, 0u};
  return ok ? *(const uint4*)p : z;
}

// ------------------------------------------------------------------------------------------------------
// forward: tile 256 cells x 128 hidden x 64 features, 8 waves as 4 (cells) x 2 (hidden)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void spmm_fwd_bf16_kernel(SpArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* As = (__bf16*)smem_raw;        // [256][LDK]
  __bf16* Bs = As + 256 * LDK;           // [128][LDK]
  int* ptrs = (int*)(Bs + 128 * LDK);    // [4][PCH + 1] entry offsets of the next PCH K-steps, per cell block
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

  // sparse side: 4 cell blocks, 128 threads each
  const int cbl = tid >> 7, tl = tid & 127;
  const int cb = (m0 >> 6) + cbl;
  const bool cb_ok = cb < g.ncb;
  const long pbase = (long)cb * g.F;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int jj = 0; jj < 2; jj++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][jj][e] = 0.f;

  uint4 breg[2];
  int ekey[PF];
  float eval_[PF];
  int p0 = 0, p1 = 0;

  auto prefetch = [&](int k0) {
    int kend = min(k0 + 64, kend_all);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      int c = tid + 512 * i;
      int row = c >> 3, kc = c & 7;
      int n = n0 + row, k = k0 + kc * 8;
      breg[i] = ld16(g.D + (long)n * g.ldd + k, n < g.N && k < g.ldd);
    }
    {
      int si = ((k0 - kbeg) >> 6) % PCH;
      p0 = ptrs[cbl * (PCH + 1) + si];
      p1 = ptrs[cbl * (PCH + 1) + si + 1];
    }
#pragma unroll
    for (int i = 0; i < PF; i++) {
      int e = p0 + tl + 128 * i;
      ekey[i] = (e < p1) ? g.key[e] : -1;
      eval_[i] = (e < p1 && g.val) ? g.val[e] : 1.f;
    }
  };

  // entry offsets of K-steps [c0, c0 + PCH] of this slice -> LDS (one global latency per PCH steps instead of a
  // dependent ptr -> key chain in every step)
  auto load_ptrs = [&](int kfirst) {
    for (int i = tid; i < 4 * (PCH + 1); i += 512) {
      int c = i / (PCH + 1), sidx = i % (PCH + 1);
      int cbx = (m0 >> 6) + c;
      int k = min(kfirst + sidx * 64, kend_all);
      ptrs[i] = (cbx < g.ncb) ? g.ptr[(long)cbx * g.F + k] : 0;
    }
  };
  if (kbeg < kend_all) {
    load_ptrs(kbeg);
    __syncthreads();
    prefetch(kbeg);
  }
  for (int k0 = kbeg; k0 < kend_all; k0 += 64) {
    // zero A, write B
    {
      uint4 z = {0u, 0u, 0u, 0u};
      uint4* q = (uint4*)As;
      for (int i = tid; i < 256 * LDK * 2 / 16; i += 512) q[i] = z;
#pragma unroll
      for (int i = 0; i < 2; i++) {
        int c = tid + 512 * i;
        int row = c >> 3, kc = c & 7;
        *(uint4*)&Bs[row * LDK + kc * 8] = breg[i];
      }
    }
    __syncthreads();
    // scatter the sparse entries of this K-step
#pragma unroll
    for (int i = 0; i < PF; i++) {
      if (ekey[i] >= 0) As[(cbl * 64 + (ekey[i] & 63)) * LDK + ((ekey[i] >> 6) - k0)] = (__bf16)eval_[i];
    }
    for (int e = p0 + tl + 128 * PF; e < p1; e += 128) {
      int key = g.key[e];
      float v = g.val ? g.val[e] : 1.f;
      As[(cbl * 64 + (key & 63)) * LDK + ((key >> 6) - k0)] = (__bf16)v;
    }
    __syncthreads();
    if (k0 + 64 < kend_all) {
      if ((((k0 + 64 - kbeg) >> 6) % PCH) == 0) {     // next step starts a new chunk of offsets (uniform)
        load_ptrs(k0 + 64);
        __syncthreads();
      }
      prefetch(k0 + 64);
    }
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        af[i] = *(const bf16x8*)&As[(wm * 64 + i * 32 + r) * LDK + kk * 16 + hq * 8];
        bfr[i] = *(const bf16x8*)&Bs[(wn * 64 + i * 32 + r) * LDK + kk * 16 + hq * 8];
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int jj = 0; jj < 2; jj++) acc[i][jj] = mfma32(af[i], bfr[jj], acc[i][jj]);
    }
    __syncthreads();
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
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void spmm_dw_bf16_kernel(SpArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* As = (__bf16*)smem_raw;        // [128][LDK]   dYT tile  [hidden][cell]
  __bf16* Bs = As + 128 * LDK;           // [256][LDK]   X tile    [feature][cell]
  int* ptrs = (int*)(Bs + 256 * LDK);    // [2][PCH + 1]: entry range [start, end) of this feature tile per cell block
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

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int jj = 0; jj < 2; jj++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][jj][e] = 0.f;

  uint4 areg[2];
  int ekey[PF];
  float eval_[PF];
  int p0 = 0, p1 = 0;

  auto prefetch = [&](int cb) {
    int k0 = cb * 64;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      int c = tid + 512 * i;
      int row = c >> 3, kc = c & 7;
      int m = m0 + row, k = k0 + kc * 8;
      areg[i] = ld16(g.D + (long)m * g.ldd + k, m < g.N && k < g.ldd);
    }
    p0 = ptrs[cb % (PCH + 1)];
    p1 = ptrs[(PCH + 1) + cb % (PCH + 1)];
#pragma unroll
    for (int i = 0; i < PF; i++) {
      int e = p0 + tid + 512 * i;
      ekey[i] = (e < p1) ? g.key[e] : -1;
      eval_[i] = (e < p1 && g.val) ? g.val[e] : 1.f;
    }
  };

  auto load_ptrs = [&](int cbfirst) {
    for (int i = tid; i < 2 * (PCH + 1); i += 512) {
      int which = i / (PCH + 1), cbx = cbfirst + i % (PCH + 1);
      ptrs[i] = (cbx < g.ncb) ? g.ptr[(long)cbx * g.F + (which ? nend : n0)] : 0;
    }
  };
  load_ptrs(0);
  __syncthreads();
  prefetch(0);
  for (int cb = 0; cb < g.ncb; cb++) {
    {
      uint4 z = {0u, 0u, 0u, 0u};
      uint4* q = (uint4*)Bs;
      for (int i = tid; i < 256 * LDK * 2 / 16; i += 512) q[i] = z;
#pragma unroll
      for (int i = 0; i < 2; i++) {
        int c = tid + 512 * i;
        int row = c >> 3, kc = c & 7;
        *(uint4*)&As[row * LDK + kc * 8] = areg[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PF; i++) {
      if (ekey[i] >= 0) Bs[((ekey[i] >> 6) - n0) * LDK + (ekey[i] & 63)] = (__bf16)eval_[i];
    }
    for (int e = p0 + tid + 512 * PF; e < p1; e += 512) {
      int key = g.key[e];
      float v = g.val ? g.val[e] : 1.f;
      Bs[((key >> 6) - n0) * LDK + (key & 63)] = (__bf16)v;
    }
    __syncthreads();
    if (cb + 1 < g.ncb) {
      if (((cb + 1) % (PCH + 1)) == 0) {
        load_ptrs(cb + 1);
        __syncthreads();
      }
      prefetch(cb + 1);
    }
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        af[i] = *(const bf16x8*)&As[(wm * 64 + i * 32 + r) * LDK + kk * 16 + hq * 8];
        bfr[i] = *(const bf16x8*)&Bs[(wn * 64 + i * 32 + r) * LDK + kk * 16 + hq * 8];
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int jj = 0; jj < 2; jj++) acc[i][jj] = mfma32(af[i], bfr[jj], acc[i][jj]);
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int jj = 0; jj < 2; jj++) {
      int n = n0 + wn * 64 + jj * 32 + r;
      if (n >= g.F) continue;
#pragma unroll
      for (int e = 0; e < 16; e++) {
        int m = m0 + wm * 64 + i * 32 + acc_row(e, lane);
        if (m < g.N) g.C[(long)m * g.ldc + n] = acc[i][jj][e];
      }
    }
}

// ------------------------------------------------------------------------------------------------------
// helpers: fp32 -> bf16 cast into a padded shadow, and fp32 [rows][cols] -> bf16 transposed [cols][ldt]
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, long rows, long cols, long lds_,
                                                        __bf16* __restrict__ dst, long ldd) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  long n = rows * ldd;
  if (i >= n) return;
  long rr = i / ldd, c = i % ldd;
  dst[i] = (c < cols) ? (__bf16)src[rr * lds_ + c] : (__bf16)0.f;
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

extern "C" int scp_cast_bf16(const float* src, long rows, long cols, long ld_src, void* dst, long ld_dst, void* stream_) {
  SCP_CHECK(src && dst && rows > 0 && cols > 0 && ld_dst >= cols, SCP_ERR_ARG, "scp_cast_bf16: bad argument");
  long n = rows * ld_dst;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, src, rows,
                     cols, ld_src, (__bf16*)dst, ld_dst);
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

static int set_smem(const void* fn, size_t smem, bool* done) {
  if (!*done) {
    SCP_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    *done = true;
  }
  return SCP_OK;
}

extern "C" int scp_spmm_fwd_bf16(int ncells, int N, int F, const int32_t* t_ptr, const int32_t* t_key, const float* t_val,
                                 const void* Wbf, long ldw, float* Y, long ldy, int nsplit, void* stream_) {
  SCP_CHECK(ncells > 0 && N > 0 && F > 0, SCP_ERR_ARG, "scp_spmm_fwd_bf16: sizes must be positive");
  SCP_CHECK(t_ptr && t_key && Wbf && Y, SCP_ERR_ARG, "scp_spmm_fwd_bf16: null pointer");
  SCP_CHECK(ldw % 8 == 0 && ldw >= F && ((uintptr_t)Wbf % 16) == 0, SCP_ERR_ARG,
            "scp_spmm_fwd_bf16: shadow weights need a 16-byte aligned base and ld %% 8 == 0");
  static bool done = false;
  size_t smem = (size_t)(256 + 128) * LDK * sizeof(__bf16) + 4 * (PCH + 1) * sizeof(int);
  int rc = set_smem((const void*)spmm_fwd_bf16_kernel, smem, &done);
  if (rc) return rc;
  SpArgs g = {};
  g.ncells = ncells; g.N = N; g.F = F; g.ptr = t_ptr; g.key = t_key; g.val = t_val; g.ncb = (ncells + 63) / 64;
  g.D = (const __bf16*)Wbf; g.ldd = ldw; g.C = Y; g.ldc = ldy;
  g.tiles_m = (ncells + 255) / 256; g.tiles_n = (N + 127) / 128;
  int ksteps = (F + 63) / 64;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > ksteps) nsplit = ksteps;
  int per = (ksteps + nsplit - 1) / nsplit;
  g.ksplit_len = per * 64;
  g.nsplit = (ksteps + per - 1) / per;
  int units = g.tiles_n * g.nsplit;
  int grid = 8 * g.tiles_m * ((units + 7) / 8);
  hipLaunchKernelGGL(spmm_fwd_bf16_kernel, dim3(grid), dim3(512), smem, (hipStream_t)stream_, g);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

extern "C" int scp_spmm_dw_bf16(int ncells, int N, int F, const int32_t* t_ptr, const int32_t* t_key, const float* t_val,
                                const void* dYT, long ldt, float* dW, long ldw, void* stream_) {
  SCP_CHECK(ncells > 0 && N > 0 && F > 0, SCP_ERR_ARG, "scp_spmm_dw_bf16: sizes must be positive");
  SCP_CHECK(t_ptr && t_key && dYT && dW, SCP_ERR_ARG, "scp_spmm_dw_bf16: null pointer");
  SCP_CHECK(ldt % 8 == 0 && ldt >= ((ncells + 63) / 64) * 64 && ((uintptr_t)dYT % 16) == 0, SCP_ERR_ARG,
            "scp_spmm_dw_bf16: dYT needs a 16-byte aligned base, ld %% 8 == 0 and ld >= cells rounded up to 64");
  static bool done = false;
  size_t smem = (size_t)(256 + 128) * LDK * sizeof(__bf16) + 2 * (PCH + 1) * sizeof(int);
  int rc = set_smem((const void*)spmm_dw_bf16_kernel, smem, &done);
  if (rc) return rc;
  SpArgs g = {};
  g.ncells = ncells; g.N = N; g.F = F; g.ptr = t_ptr; g.key = t_key; g.val = t_val; g.ncb = (ncells + 63) / 64;
  g.D = (const __bf16*)dYT; g.ldd = ldt; g.C = dW; g.ldc = ldw;
  g.tiles_m = (N + 127) / 128; g.tiles_n = (F + 255) / 256; g.nsplit = 1;
  int grid = 8 * g.tiles_m * ((g.tiles_n + 7) / 8);
  hipLaunchKernelGGL(spmm_dw_bf16_kernel, dim3(grid), dim3(512), smem, (hipStream_t)stream_, g);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}
