// Minibatch assembly: CSR rows of a device-resident cell x feature matrix -> cell-block tile list.
//
// The reference gathers dense rows on the CPU per batch (model.py:26-38, 731-732).  Here the whole
// split stays in HBM as CSR and a minibatch is the list of its row ids.  For the MFMA kernels a
// batch is re-bucketed once per step into a "tile list": entries sorted by (64-cell block, feature)
//     ptr[cb * F + f] .. ptr[cb * F + f + 1]   -> entries of feature f inside cell block cb
//     key[e] = (feature << 6) | (cell index inside its block),  val[e] = count
// so that any (cell block x feature range) rectangle is one contiguous entry range, which is what
// the LDS densifiers of the first-layer GEMMs and of the decoder label tiles consume.
// Order inside one (cb, f) bucket is arbitrary (atomic fill) but harmless: every entry lands on a
// distinct (cell, feature) position of a dense LDS tile, so downstream sums stay deterministic.
// The CSR must be canonical (no duplicate (row, col) entries).
#include "common.h"
#include "scpair_hip.h"

// Build, without floating-point atomics (integer atomics only: the result does not depend on their order):
//   1. tiles_rowscan_kernel one workgroup per batch row: one streaming pass over the row's (column-sorted) indices finds
//                           where every feature slice starts (neighbouring entries that fall into different slices),
//                           integer atomics add the slice lengths to the (cell block, slice) totals; the same pass sums
//                           the row's values (library size, fixed order) and counts its NaNs when asked for
//   2. tiles_scan_kernel    one workgroup: exclusive scan of the (cell block, slice) totals -> slice bases; NaN total
//   3. tiles_place_kernel   one workgroup per (slice, cell block): per-feature counts, their scan and the ranks of
//                           the entries inside a bucket all live in LDS; writes ptr, key and val
#define TILE_MAX_SF 2048   // features per slice (LDS histogram), a power of two >= 256

static int tiles_sf(int B, int F) {
  int ncb = (B + 63) / 64, sf = TILE_MAX_SF;
  while (sf > 256 && (long)ncb * ((F + sf - 1) / sf) < 768) sf >>= 1;
  return sf;
}

__global__ __launch_bounds__(256) void tiles_rowscan_kernel(const int64_t* __restrict__ indptr,
                                                            const int32_t* __restrict__ indices,
                                                            const float* __restrict__ values,
                                                            const int64_t* __restrict__ rows, int lsf /* log2 sf */,
                                                            int nslice, int32_t* __restrict__ bnd /* [B][nslice+1] */,
                                                            int32_t* __restrict__ tot /* [ncb][nslice], zeroed */,
                                                            float* __restrict__ lib, int32_t* __restrict__ nnrow) {
  __shared__ float red[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t r = rows[b];
  const int64_t e0 = indptr[r];
  const int len = (int)(indptr[r + 1] - e0);
  const int32_t* ix = indices + e0;
  int32_t* bo = bnd + (long)b * (nslice + 1);
  // entry e opens the slices (slice of entry e-1, slice of entry e]
  {
    int e = tid;
    for (; e + 768 < len; e += 1024) {             // four entries (and their left neighbours) per trip: eight loads in flight
      int cu[4], pv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        cu[u] = ix[e + 256 * u];
        pv[u] = ix[max(e + 256 * u - 1, 0)];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int ee = e + 256 * u;
        const int cur = cu[u] >> lsf, prev = ee ? (pv[u] >> lsf) : -1;
        for (int sidx = prev + 1; sidx <= cur; sidx++) bo[sidx] = ee;
      }
    }
    for (; e < len; e += 256) {
      const int cur = ix[e] >> lsf;
      const int prev = e ? (ix[e - 1] >> lsf) : -1;
      for (int sidx = prev + 1; sidx <= cur; sidx++) bo[sidx] = e;
    }
  }
  const int last = len ? (ix[len - 1] >> lsf) : -1;
  for (int sidx = last + 1 + tid; sidx <= nslice; sidx += 256) bo[sidx] = len;
  float s = 0.f;
  int nn = 0;
  const bool want_sum = lib || (nnrow && values);
  if (want_sum && values) {
    const float* vx = values + e0;
    int e = tid;
    for (; e + 768 < len; e += 1024) {
      float v0 = vx[e], v1 = vx[e + 256], v2 = vx[e + 512], v3 = vx[e + 768];
      if (v0 != v0) nn++; else s += v0;
      if (v1 != v1) nn++; else s += v1;
      if (v2 != v2) nn++; else s += v2;
      if (v3 != v3) nn++; else s += v3;
    }
    for (; e < len; e += 256) {
      float v = vx[e];
      if (v != v) nn++; else s += v;
    }
  }
  __syncthreads();                                 // the row's slice starts are complete (and visible to the workgroup)
  const int cb = b >> 6;
  for (int sidx = tid; sidx < nslice; sidx += 256) {
    const int c = bo[sidx + 1] - bo[sidx];
    if (c) atomicAdd(&tot[cb * nslice + sidx], c);
  }
  if (want_sum) {
    if (values) {
      s = block_sum(s, red);
      const float t = block_sum((float)nn, red);
      if (tid == 0 && nnrow) nnrow[b] = (int)t;
    } else {
      s = (float)len;                               // implicit ones cannot hold NaN
      if (tid == 0 && nnrow) nnrow[b] = 0;
    }
    if (lib && tid == 0) lib[b] = s;
  }
}

// exclusive scan of tot[n] in place (single workgroup of 1024); also the two closing ptr entries
__global__ __launch_bounds__(1024) void tiles_scan_kernel(int32_t* __restrict__ tot, int n, int32_t* __restrict__ ptr_end,
                                                          const int32_t* __restrict__ nnrow, int B,
                                                          int32_t* __restrict__ nnan) {
  __shared__ int32_t wsum[16];
  __shared__ int32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (nnan) {                                      // total NaN count of the batch (per-row counts: tiles_rowscan_kernel)
    int32_t c = 0;
    if (nnrow)
      for (int i = threadIdx.x; i < B; i += 1024) c += nnrow[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) wsum[w] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
      int32_t t = 0;
      for (int k = 0; k < 16; k++) t += wsum[k];
      nnan[0] = t;
    }
    __syncthreads();
  }
  for (int c0 = 0; c0 < n; c0 += 1024) {
    int i = c0 + threadIdx.x;
    int32_t v = (i < n) ? tot[i] : 0;
    int32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int32_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int32_t woff = 0;
    for (int k = 0; k < w; k++) woff += wsum[k];
    int32_t carry = carry_s;
    if (i < n) tot[i] = carry + woff + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    ptr_end[0] = carry_s;
    ptr_end[1] = carry_s;
  }
}

// packed entry of the bf16 first-layer kernels: (bf16 bits of the value, round to nearest even as the kernels' own
// conversion did) << 16 | low 16 key bits (position inside any feature tile of up to 1024 features)
__device__ __forceinline__ uint32_t tile_pack(int key, float v) {
  const __bf16 b = (__bf16)v;
  return ((uint32_t)__builtin_bit_cast(unsigned short, b) << 16) | ((uint32_t)key & 0xffffu);
}

__global__ __launch_bounds__(256) void tiles_place_kernel(const int64_t* __restrict__ indptr,
                                                          const int32_t* __restrict__ indices,
                                                          const float* __restrict__ values,
                                                          const int64_t* __restrict__ rows, int B, int F, int sf,
                                                          int nslice, const int32_t* __restrict__ bnd,
                                                          const int32_t* __restrict__ base_of /* scanned tot */,
                                                          int32_t* __restrict__ ptr, int32_t* __restrict__ key,
                                                          float* __restrict__ val, uint32_t* __restrict__ pk) {
  __shared__ int32_t cnt[TILE_MAX_SF];
  __shared__ int64_t row_e0[64];
  __shared__ int32_t row_lo[64], row_hi[64];
  __shared__ int32_t wsum[4];
  int sl = blockIdx.x, cb = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int f0 = sl * sf;
  int nf = min(sf, F - f0);
  for (int j = tid; j < sf; j += 256) cnt[j] = 0;
  if (tid < 64) {
    int b = cb * 64 + tid;
    int64_t e0 = 0;
    int lo = 0, hi = 0;
    if (b < B) {
      e0 = indptr[rows[b]];
      lo = bnd[(long)b * (nslice + 1) + sl];
      hi = bnd[(long)b * (nslice + 1) + sl + 1];
    }
    row_e0[tid] = e0; row_lo[tid] = lo; row_hi[tid] = hi;
  }
  __syncthreads();
  // Wave w owns rows w, w+4, ..., w+60.  The first 64 slice entries of each of its 16 rows are loaded ONCE (sixteen
  // independent loads per lane in flight) and stay in registers for both passes; longer rows (rare: a slice holds
  // sf / F of a row) loop over the rest.
  int c[16];
  float v[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int r = w + 4 * k;
    const int e = row_lo[r] + lane;
    const bool ok = e < row_hi[r];
    c[k] = ok ? indices[row_e0[r] + e] : -1;
    v[k] = (ok && (val || pk) && values) ? values[row_e0[r] + e] : 1.f;
  }
  // pass 1: per-feature counts
#pragma unroll
  for (int k = 0; k < 16; k++)
    if (c[k] >= 0) atomicAdd(&cnt[c[k] - f0], 1);
  for (int k = 0; k < 16; k++) {
    const int r = w + 4 * k;
    for (int e = row_lo[r] + 64 + lane; e < row_hi[r]; e += 64) atomicAdd(&cnt[indices[row_e0[r] + e] - f0], 1);
  }
  __syncthreads();
  // exclusive scan of cnt[0..sf): thread t owns E consecutive features
  int E = sf >> 8;
  int32_t loc[TILE_MAX_SF / 256];
  int32_t s = 0;
#pragma unroll
  for (int q = 0; q < TILE_MAX_SF / 256; q++) {
    loc[q] = (q < E) ? cnt[tid * E + q] : 0;
    s += loc[q];
  }
  int32_t inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int32_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int32_t excl = inc - s;
  for (int k = 0; k < w; k++) excl += wsum[k];
  int32_t base = base_of[cb * nslice + sl];
  int32_t* pout = ptr + (long)cb * F + f0;
#pragma unroll
  for (int q = 0; q < TILE_MAX_SF / 256; q++) {
    if (q < E) {
      int j = tid * E + q;
      cnt[j] = excl;               // becomes the bucket's running write position
      if (j < nf) pout[j] = base + excl;
      excl += loc[q];
    }
  }
  __syncthreads();
  // pass 2: place
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int r = w + 4 * k;
    if (c[k] >= 0) {
      const int pos = base + atomicAdd(&cnt[c[k] - f0], 1);
      key[pos] = (c[k] << 6) | r;
      if (val) val[pos] = v[k];
      if (pk) pk[pos] = tile_pack((c[k] << 6) | r, v[k]);
    }
  }
  for (int k = 0; k < 16; k++) {
    const int r = w + 4 * k;
    for (int e = row_lo[r] + 64 + lane; e < row_hi[r]; e += 64) {
      const int col = indices[row_e0[r] + e];
      const int pos = base + atomicAdd(&cnt[col - f0], 1);
      key[pos] = (col << 6) | r;
      const float vv = values ? values[row_e0[r] + e] : 1.f;
      if (val) val[pos] = vv;
      if (pk) pk[pos] = tile_pack((col << 6) | r, vv);
    }
  }
}

extern "C" long scp_tiles_ptr_len(int B, int F) { return (long)((B + 63) / 64) * F + 2; }
extern "C" long scp_tiles_ws_len(int B, int F) {
  int sf = tiles_sf(B, F);
  long nslice = (F + sf - 1) / sf, ncb = (B + 63) / 64;
  return (long)B * (nslice + 1) + ncb * nslice + B + 2;
}

extern "C" int scp_tiles_build(const int64_t* indptr, const int32_t* indices, const float* values, const int64_t* rows,
                               int B, int F, int32_t* ptr, int32_t* key, float* val, uint32_t* pk, float* lib,
                               int32_t* nnan, int32_t* ws, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SCP_CHECK(B > 0 && F > 0, SCP_ERR_ARG, "scp_tiles_build: B and F must be positive (%d,%d)", B, F);
  SCP_CHECK(indptr && indices && rows && ptr && key && ws, SCP_ERR_ARG, "scp_tiles_build: null pointer");
  SCP_CHECK(F < (1 << 25), SCP_ERR_SHAPE, "scp_tiles_build: F=%d exceeds the 2^25 key range", F);
  int sf = tiles_sf(B, F);
  int nslice = (F + sf - 1) / sf, ncb = (B + 63) / 64;
  int lsf = 0;
  while ((1 << lsf) < sf) lsf++;
  int32_t* bnd = ws;
  int32_t* tot = ws + (long)B * (nslice + 1);
  int32_t* nnrow = tot + (long)ncb * nslice;
  const bool want_sum = lib || (nnan && values);
  SCP_HIP(hipMemsetAsync(tot, 0, (size_t)ncb * nslice * sizeof(int32_t), stream));
  hipLaunchKernelGGL(tiles_rowscan_kernel, dim3(B), dim3(256), 0, stream, indptr, indices, values, rows, lsf, nslice, bnd,
                     tot, lib, want_sum ? nnrow : (int32_t*)nullptr);
  SCP_LAUNCH_CHECK();
  hipLaunchKernelGGL(tiles_scan_kernel, dim3(1), dim3(1024), 0, stream, tot, ncb * nslice, ptr + (long)ncb * F,
                     want_sum ? nnrow : (const int32_t*)nullptr, B, nnan);
  SCP_LAUNCH_CHECK();
  hipLaunchKernelGGL(tiles_place_kernel, dim3(nslice, ncb), dim3(256), 0, stream, indptr, indices, values, rows, B, F,
                     sf, nslice, bnd, tot, ptr, key, val, (val && values) ? pk : (uint32_t*)nullptr);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

// ---- tile list -> dense [B][ld] fp32 (labels of the likelihoods that need the materialised form: ZINB, Gaussian) ----
__global__ __launch_bounds__(256) void tiles_dense_kernel(const int32_t* __restrict__ ptr, const int32_t* __restrict__ key,
                                                          const float* __restrict__ val, int B, int F,
                                                          float* __restrict__ out, long ld) {
  // one workgroup per (64-cell block, slice of the block's entry range)
  int cb = blockIdx.y;
  int e0 = ptr[(long)cb * F], e1 = ptr[(long)(cb + 1) * F];
  for (int e = e0 + blockIdx.x * 256 + threadIdx.x; e < e1; e += gridDim.x * 256) {
    int k = key[e];
    int c = cb * 64 + (k & 63);
    if (c < B) out[(long)c * ld + (k >> 6)] = val ? val[e] : 1.f;
  }
}

extern "C" int scp_tiles_to_dense(int B, int F, const int32_t* ptr, const int32_t* key, const float* val, float* out,
                                  long ld, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SCP_CHECK(B > 0 && F > 0 && ptr && key && out && ld >= F, SCP_ERR_ARG, "scp_tiles_to_dense: bad argument");
  SCP_HIP(hipMemset2DAsync(out, (size_t)ld * sizeof(float), 0, (size_t)F * sizeof(float), (size_t)B, stream));
  int ncb = (B + 63) / 64;
  hipLaunchKernelGGL(tiles_dense_kernel, dim3(64, ncb), dim3(256), 0, stream, ptr, key, val, B, F, out, ld);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

// ---- optional input pre-transform, applied to the minibatch tile list in place (opt-in; the reference feeds raw
//      counts, so the default is off): mode 1: v <- log1p(v); mode 2: v <- log1p(v * scale / lib[cell]) with lib the
//      cell's library size.  Only the batch's own entries are touched; no normalised matrix is ever materialised. ----
__global__ __launch_bounds__(256) void tiles_transform_kernel(const int32_t* __restrict__ ptr, const int32_t* __restrict__ key,
                                                              float* __restrict__ val, uint32_t* __restrict__ pk, int B,
                                                              int F, const float* __restrict__ lib, int mode, float scale) {
  int cb = blockIdx.y;
  int e0 = ptr[(long)cb * F], e1 = ptr[(long)(cb + 1) * F];
  for (int e = e0 + blockIdx.x * 256 + threadIdx.x; e < e1; e += gridDim.x * 256) {
    float v = val[e];
    if (mode == 2) {
      int c = cb * 64 + (key[e] & 63);
      float l = (c < B) ? lib[c] : 1.f;
      v = (l > 0.f) ? v * (scale / l) : v;
    }
    v = log1pf(v);
    val[e] = v;
    if (pk) pk[e] = tile_pack(key[e], v);
  }
}

extern "C" int scp_tiles_transform(int B, int F, const int32_t* ptr, const int32_t* key, float* val, uint32_t* pk,
                                   const float* lib, int mode, float scale, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SCP_CHECK(B > 0 && F > 0 && ptr && key && val, SCP_ERR_ARG, "scp_tiles_transform: bad argument (values are required)");
  SCP_CHECK(mode == 1 || (mode == 2 && lib && scale > 0.f), SCP_ERR_ARG,
            "scp_tiles_transform: mode must be 1 (log1p) or 2 (log1p of library-size normalised counts, needs lib and scale > 0)");
  int ncb = (B + 63) / 64;
  hipLaunchKernelGGL(tiles_transform_kernel, dim3(64, ncb), dim3(256), 0, stream, ptr, key, val, pk, B, F, lib, mode, scale);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}
