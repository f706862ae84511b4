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

// one 256-thread workgroup per batch row, four entries in flight per thread
#define TILE_ROW_THREADS 256
__global__ __launch_bounds__(TILE_ROW_THREADS) void tiles_count_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, const float* __restrict__ values,
    const int64_t* __restrict__ rows, int B, int F, int32_t* __restrict__ cnt, float* __restrict__ lib,
    int32_t* __restrict__ nnan) {
  __shared__ float red[16];
  int b = blockIdx.x, tid = threadIdx.x;
  int64_t r = rows[b];
  int64_t e0 = indptr[r], e1 = indptr[r + 1];
  int32_t* cb = cnt + (long)(b >> 6) * F;
  float s = 0.f;
  int nn = 0;
  int64_t e = e0 + tid;
  constexpr int T = TILE_ROW_THREADS;
  for (; e + 3 * T < e1; e += 4 * T) {
    int c0 = indices[e], c1 = indices[e + T], c2 = indices[e + 2 * T], c3 = indices[e + 3 * T];
    float v0 = 1.f, v1 = 1.f, v2 = 1.f, v3 = 1.f;
    if (values) { v0 = values[e]; v1 = values[e + T]; v2 = values[e + 2 * T]; v3 = values[e + 3 * T]; }
    atomicAdd(&cb[c0], 1); atomicAdd(&cb[c1], 1); atomicAdd(&cb[c2], 1); atomicAdd(&cb[c3], 1);
    if (v0 != v0) nn++; else s += v0;
    if (v1 != v1) nn++; else s += v1;
    if (v2 != v2) nn++; else s += v2;
    if (v3 != v3) nn++; else s += v3;
  }
  for (; e < e1; e += T) {
    int col = indices[e];
    atomicAdd(&cb[col], 1);
    float v = values ? values[e] : 1.f;
    if (v != v) nn++; else s += v;
  }
  if (lib) {
    s = block_sum(s, red);
    if (tid == 0) lib[b] = s;
  }
  if (nnan) {
    float t = block_sum((float)nn, red);
    if (tid == 0 && t > 0.f) atomicAdd(nnan, (int)t);
  }
}

__global__ __launch_bounds__(TILE_ROW_THREADS) void tiles_fill_kernel(
    const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices, const float* __restrict__ values,
    const int64_t* __restrict__ rows, int B, int F, int32_t* __restrict__ start /* ptr + 1 : becomes bucket ends */,
    int32_t* __restrict__ key, float* __restrict__ val) {
  int b = blockIdx.x, tid = threadIdx.x;
  int64_t r = rows[b];
  int64_t e0 = indptr[r], e1 = indptr[r + 1];
  int32_t* sb = start + (long)(b >> 6) * F;
  int rl = b & 63;
  int64_t e = e0 + tid;
  constexpr int T = TILE_ROW_THREADS;
  for (; e + 3 * T < e1; e += 4 * T) {
    int c0 = indices[e], c1 = indices[e + T], c2 = indices[e + 2 * T], c3 = indices[e + 3 * T];
    float v0 = 1.f, v1 = 1.f, v2 = 1.f, v3 = 1.f;
    if (val && values) { v0 = values[e]; v1 = values[e + T]; v2 = values[e + 2 * T]; v3 = values[e + 3 * T]; }
    int p0 = atomicAdd(&sb[c0], 1), p1 = atomicAdd(&sb[c1], 1), p2 = atomicAdd(&sb[c2], 1), p3 = atomicAdd(&sb[c3], 1);
    key[p0] = (c0 << 6) | rl; key[p1] = (c1 << 6) | rl; key[p2] = (c2 << 6) | rl; key[p3] = (c3 << 6) | rl;
    if (val) { val[p0] = v0; val[p1] = v1; val[p2] = v2; val[p3] = v3; }
  }
  for (; e < e1; e += T) {
    int col = indices[e];
    int pos = atomicAdd(&sb[col], 1);
    key[pos] = (col << 6) | rl;
    if (val) val[pos] = values ? values[e] : 1.f;
  }
}

// ---- exclusive scan over int32 (3 phases, 4096 elements per block) ------------------------------
#define SCAN_CHUNK 4096
__global__ __launch_bounds__(256) void scan_block_kernel(const int32_t* in, int32_t* out, int32_t* __restrict__ totals,
                                                         long n) {
  __shared__ int32_t wsum[4];
  long base = (long)blockIdx.x * SCAN_CHUNK + threadIdx.x * 16;
  int32_t v[16];
  int32_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    v[i] = (base + i < n) ? in[base + i] : 0;
    s += v[i];
  }
  // inclusive scan of s over the block
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int32_t inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int32_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int32_t woff = 0;
  for (int i = 0; i < w; i++) woff += wsum[i];
  int32_t excl = woff + inc - s;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    if (base + i < n) out[base + i] = excl;
    excl += v[i];
  }
  if (threadIdx.x == 255) totals[blockIdx.x] = woff + inc;
}

__global__ __launch_bounds__(1024) void scan_totals_kernel(int32_t* __restrict__ totals, int nblk) {
  // single block, sequential over chunks of 1024
  __shared__ int32_t wsum[16];
  __shared__ int32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int c0 = 0; c0 < nblk; c0 += 1024) {
    int i = c0 + threadIdx.x;
    int32_t s = (i < nblk) ? totals[i] : 0;
    int32_t inc = s;
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
    if (i < nblk) totals[i] = carry + woff + inc - s;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + inc;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void scan_add_kernel(int32_t* __restrict__ out, const int32_t* __restrict__ totals,
                                                       long n) {
  long base = (long)blockIdx.x * SCAN_CHUNK + threadIdx.x * 16;
  int32_t off = totals[blockIdx.x];
#pragma unroll
  for (int i = 0; i < 16; i++)
    if (base + i < n) out[base + i] += off;
}

extern "C" long scp_tiles_ptr_len(int B, int F) { return (long)((B + 63) / 64) * F + 2; }
extern "C" long scp_tiles_ws_len(int B, int F) {
  long n = (long)((B + 63) / 64) * F + 1;
  return (n + SCAN_CHUNK - 1) / SCAN_CHUNK + 1;
}

extern "C" int scp_tiles_build(const int64_t* indptr, const int32_t* indices, const float* values, const int64_t* rows,
                               int B, int F, int32_t* ptr, int32_t* key, float* val, float* lib, int32_t* nnan,
                               int32_t* ws, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SCP_CHECK(B > 0 && F > 0, SCP_ERR_ARG, "scp_tiles_build: B and F must be positive (%d,%d)", B, F);
  SCP_CHECK(indptr && indices && rows && ptr && key && ws, SCP_ERR_ARG, "scp_tiles_build: null pointer");
  SCP_CHECK(F < (1 << 25), SCP_ERR_SHAPE, "scp_tiles_build: F=%d exceeds the 2^25 key range", F);
  long n = (long)((B + 63) / 64) * F + 1;   // buckets + 1
  // layout: ptr[0] = 0 ; ptr[1 + i] = count(i) -> exclusive scan -> start(i) -> (fill) end(i) = start(i+1)
  SCP_HIP(hipMemsetAsync(ptr, 0, (size_t)(n + 1) * sizeof(int32_t), stream));
  if (nnan) SCP_HIP(hipMemsetAsync(nnan, 0, sizeof(int32_t), stream));
  int grid = B;
  hipLaunchKernelGGL(tiles_count_kernel, dim3(grid), dim3(TILE_ROW_THREADS), 0, stream, indptr, indices, values, rows, B, F,
                     ptr + 1, lib, nnan);
  SCP_LAUNCH_CHECK();
  int nblk = (int)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
  hipLaunchKernelGGL(scan_block_kernel, dim3(nblk), dim3(256), 0, stream, ptr + 1, ptr + 1, ws, n);
  SCP_LAUNCH_CHECK();
  if (nblk > 1) {
    hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(1024), 0, stream, ws, nblk);
    SCP_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_add_kernel, dim3(nblk), dim3(256), 0, stream, ptr + 1, ws, n);
    SCP_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(tiles_fill_kernel, dim3(grid), dim3(TILE_ROW_THREADS), 0, stream, indptr, indices, values, rows, B, F,
                     ptr + 1, key, val);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}
