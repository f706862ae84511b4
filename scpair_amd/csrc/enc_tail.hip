// Encoder "tail": everything of Input_Module (model.py:60-143) between the wide first-layer product and the latent
// output, in one forward kernel and three backward kernels instead of ~30 launches:
//
//   slabs (split-K partials of x W1^T)  -> + b1 -> LayerNorm -> LeakyReLU -> Dropout                  (utils.py:57-63)
//                                       -> Linear(H1, H2) -> LayerNorm -> LeakyReLU -> Dropout         (second FCNN layer)
//                                       -> Linear(H2, N3)   (extra_linear_rep, model.py:83-84; the library head
//                                                            Linear(H2, 1), model.py:92; N3 = 0: none)
//
// The work is row-local and tiny (2 B H1 H2 flop), so it runs in exact fp32 on the vector units: a half-wave (32 lanes)
// owns one cell, lane l holds columns 128 c + 4 l .. + 3 of the H1-wide row in registers (c < 8: H1 <= 1024,
// H1 % 4 == 0), the H2-wide row sits one or two values per lane (H2 <= 64), row reductions are 5-step butterflies
// inside the half-wave.  W2 is streamed from L2 as float4 rows.  Backward: a row kernel (same layout: gradients of
// the small tensors as per-workgroup partials, dY1 = gradient of the first-layer product, dz2), a column kernel for
// dW2 = dz2^T a1 (a1 recomputed from the saved xhat1), and a fixed-order reduction of the partials.
#include "common.h"
#include "scpair_hip.h"

#define ET_NCH 8
#define ET_ROWS 8
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float leaky_(float z, float slope) { return z > 0.f ? z : z * slope; }
__device__ __forceinline__ float f4sum(f4 v) { return (v.x + v.y) + (v.z + v.w); }
__device__ __forceinline__ f4 keep4(uint64_t seed, uint64_t idx, float p) {
  f4 k;
  k.x = keep_scale(seed, idx, p); k.y = keep_scale(seed, idx + 1, p);
  k.z = keep_scale(seed, idx + 2, p); k.w = keep_scale(seed, idx + 3, p);
  return k;
}
__device__ __forceinline__ f4 leaky4(f4 z, float slope) {
  f4 r;
  r.x = leaky_(z.x, slope); r.y = leaky_(z.y, slope); r.z = leaky_(z.z, slope); r.w = leaky_(z.w, slope);
  return r;
}
// dz = z > 0 ? g : g * slope, element-wise
__device__ __forceinline__ f4 dleaky4(f4 z, f4 gq, float slope) {
  f4 r;
  r.x = z.x > 0.f ? gq.x : gq.x * slope; r.y = z.y > 0.f ? gq.y : gq.y * slope;
  r.z = z.z > 0.f ? gq.z : gq.z * slope; r.w = z.w > 0.f ? gq.w : gq.w * slope;
  return r;
}

// Hot loops carry no branch: the chunk count is a template parameter and the loops over the H2 rows of W2 are plain
// runtime loops (an `if` inside an unrolled loop, even a uniform one, ends a basic block, and the loads of a block
// are waited for at its end -- the ~400 L2 loads of a lane then run one after the other).  Lanes of the last chunk
// that lie past H1 read a clamped, valid address and mask the value.
struct TailOff {
  int g1, be1, b1, g2, be2, b2, w3, b3, pv;
};
// Layout of one workgroup's partial-gradient record (PV floats, PV % 4 == 0):
//   dgamma1[H1] dbeta1[H1] dbias1[H1] dgamma2[H2] dbeta2[H2] dbias2[H2] dW3[N3*H2] db3[N3]
static __host__ __device__ inline TailOff tail_off(int H1, int H2, int N3) {
  TailOff o;
  o.g1 = 0; o.be1 = H1; o.b1 = 2 * H1; o.g2 = 3 * H1; o.be2 = o.g2 + H2; o.b2 = o.be2 + H2; o.w3 = o.b2 + H2;
  o.b3 = o.w3 + N3 * H2;
  o.pv = (o.b3 + N3 + 3) / 4 * 4;
  return o;
}

// ---------------------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void enc_tail_fwd_kernel(ScpEncTail m, const float* __restrict__ slabs, long slab_stride,
                                                           int nslab, float* __restrict__ out, long ldo) {
  __shared__ float red[ET_ROWS][32][33];         // partial dot products of 32 rows of W2: [cell][j][lane]
  const int tid = threadIdx.x, lane = tid & 31, sub = tid >> 5;
  const int row_ = blockIdx.x * ET_ROWS + sub;
  const bool live = row_ < m.B;
  const int row = live ? row_ : m.B - 1;        // dead half-waves recompute the last row and store nothing
  const int H1 = m.H1, H2 = m.H2, N3 = m.N3;
  const uint64_t sd1 = m.seed1 + (m.seed_dev ? m.seed_dev[0] : 0), sd2 = m.seed2 + (m.seed_dev ? m.seed_dev[0] : 0);
  const f4 zero4 = f4{0.f, 0.f, 0.f, 0.f};
  bool ok[NCH];
  int ko[NCH];                                  // clamped column offset of this lane's float4 in chunk c
  f4 v[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int k = c * 128 + lane * 4;
    ok[c] = k < H1;
    ko[c] = min(k, H1 - 4);
    v[c] = zero4;
  }
  if (m.b1) {
#pragma unroll
    for (int c = 0; c < NCH; c++) v[c] = *(const f4*)(m.b1 + ko[c]);
  }
  f4 gam[NCH], bet[NCH];                        // issued ahead of the slab loads: nothing later waits on them alone
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    gam[c] = *(const f4*)(m.g1 + ko[c]);
    bet[c] = *(const f4*)(m.be1 + ko[c]);
  }
  // split-K slabs of the first-layer product, summed in ascending order; four slabs of loads in flight
  {
    const float* base = slabs + (long)row * H1;
    int sl = 0;
    for (; sl + 4 <= nslab; sl += 4) {
      f4 t[4][NCH];
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int c = 0; c < NCH; c++) t[u][c] = *(const f4*)(base + (long)(sl + u) * slab_stride + ko[c]);
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int c = 0; c < NCH; c++) v[c] += t[u][c];
    }
    for (; sl < nslab; sl++) {
      f4 t[NCH];
#pragma unroll
      for (int c = 0; c < NCH; c++) t[c] = *(const f4*)(base + (long)sl * slab_stride + ko[c]);
#pragma unroll
      for (int c = 0; c < NCH; c++) v[c] += t[c];
    }
  }
#pragma unroll
  for (int c = 0; c < NCH; c++) v[c] = ok[c] ? v[c] : zero4;
  // LayerNorm 1 (two-pass statistics, biased variance: nn.LayerNorm)
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; c++) s += f4sum(v[c]);
  const float mean1 = half_sum(s) / (float)H1;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const f4 d = ok[c] ? v[c] - mean1 : zero4;
    q += f4sum(d * d);
  }
  const float rstd1 = 1.f / sqrtf(half_sum(q) / (float)H1 + m.eps1);
  if (lane == 0 && m.rstd1 && live) m.rstd1[row] = rstd1;
  f4 a[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const f4 xh = (v[c] - mean1) * rstd1;
    if (m.xhat1 && ok[c] && live) *(f4*)(m.xhat1 + (long)row * H1 + ko[c]) = xh;
    const f4 z = xh * gam[c] + bet[c];
    const f4 t = leaky4(z, m.slope) * keep4(sd1, (uint64_t)row * H1 + ko[c], m.p1);
    a[c] = ok[c] ? t : zero4;
  }
  // second layer: z2[j] = a1 . W2[j] + b2[j].  Every lane forms its partial dot product with each row of W2 (independent
  // loads, four rows in flight) and parks it in LDS; lane j then adds the 32 partials of row j (and of row j + 32).
  const bool vlo = lane < H2, vhi = lane + 32 < H2;
  float z2lo = 0.f, z2hi = 0.f;
  for (int half = 0; half * 32 < H2; half++) {
    const int jn = min(32, H2 - half * 32);
    if (half) __syncthreads();                  // the first half's readers are done
#pragma unroll 4
    for (int jj = 0; jj < jn; jj++) {
      const float* w = m.W2 + (long)(half * 32 + jj) * H1;
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; c++) acc += f4sum(a[c] * *(const f4*)(w + ko[c]));      // a[c] = 0 in masked lanes
      red[sub][jj][lane] = acc;
    }
    __syncthreads();
    const int jr = min(lane, jn - 1);
    float tot = 0.f;
#pragma unroll
    for (int l = 0; l < 32; l++) tot += red[sub][jr][l];
    tot += m.b2[min(half * 32 + lane, H2 - 1)];
    if (half == 0) z2lo = vlo ? tot : 0.f;
    else z2hi = vhi ? tot : 0.f;
  }
  const float mean2 = half_sum(z2lo + z2hi) / (float)H2;
  const float dlo = vlo ? z2lo - mean2 : 0.f, dhi = vhi ? z2hi - mean2 : 0.f;
  const float rstd2 = 1.f / sqrtf(half_sum(dlo * dlo + dhi * dhi) / (float)H2 + m.eps2);
  if (lane == 0 && m.rstd2 && live) m.rstd2[row] = rstd2;
  const int jlo = min(lane, H2 - 1), jhi = min(lane + 32, H2 - 1);
  const float xhlo = dlo * rstd2, xhhi = dhi * rstd2;
  if (m.xhat2 && live) {
    if (vlo) m.xhat2[(long)row * H2 + lane] = xhlo;
    if (vhi) m.xhat2[(long)row * H2 + lane + 32] = xhhi;
  }
  float x2lo = leaky_(xhlo * m.g2[jlo] + m.be2[jlo], m.slope) * keep_scale(sd2, (uint64_t)row * H2 + jlo, m.p2);
  float x2hi = leaky_(xhhi * m.g2[jhi] + m.be2[jhi], m.slope) * keep_scale(sd2, (uint64_t)row * H2 + jhi, m.p2);
  x2lo = vlo ? x2lo : 0.f;
  x2hi = vhi ? x2hi : 0.f;
  if (N3 == 0) {
    if (live) {
      if (vlo) out[(long)row * ldo + lane] = x2lo;
      if (vhi) out[(long)row * ldo + lane + 32] = x2hi;
    }
    return;
  }
  // final Linear(H2, N3): lane n computes outputs n and n + 32 (clamped rows of W3: the loads stay unpredicated)
  const int nlo = min(lane, N3 - 1), nhi = min(lane + 32, N3 - 1);
  float olo = m.b3[nlo], ohi = m.b3[nhi];
  const float* w3lo = m.W3 + (long)nlo * H2;
  const float* w3hi = m.W3 + (long)nhi * H2;
#pragma unroll 8
  for (int k = 0; k < H2; k++) {
    const float xk = __shfl(k < 32 ? x2lo : x2hi, k & 31, 32);
    olo += xk * w3lo[k];
    ohi += xk * w3hi[k];
  }
  if (live) {
    if (lane < N3) out[(long)row * ldo + lane] = olo;
    if (lane + 32 < N3) out[(long)row * ldo + lane + 32] = ohi;
  }
}

template <int NCH>
__global__ __launch_bounds__(256) void enc_tail_bwd_kernel(ScpEncTail m, const float* __restrict__ dOut, long lddo,
                                                           float* __restrict__ dY1, long ldy, float* __restrict__ dz2g,
                                                           float* __restrict__ part) {
  __shared__ float sh_do[ET_ROWS][64];
  __shared__ float sh_a2[ET_ROWS][64];
  __shared__ float sh_v2[ET_ROWS][3][64];
  __shared__ __attribute__((aligned(16))) float colA[ET_ROWS][NCH * 128];   // one column-sum staging buffer, used 3x
  const int tid = threadIdx.x, lane = tid & 31, sub = tid >> 5;
  const int row_ = blockIdx.x * ET_ROWS + sub;
  const bool live = row_ < m.B;
  const int row = live ? row_ : m.B - 1;          // dead half-waves recompute the last row; their results are zeroed
  const float lv = live ? 1.f : 0.f;
  const int H1 = m.H1, H2 = m.H2, N3 = m.N3;
  const int NO = N3 > 0 ? N3 : H2;                 // width of dOut
  const TailOff off = tail_off(H1, H2, N3);
  const uint64_t sd1 = m.seed1 + (m.seed_dev ? m.seed_dev[0] : 0), sd2 = m.seed2 + (m.seed_dev ? m.seed_dev[0] : 0);
  const f4 zero4 = f4{0.f, 0.f, 0.f, 0.f};
  bool ok[NCH];
  int ko[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    ok[c] = c * 128 + lane * 4 < H1;
    ko[c] = min(c * 128 + lane * 4, H1 - 4);
  }
  const bool vlo = lane < H2, vhi = lane + 32 < H2;
  const int jlo = min(lane, H2 - 1), jhi = min(lane + 32, H2 - 1);
  const int olo_i = min(lane, NO - 1), ohi_i = min(lane + 32, NO - 1);
  const float dolo = lane < NO ? lv * dOut[(long)row * lddo + olo_i] : 0.f;
  const float dohi = lane + 32 < NO ? lv * dOut[(long)row * lddo + ohi_i] : 0.f;
  const float r2 = m.rstd2[row];
  const float xlo = vlo ? m.xhat2[(long)row * H2 + jlo] : 0.f, xhi = vhi ? m.xhat2[(long)row * H2 + jhi] : 0.f;
  const float glo = vlo ? m.g2[jlo] : 0.f, ghi = vhi ? m.g2[jhi] : 0.f;
  const float znlo = xlo * glo + m.be2[jlo], znhi = xhi * ghi + m.be2[jhi];
  const float klo = vlo ? keep_scale(sd2, (uint64_t)row * H2 + jlo, m.p2) : 0.f;
  const float khi = vhi ? keep_scale(sd2, (uint64_t)row * H2 + jhi, m.p2) : 0.f;
  const float a2lo = leaky_(znlo, m.slope) * klo * lv, a2hi = leaky_(znhi, m.slope) * khi * lv;
  float dalo = 0.f, dahi = 0.f;                   // gradient wrt a2
  if (N3 > 0) {
#pragma unroll 8
    for (int n = 0; n < N3; n++) {
      const float dn = __shfl(n < 32 ? dolo : dohi, n & 31, 32);
      dalo += dn * m.W3[(long)n * H2 + jlo];
      dahi += dn * m.W3[(long)n * H2 + jhi];
    }
  } else {
    dalo = dolo;
    dahi = dohi;
  }
  sh_do[sub][lane] = dolo; sh_do[sub][lane + 32] = dohi;
  sh_a2[sub][lane] = a2lo; sh_a2[sub][lane + 32] = a2hi;
  // LayerNorm 2 / activation / dropout backward
  const float gq_lo = dalo * klo, gq_hi = dahi * khi;
  const float dzlo = vlo ? (znlo > 0.f ? gq_lo : gq_lo * m.slope) : 0.f;
  const float dzhi = vhi ? (znhi > 0.f ? gq_hi : gq_hi * m.slope) : 0.f;
  const float dxlo = dzlo * glo, dxhi = dzhi * ghi;
  const float s1 = half_sum(dxlo + dxhi) / (float)H2;
  const float s2 = half_sum(dxlo * xlo + dxhi * xhi) / (float)H2;
  const float d2lo = vlo ? r2 * (dxlo - s1 - xlo * s2) : 0.f, d2hi = vhi ? r2 * (dxhi - s1 - xhi * s2) : 0.f;
  sh_v2[sub][0][lane] = dzlo * xlo; sh_v2[sub][0][lane + 32] = dzhi * xhi;
  sh_v2[sub][1][lane] = dzlo;       sh_v2[sub][1][lane + 32] = dzhi;
  sh_v2[sub][2][lane] = d2lo;       sh_v2[sub][2][lane + 32] = d2hi;
  if (live) {
    if (vlo) dz2g[(long)row * H2 + lane] = d2lo;
    if (vhi) dz2g[(long)row * H2 + lane + 32] = d2hi;
  }
  // gradient wrt a1: da[k] = sum_j dz2[j] W2[j][k]  (rows of W2: independent loads, four rows in flight)
  f4 da[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++) da[c] = zero4;
#pragma unroll 4
  for (int j = 0; j < H2; j++) {
    const float dzj = __shfl(j < 32 ? d2lo : d2hi, j & 31, 32);
    const float* w = m.W2 + (long)j * H1;
#pragma unroll
    for (int c = 0; c < NCH; c++) da[c] += dzj * *(const f4*)(w + ko[c]);
  }
  // LayerNorm 1 / activation / dropout backward
  const float r1 = m.rstd1[row];
  f4 xh[NCH], dxh[NCH], cb[NCH], cd[NCH];
  float t1 = 0.f, t2 = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const f4 xl = *(const f4*)(m.xhat1 + (long)row * H1 + ko[c]);
    const f4 gam = *(const f4*)(m.g1 + ko[c]);
    const f4 z = xl * gam + *(const f4*)(m.be1 + ko[c]);
    const f4 dzl = dleaky4(z, da[c] * keep4(sd1, (uint64_t)row * H1 + ko[c], m.p1), m.slope);
    const f4 dz = ok[c] ? dzl : zero4;           // dead rows: dz2 = 0, hence da = 0 and dz = 0
    xh[c] = ok[c] ? xl : zero4;
    cb[c] = dz;
    dxh[c] = dz * gam;
    t1 += f4sum(dxh[c]);
    t2 += f4sum(dxh[c] * xh[c]);
    *(f4*)&colA[sub][c * 128 + lane * 4] = dz * xh[c];
  }
  t1 = half_sum(t1) / (float)H1;
  t2 = half_sum(t2) / (float)H1;
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    cd[c] = ok[c] ? (dxh[c] - t1 - xh[c] * t2) * r1 : zero4;
    if (ok[c] && live) *(f4*)(dY1 + (long)row * ldy + ko[c]) = cd[c];
  }
  __syncthreads();
  float* rec = part + (long)blockIdx.x * off.pv;
  // column sums over the workgroup's rows (fixed order)
  if (tid * 4 < H1) {
    f4 sa = zero4;
#pragma unroll
    for (int r = 0; r < ET_ROWS; r++) sa += *(const f4*)&colA[r][tid * 4];
    *(f4*)(rec + off.g1 + tid * 4) = sa;
  }
  if (tid < H2) {
    float s0 = 0.f, s1_ = 0.f, s2_ = 0.f;
#pragma unroll
    for (int r = 0; r < ET_ROWS; r++) {
      s0 += sh_v2[r][0][tid];
      s1_ += sh_v2[r][1][tid];
      s2_ += sh_v2[r][2][tid];
    }
    rec[off.g2 + tid] = s0;
    rec[off.be2 + tid] = s1_;
    rec[off.b2 + tid] = s2_;
  }
  for (int idx = tid; idx < N3 * H2; idx += 256) {
    const int n = idx / H2, j = idx - n * H2;
    float s0 = 0.f;
#pragma unroll
    for (int r = 0; r < ET_ROWS; r++) s0 += sh_do[r][n] * sh_a2[r][j];
    rec[off.w3 + idx] = s0;
  }
  if (tid < N3) {
    float s0 = 0.f;
#pragma unroll
    for (int r = 0; r < ET_ROWS; r++) s0 += sh_do[r][tid];
    rec[off.b3 + tid] = s0;
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NCH; c++) *(f4*)&colA[sub][c * 128 + lane * 4] = cb[c];
  __syncthreads();
  if (tid * 4 < H1) {
    f4 sa = zero4;
#pragma unroll
    for (int r = 0; r < ET_ROWS; r++) sa += *(const f4*)&colA[r][tid * 4];
    *(f4*)(rec + off.be1 + tid * 4) = sa;
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NCH; c++) *(f4*)&colA[sub][c * 128 + lane * 4] = cd[c];
  __syncthreads();
  if (tid * 4 < H1) {
    f4 sa = zero4;
#pragma unroll
    for (int r = 0; r < ET_ROWS; r++) sa += *(const f4*)&colA[r][tid * 4];
    *(f4*)(rec + off.b1 + tid * 4) = sa;
  }
}

// dW2 slab of one (128-column block, 64-row chunk): slab[chunk][j][k] = sum over the chunk's rows of dz2[row][j] a1[row][k]
#define ETW_ROWS 64
__global__ __launch_bounds__(256) void enc_tail_w2_kernel(ScpEncTail m, const float* __restrict__ dz2g,
                                                          float* __restrict__ slab) {
  __shared__ float dzs[ETW_ROWS][64];
  __shared__ __attribute__((aligned(16))) float red[8][8][128];
  const int tid = threadIdx.x, kq = tid & 31, slot = tid >> 5;
  const int H1 = m.H1, H2 = m.H2;
  const int k = blockIdx.x * 128 + kq * 4;
  const bool okk = k < H1;
  const int r0 = blockIdx.y * ETW_ROWS;
  const uint64_t sd1 = m.seed1 + (m.seed_dev ? m.seed_dev[0] : 0);
  for (int idx = tid; idx < ETW_ROWS * 64; idx += 256) {
    const int rr = idx >> 6, j = idx & 63;
    dzs[rr][j] = (r0 + rr < m.B && j < H2) ? dz2g[(long)(r0 + rr) * H2 + j] : 0.f;
  }
  const int kc = min(k, H1 - 4);
  const f4 gam = *(const f4*)(m.g1 + kc), bet = *(const f4*)(m.be1 + kc);
  (void)okk;
  __syncthreads();
  for (int pass = 0; pass * 32 < H2; pass++) {
    f4 acc[32];
#pragma unroll
    for (int jj = 0; jj < 32; jj++) acc[jj] = f4{0.f, 0.f, 0.f, 0.f};
    f4 xr[ETW_ROWS / 8];                           // the 8 rows of this slot: loads issued together, unpredicated
#pragma unroll
    for (int i = 0; i < ETW_ROWS / 8; i++) {
      const int row = min(r0 + slot + 8 * i, m.B - 1);
      xr[i] = *(const f4*)(m.xhat1 + (long)row * H1 + kc);
    }
#pragma unroll
    for (int i = 0; i < ETW_ROWS / 8; i++) {
      const int rr = slot + 8 * i, row = r0 + rr;
      // rows past the batch end have dz2 = 0 in dzs; columns past H1 are never written out
      const f4 a1 = leaky4(xr[i] * gam + bet, m.slope) * keep4(sd1, (uint64_t)min(row, m.B - 1) * H1 + kc, m.p1);
#pragma unroll
      for (int jj = 0; jj < 32; jj++) acc[jj] += dzs[rr][pass * 32 + jj] * a1;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      __syncthreads();
#pragma unroll
      for (int j8 = 0; j8 < 8; j8++) *(f4*)&red[slot][j8][kq * 4] = acc[q * 8 + j8];
      __syncthreads();
      const int j8 = tid >> 5, kk = tid & 31;           // one (j, 4 columns) item per thread
      const int j = pass * 32 + q * 8 + j8, ko = blockIdx.x * 128 + kk * 4;
      if (j < H2 && ko < H1) {
        f4 s = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sl = 0; sl < 8; sl++) s += *(const f4*)&red[sl][j8][kk * 4];
        *(f4*)(slab + ((long)blockIdx.y * H2 + j) * H1 + ko) = s;
      }
    }
  }
}

// out = sum over records / slabs, in ascending order
struct TailRed {
  const float* part; int nblk; int pv;
  const float* slab; int nchunk; long wn;
  int nseg; int seg_beg[8]; int seg_end[8]; float* seg_dst[8];
  float* dW2;
};
__global__ __launch_bounds__(256) void enc_tail_reduce_kernel(TailRed r) {
  // 64 outputs per workgroup, 4 threads per output: thread (o, q) sums records q, q + 4, ... (eight loads in flight),
  // the four partial sums are combined in a fixed order
  __shared__ float red[4][64];
  const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + o;
  const bool in_part = i < r.pv;
  const long w = i - r.pv;
  const float* src = in_part ? r.part + i : r.slab + w;
  const long stride = in_part ? r.pv : r.wn;
  const int n = in_part ? r.nblk : ((w < r.wn) ? r.nchunk : 0);
  float s = 0.f;
  int b = q;
  for (; b + 28 < n; b += 32) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; u++) t[u] = src[(long)(b + 4 * u) * stride];
#pragma unroll
    for (int u = 0; u < 8; u++) s += t[u];
  }
  for (; b < n; b += 4) s += src[(long)b * stride];
  red[q][o] = s;
  __syncthreads();
  if (q != 0 || n == 0) return;
  s = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
  if (in_part) {
    for (int g_ = 0; g_ < r.nseg; g_++)
      if (i >= r.seg_beg[g_] && i < r.seg_end[g_] && r.seg_dst[g_]) r.seg_dst[g_][i - r.seg_beg[g_]] = s;
  } else if (r.dW2) {
    r.dW2[w] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------
static int tail_check(const ScpEncTail* m, const char* who) {
  SCP_CHECK(m, SCP_ERR_ARG, "%s: null module description", who);
  SCP_CHECK(m->B > 0 && m->H1 > 0 && m->H2 > 0 && m->N3 >= 0, SCP_ERR_ARG, "%s: sizes must be positive", who);
  SCP_CHECK(m->H1 % 4 == 0 && m->H1 >= 4 && m->H1 <= ET_NCH * 128 && m->H2 <= 64 && m->N3 <= 64, SCP_ERR_SHAPE,
            "%s: needs H1 %% 4 == 0, H1 <= %d, H2 <= 64, N3 <= 64 (got %d, %d, %d)", who, ET_NCH * 128, m->H1, m->H2, m->N3);
  SCP_CHECK(m->g1 && m->be1 && m->W2 && m->b2 && m->g2 && m->be2, SCP_ERR_ARG, "%s: null parameter pointer", who);
  SCP_CHECK(m->N3 == 0 || (m->W3 && m->b3), SCP_ERR_ARG, "%s: the final Linear needs weight and bias", who);
  SCP_CHECK((((uintptr_t)m->b1 | (uintptr_t)m->g1 | (uintptr_t)m->be1 | (uintptr_t)m->W2 | (uintptr_t)m->xhat1) & 15) == 0,
            SCP_ERR_SHAPE, "%s: H1-wide tensors must be 16-byte aligned", who);
  SCP_CHECK(m->p1 >= 0.f && m->p1 < 1.f && m->p2 >= 0.f && m->p2 < 1.f, SCP_ERR_ARG, "%s: dropout p must be in [0,1)", who);
  return SCP_OK;
}

extern "C" int scp_enc_tail_fwd(const ScpEncTail* m, const float* slabs, long slab_stride, int nslab, float* out, long ldo,
                                void* stream_) {
  int rc = tail_check(m, "scp_enc_tail_fwd");
  if (rc) return rc;
  SCP_CHECK(slabs && out && nslab > 0 && ldo >= (m->N3 > 0 ? m->N3 : m->H2), SCP_ERR_ARG, "scp_enc_tail_fwd: bad argument");
  SCP_CHECK(((uintptr_t)slabs & 15) == 0 && slab_stride % 4 == 0, SCP_ERR_SHAPE,
            "scp_enc_tail_fwd: slabs must be 16-byte aligned with a stride that is a multiple of 4");
  SCP_CHECK((m->xhat1 == nullptr) == (m->rstd1 == nullptr) && (m->xhat2 == nullptr) == (m->rstd2 == nullptr) &&
                (m->xhat1 == nullptr) == (m->xhat2 == nullptr),
            SCP_ERR_ARG, "scp_enc_tail_fwd: the four saved-for-backward buffers go together");
  const int grid = (m->B + ET_ROWS - 1) / ET_ROWS;
#define ET_FWD(N)                                                                                                     \
  case N:                                                                                                             \
    hipLaunchKernelGGL(enc_tail_fwd_kernel<N>, dim3(grid), dim3(256), 0, (hipStream_t)stream_, *m, slabs, slab_stride, \
                       nslab, out, ldo);                                                                              \
    break;
  switch ((m->H1 + 127) / 128) { ET_FWD(1) ET_FWD(2) ET_FWD(3) ET_FWD(4) ET_FWD(5) ET_FWD(6) ET_FWD(7) ET_FWD(8) }
#undef ET_FWD
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

extern "C" long scp_enc_tail_ws_floats(int B, int H1, int H2, int N3) {
  const TailOff off = tail_off(H1, H2, N3);
  const long nblk = (B + ET_ROWS - 1) / ET_ROWS, nchunk = (B + ETW_ROWS - 1) / ETW_ROWS;
  return nblk * off.pv + nchunk * (long)H2 * H1 + (long)B * H2 + 8;
}

extern "C" int scp_enc_tail_bwd(const ScpEncTail* m, const float* dOut, long lddo, float* dY1, long ldy, float* ws,
                                float* dg1, float* dbe1, float* db1, float* dW2, float* dg2, float* dbe2, float* db2,
                                float* dW3, float* db3, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = tail_check(m, "scp_enc_tail_bwd");
  if (rc) return rc;
  SCP_CHECK(dOut && dY1 && ws && m->xhat1 && m->rstd1 && m->xhat2 && m->rstd2, SCP_ERR_ARG, "scp_enc_tail_bwd: null pointer");
  SCP_CHECK(ldy >= m->H1 && ldy % 4 == 0 && (((uintptr_t)dY1 | (uintptr_t)ws) & 15) == 0, SCP_ERR_SHAPE,
            "scp_enc_tail_bwd: dY1 / ws must be 16-byte aligned, ldy %% 4 == 0");
  SCP_CHECK(lddo >= (m->N3 > 0 ? m->N3 : m->H2), SCP_ERR_ARG, "scp_enc_tail_bwd: dOut leading dimension too small");
  const TailOff off = tail_off(m->H1, m->H2, m->N3);
  const int nblk = (m->B + ET_ROWS - 1) / ET_ROWS, nchunk = (m->B + ETW_ROWS - 1) / ETW_ROWS;
  const long wn = (long)m->H2 * m->H1;
  float* part = ws;
  float* slab = part + (long)nblk * off.pv;        // pv % 4 == 0 keeps it 16-byte aligned
  float* dz2 = slab + (long)nchunk * wn;
#define ET_BWD(N)                                                                                                       \
  case N:                                                                                                               \
    hipLaunchKernelGGL(enc_tail_bwd_kernel<N>, dim3(nblk), dim3(256), 0, stream, *m, dOut, lddo, dY1, ldy, dz2, part);   \
    break;
  switch ((m->H1 + 127) / 128) { ET_BWD(1) ET_BWD(2) ET_BWD(3) ET_BWD(4) ET_BWD(5) ET_BWD(6) ET_BWD(7) ET_BWD(8) }
#undef ET_BWD
  SCP_LAUNCH_CHECK();
  if (dW2) {
    hipLaunchKernelGGL(enc_tail_w2_kernel, dim3((m->H1 + 127) / 128, nchunk), dim3(256), 0, stream, *m, dz2, slab);
    SCP_LAUNCH_CHECK();
  }
  TailRed r = {};
  r.part = part; r.nblk = nblk; r.pv = off.pv; r.slab = slab; r.nchunk = nchunk; r.wn = wn; r.dW2 = dW2;
  const int begs[8] = {off.g1, off.be1, off.b1, off.g2, off.be2, off.b2, off.w3, off.b3};
  const int ends[8] = {off.be1, off.b1, off.g2, off.be2, off.b2, off.w3, off.b3, off.b3 + m->N3};
  float* dsts[8] = {dg1, dbe1, db1, dg2, dbe2, db2, dW3, db3};
  r.nseg = 8;
  for (int q = 0; q < 8; q++) { r.seg_beg[q] = begs[q]; r.seg_end[q] = ends[q]; r.seg_dst[q] = dsts[q]; }
  const long total = off.pv + wn;
  hipLaunchKernelGGL(enc_tail_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, stream, r);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}
