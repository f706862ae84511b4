// Encoder "tail": everything of Input_Module (model.py:60-143) between the wide first-layer product and the latent
// output, in one forward kernel and three backward kernels instead of ~30 launches, for up to four branches of one
// encoder per launch (hidden_rep + extra_linear_rep and hidden_lib + its head share the minibatch):
//
//   slabs (split-K partials of x W1^T)  -> + b1 -> LayerNorm -> LeakyReLU -> Dropout                  (utils.py:57-63)
//                                       -> Linear(H1, H2) -> LayerNorm -> LeakyReLU -> Dropout         (second FCNN layer)
//                                       -> Linear(H2, N3)   (extra_linear_rep, model.py:83-84; the library head
//                                                            Linear(H2, 1), model.py:92; N3 = 0: none)
// An optional covariate block (model.py:128: torch.cat([x, cov]) in front of the first layer) enters after the slab sum
// as cov W1[:, F:]^T; the gradient of those weight columns comes out of the backward's records, either as a strided
// block of the layer's gradient or consumed by the AdamW step of those columns in the reduction kernel.
//
// The work is row-local and tiny (2 B H1 H2 flop), so it runs in exact fp32 on the vector units.  A wave owns FOUR
// cells: every row of W2 it loads from L2 is used four times, and the four dot products of a W2 row are reduced
// across the wave together.  Backward: a row kernel (same layout: gradients of the small tensors as per-wave
// partial records, dY1 = gradient of the first-layer product -- fp32 and / or transposed bf16, the operand of the
// first-layer weight-gradient kernel -- and dz2), a column kernel for dW2 = dz2^T a1 (a1 recomputed from the saved
// xhat1), and a fixed-order reduction of the partials.
#include <type_traits>
#include "common.h"
#include "scpair_hip.h"

#define ET_NCH 4        // chunks of 256 columns per row (H1 <= 1024)
#define ET_RW 2         // cells per wave (1, 2 or 4)
#define ET_QN (64 / ET_RW)
#define ET_LG (ET_RW == 4 ? 2 : (ET_RW == 2 ? 1 : 0))
#define ET_WAVES 2      // waves per workgroup
typedef float f4 __attribute__((ext_vector_type(4)));

struct TailSet {
  ScpEncTail b[SCP_ENC_TAIL_MAX_BRANCHES];
};

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
__device__ __forceinline__ float lane_bcast(float v, int src) {       // src uniform across the wave
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// RW per-lane values, each summed over the RW lanes that differ in the low log2(RW) bits ("transposing": lane with
// (lane & (RW-1)) = i gets the sum of value i).  RW = 1, 2 or 4.
template <int RW> __device__ __forceinline__ float low_tsum(const float (&v)[RW], int lane) {
  if constexpr (RW == 1) {
    return v[0];
  } else if constexpr (RW == 2) {
    const bool o1 = lane & 1;
    const float a = o1 ? v[1] : v[0], sa = o1 ? v[0] : v[1];
    return a + dppf<0xB1>(sa);
  } else {
    const bool o1 = lane & 1, o2 = lane & 2;
    float a = o1 ? v[1] : v[0], sa = o1 ? v[0] : v[1];
    float b = o1 ? v[3] : v[2], sb = o1 ? v[2] : v[3];
    a += dppf<0xB1>(sa);
    b += dppf<0xB1>(sb);
    const float k = o2 ? b : a, sk = o2 ? a : b;
    return k + dppf<0x4E>(sk);
  }
}
// sum over the 64 / RW lanes with the same (lane & (RW-1)): quad steps and two rotations inside the row of 16, then
// across the four rows
template <int RW> __device__ __forceinline__ float cell_sum(float v) {
  if constexpr (RW == 1) v += dppf<0xB1>(v);
  if constexpr (RW <= 2) v += dppf<0x4E>(v);
  v += dppf<0x128>(v);
  v += dppf<0x124>(v);
  return xor32_sum(xor16_sum(v));
}
// RW per-lane values summed over the whole wave: every lane returns the total of value (lane & (RW-1)).
template <int RW> __device__ __forceinline__ float reduce_rw(const float (&v)[RW], int lane) {
  return cell_sum<RW>(low_tsum<RW>(v, lane));
}

// Layouts inside a wave (RW cells, RW = 1, 2 or 4; QN = 64 / RW):
//   wide rows (H1):   lane l holds columns 256 c + 4 l .. + 3 (c < NCH) of ALL the wave's cells;
//   narrow rows (H2, N3 <= 64): lane l holds cell (l & (RW-1)), columns QN i + (l / RW), i < RW.
// The loops over the rows of W2 carry no branch (a branch, even a uniform one, ends a basic block and the loads of a
// block are waited for at its end): the chunk count is a template parameter, lanes past H1 read a clamped address and
// mask the value, and the H2 loop is a runtime loop inside at most RW blocks of QN columns.
struct TailOff {
  int g1, be1, b1, g2, be2, b2, w3, b3, wc, pv;
};
// Layout of one wave's partial-gradient record (PV floats, PV % 4 == 0):
//   dgamma1[H1] dbeta1[H1] dbias1[H1] dgamma2[H2] dbeta2[H2] dbias2[H2] dW3[N3*H2] db3[N3] (pad to 4) dW1c[C][H1]
// (dW1c: gradient of the first-layer weight columns that multiply the covariate block, C = 0: none)
static __host__ __device__ inline TailOff tail_off(int H1, int H2, int N3, int C) {
  TailOff o;
  o.g1 = 0; o.be1 = H1; o.b1 = 2 * H1; o.g2 = 3 * H1; o.be2 = o.g2 + H2; o.b2 = o.be2 + H2; o.w3 = o.b2 + H2;
  o.b3 = o.w3 + N3 * H2;
  o.wc = (o.b3 + N3 + 3) / 4 * 4;
  o.pv = o.wc + C * H1;                             // H1 % 4 == 0
  return o;
}
#define ETW_ROWS 64
// workspace of one branch: records [nrec][pv] | dW2 slabs [nchunk][H2*H1] | dz2 [B][H2]
struct TailWs {
  int nrec, nchunk;
  long wn, o_slab, o_dz2, total;
};
static __host__ __device__ inline TailWs tail_ws(int B, int H1, int H2, int N3, int C) {
  TailWs w;
  w.nrec = (B + ET_RW - 1) / ET_RW;
  w.nchunk = (B + ETW_ROWS - 1) / ETW_ROWS;
  w.wn = (long)H2 * H1;
  w.o_slab = (long)w.nrec * tail_off(H1, H2, N3, C).pv;
  w.o_dz2 = w.o_slab + w.nchunk * w.wn;
  w.total = w.o_dz2 + (long)B * H2 + 8;
  return w;
}

// Layout of a branch's BatchNorm workspace (ScpEncTail.bn_ws): per layer scale = w * rstd, shift = b - mean * scale, mean,
// rstd (written by the forward's column-statistics kernel, read again by the backward) and the backward's column sums
// S1 = sum_cells dy, S2 = sum_cells dy * xhat_bn.  The H2-wide vectors are padded to multiples of 4.
struct BnOff {
  int sc1, sh1, mu1, rs1, s1a, s2a, sc2, sh2, mu2, rs2, s1b, s2b, total;
};
static __host__ __device__ inline BnOff bn_off(int H1, int H2) {
  BnOff o;
  const int h2p = (H2 + 3) / 4 * 4;
  o.sc1 = 0; o.sh1 = H1; o.mu1 = 2 * H1; o.rs1 = 3 * H1; o.s1a = 4 * H1; o.s2a = 5 * H1;
  o.sc2 = 6 * H1; o.sh2 = o.sc2 + h2p; o.mu2 = o.sh2 + h2p; o.rs2 = o.mu2 + h2p; o.s1b = o.rs2 + h2p; o.s2b = o.s1b + h2p;
  o.total = o.s2b + h2p;
  return o;
}

// W3 [N3][H2] and b3 into LDS as [n][65] + [64]; every thread of the workgroup takes part
__device__ __forceinline__ void stage_w3(const ScpEncTail& m, float* W3s, int tid, int nthreads) {
  const int nw = m.N3 * m.H2;
  for (int idx = tid; idx < nw; idx += nthreads) {
    const int n = idx / m.H2, j = idx - n * m.H2;
    W3s[n * 65 + j] = m.W3[idx];
  }
  for (int idx = tid; idx < m.N3; idx += nthreads) W3s[64 * 65 + idx] = m.b3[idx];
}

// ---------------------------------------------------------------------------------------------------------------
// BN (batchnorm=True encoders): `seg` selects a segment of the tail -- 0: all of it, BatchNorm applied with the scale /
// shift vectors in bn_ws (running statistics); 1: slabs -> xhat1, rstd1; 2: xhat1 -> BatchNorm 1, dropout, second layer ->
// xhat2, rstd2; 3: xhat2 -> BatchNorm 2, dropout, final Linear -> out.  The column-statistics kernel runs between them.
template <int NCH, bool BN>
__global__ __launch_bounds__(64 * ET_WAVES) void enc_tail_fwd_kernel(TailSet S, int seg) {
  __shared__ float W3s[64 * 65 + 64];
  const ScpEncTail& m = S.b[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int H1 = m.H1, H2 = m.H2, N3 = m.N3, B = m.B;
  if (N3 > 0) stage_w3(m, W3s, tid, 64 * ET_WAVES);
  __syncthreads();
  const int row0 = (blockIdx.x * ET_WAVES + wv) * ET_RW;
  if (row0 >= B) return;                          // no barrier below
  const uint64_t sd1 = m.seed1 + (m.seed_dev ? m.seed_dev[0] : 0), sd2 = m.seed2 + (m.seed_dev ? m.seed_dev[0] : 0);
  const f4 zero4 = f4{0.f, 0.f, 0.f, 0.f};
  bool ok[NCH];
  int ko[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int k = c * 256 + lane * 4;
    ok[c] = k < H1;
    ko[c] = min(k, H1 - 4);
  }
  int rowc[ET_RW];
  bool live[ET_RW];
#pragma unroll
  for (int r = 0; r < ET_RW; r++) {
    live[r] = row0 + r < B;
    rowc[r] = min(row0 + r, B - 1);               // cells past the batch end recompute the last one and store nothing
  }
  f4 a[ET_RW][NCH];
  const int zc = lane & (ET_RW - 1), q = lane >> ET_LG;      // narrow layout: my cell, my column residue
  const int rowz = min(row0 + zc, B - 1);
  const bool livez = row0 + zc < B;
  float zr[ET_RW];
  bool vz[ET_RW];
  int jc[ET_RW];
  float xh2[ET_RW];
#pragma unroll
  for (int i = 0; i < ET_RW; i++) {
    vz[i] = ET_QN * i + q < H2;
    jc[i] = min(ET_QN * i + q, H2 - 1);
  }
  [[maybe_unused]] const BnOff bo = bn_off(H1, H2);
  if (!BN || seg != 3) {
  if (!BN || seg < 2) {
  f4 v[ET_RW][NCH];
  f4 gam[NCH], bet[NCH];                          // issued ahead of the slab loads: nothing later waits on them alone
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const f4 b = m.b1 ? *(const f4*)(m.b1 + ko[c]) : zero4;
#pragma unroll
    for (int r = 0; r < ET_RW; r++) v[r][c] = b;
    gam[c] = *(const f4*)(m.g1 + ko[c]);
    bet[c] = *(const f4*)(m.be1 + ko[c]);
  }
  // split-K slabs of the first-layer product, summed in ascending order; four slabs of loads in flight
  {
    const float* base[ET_RW];
#pragma unroll
    for (int r = 0; r < ET_RW; r++) base[r] = m.slabs + (long)rowc[r] * m.ld_slab;
    int sl = 0;
    for (; sl + 4 <= m.nslab; sl += 4) {
      f4 t[4][ET_RW][NCH];
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int r = 0; r < ET_RW; r++)
#pragma unroll
          for (int c = 0; c < NCH; c++) t[u][r][c] = *(const f4*)(base[r] + (long)(sl + u) * m.slab_stride + ko[c]);
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int r = 0; r < ET_RW; r++)
#pragma unroll
          for (int c = 0; c < NCH; c++) v[r][c] += t[u][r][c];
    }
    for (; sl < m.nslab; sl++) {
      f4 t[ET_RW][NCH];
#pragma unroll
      for (int r = 0; r < ET_RW; r++)
#pragma unroll
        for (int c = 0; c < NCH; c++) t[r][c] = *(const f4*)(base[r] + (long)sl * m.slab_stride + ko[c]);
#pragma unroll
      for (int r = 0; r < ET_RW; r++)
#pragma unroll
        for (int c = 0; c < NCH; c++) v[r][c] += t[r][c];
    }
  }
  // covariate block of the input (model.py:128 concatenates it to x): + cov[cell][j] * W1[k][F + j], exact fp32
  for (int j = 0; j < m.C; j++) {
    f4 wj[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      const float* w = m.W1c + (long)ko[c] * m.ldw1c + j;
      wj[c] = f4{w[0], w[m.ldw1c], w[2 * m.ldw1c], w[3 * m.ldw1c]};
    }
#pragma unroll
    for (int r = 0; r < ET_RW; r++) {
      const float cv = m.cov[(long)rowc[r] * m.ldcov + j];
#pragma unroll
      for (int c = 0; c < NCH; c++) v[r][c] += cv * wj[c];
    }
  }
  // LayerNorm 1 (two-pass statistics, biased variance: nn.LayerNorm), activation, dropout
  float part[ET_RW];
#pragma unroll
  for (int r = 0; r < ET_RW; r++) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      v[r][c] = ok[c] ? v[r][c] : zero4;
      s += f4sum(v[r][c]);
    }
    part[r] = s;
  }
  const float ssum = reduce_rw<ET_RW>(part, lane);
  float mean1[ET_RW];
#pragma unroll
  for (int r = 0; r < ET_RW; r++) {
    mean1[r] = lane_bcast(ssum, r) / (float)H1;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      const f4 d = ok[c] ? v[r][c] - mean1[r] : zero4;
      q += f4sum(d * d);
    }
    part[r] = q;
  }
  const float qsum = reduce_rw<ET_RW>(part, lane);
#pragma unroll
  for (int r = 0; r < ET_RW; r++) {
    const float rstd1 = 1.f / sqrtf(lane_bcast(qsum, r) / (float)H1 + m.eps1);
    if (lane == 0 && m.rstd1 && live[r]) m.rstd1[rowc[r]] = rstd1;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      const f4 xh = (v[r][c] - mean1[r]) * rstd1;
      if (m.xhat1 && ok[c] && live[r]) *(f4*)(m.xhat1 + (long)rowc[r] * H1 + ko[c]) = xh;
      if constexpr (BN) {
        a[r][c] = xh;                              // activation, BatchNorm and dropout follow below
      } else {
        const f4 z = xh * gam[c] + bet[c];
        const f4 t = leaky4(z, m.slope) * keep4(sd1, (uint64_t)rowc[r] * H1 + ko[c], m.p1);
        a[r][c] = ok[c] ? t : zero4;
      }
    }
  }
  if (BN && seg == 1) return;                     // the column statistics of the activations come next
  }  // seg < 2
  if constexpr (BN) {
    f4 gam[NCH], bet[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      gam[c] = *(const f4*)(m.g1 + ko[c]);
      bet[c] = *(const f4*)(m.be1 + ko[c]);
    }
    if (seg == 2) {
#pragma unroll
      for (int r = 0; r < ET_RW; r++)
#pragma unroll
        for (int c = 0; c < NCH; c++) a[r][c] = *(const f4*)(m.xhat1 + (long)rowc[r] * H1 + ko[c]);
    }
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      const f4 sc = *(const f4*)(m.bn_ws + bo.sc1 + ko[c]), sh = *(const f4*)(m.bn_ws + bo.sh1 + ko[c]);
#pragma unroll
      for (int r = 0; r < ET_RW; r++) {
        const f4 z = a[r][c] * gam[c] + bet[c];
        const f4 t = (leaky4(z, m.slope) * sc + sh) * keep4(sd1, (uint64_t)rowc[r] * H1 + ko[c], m.p1);
        a[r][c] = ok[c] ? t : zero4;
      }
    }
  }
  // second layer: z2[cell][j] = a1[cell] . W2[j] + b2[j]; the wave streams the rows of W2 (independent loads, four
  // rows in flight) and forms the four cells' dot products with each
#pragma unroll
  for (int i = 0; i < ET_RW; i++) zr[i] = 0.f;
#pragma unroll
  for (int jb = 0; jb < ET_RW; jb++) {
    if (jb * ET_QN < H2) {                        // uniform
      const int jn = min(ET_QN, H2 - jb * ET_QN);
      float zk = 0.f;
      // Four rows of W2 per group, software-pipelined: the loads of the next group are in flight while this one is
      // multiplied (the compiler does not unroll a loop of unknown trip count on request, and a group per trip with
      // nothing in flight exposes the L2 latency ten times).  The row count is padded to a multiple of four with
      // clamped (valid) row addresses: a padding row lands in no lane (jj + u >= QN) or in a lane that is masked below.
      auto load4 = [&](f4 (&w)[4][NCH], int jj) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const float* row = m.W2 + (long)min(jb * ET_QN + jj + u, H2 - 1) * H1;
#pragma unroll
          for (int c = 0; c < NCH; c++) w[u][c] = *(const f4*)(row + ko[c]);
        }
      };
      auto mul4 = [&](const f4 (&w)[4][NCH], int jj) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
          float p[ET_RW];
#pragma unroll
          for (int r = 0; r < ET_RW; r++) {
            f4 acc = a[r][0] * w[u][0];           // a = 0 in masked lanes
#pragma unroll
            for (int c = 1; c < NCH; c++) acc += a[r][c] * w[u][c];
            p[r] = f4sum(acc);
          }
          const float k = reduce_rw<ET_RW>(p, lane);
          zk = (jj + u == q) ? k : zk;
        }
      };
      {
        const int ng = (jn + 3) >> 2;             // groups of four rows
        f4 wa[4][NCH], wb[4][NCH];
        load4(wa, 0);
        for (int gq = 0; gq < ng; gq += 2) {
          load4(wb, 4 * min(gq + 1, ng - 1));
          mul4(wa, 4 * gq);
          load4(wa, 4 * min(gq + 2, ng - 1));
          if (gq + 1 < ng) mul4(wb, 4 * (gq + 1));
        }
      }
      zr[jb] = zk;
    }
  }
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < ET_RW; i++) {
    zr[i] = vz[i] ? zr[i] + m.b2[jc[i]] : 0.f;
    s2 += zr[i];
  }
  const float mean2 = cell_sum<ET_RW>(s2) / (float)H2;
  float q2 = 0.f;
#pragma unroll
  for (int i = 0; i < ET_RW; i++) {
    xh2[i] = vz[i] ? zr[i] - mean2 : 0.f;
    q2 += xh2[i] * xh2[i];
  }
  const float rstd2 = 1.f / sqrtf(cell_sum<ET_RW>(q2) / (float)H2 + m.eps2);
  if (q == 0 && m.rstd2 && livez) m.rstd2[rowz] = rstd2;
#pragma unroll
  for (int i = 0; i < ET_RW; i++) {
    xh2[i] *= rstd2;
    if (m.xhat2 && vz[i] && livez) m.xhat2[(long)rowz * H2 + jc[i]] = xh2[i];
  }
  if (BN && seg == 2) return;                     // column statistics of the second layer's activations come next
  } else {   // BN, seg == 3
#pragma unroll
    for (int i = 0; i < ET_RW; i++) xh2[i] = vz[i] ? m.xhat2[(long)rowz * H2 + jc[i]] : 0.f;
  }
  float x2[ET_RW];
#pragma unroll
  for (int i = 0; i < ET_RW; i++) {
    float t = leaky_(xh2[i] * m.g2[jc[i]] + m.be2[jc[i]], m.slope);
    if constexpr (BN) t = t * m.bn_ws[bo.sc2 + jc[i]] + m.bn_ws[bo.sh2 + jc[i]];
    t *= keep_scale(sd2, (uint64_t)rowz * H2 + jc[i], m.p2);
    x2[i] = vz[i] ? t : 0.f;
  }
  if (N3 == 0) {
#pragma unroll
    for (int i = 0; i < ET_RW; i++)
      if (vz[i] && livez) m.out[(long)rowz * m.ldo + jc[i]] = x2[i];
    return;
  }
  // final Linear(H2, N3): a lane forms the partial dot product over its columns, the cell's 16 lanes add up
  float o[ET_RW];
#pragma unroll
  for (int i = 0; i < ET_RW; i++) o[i] = 0.f;
#pragma unroll
  for (int nb = 0; nb < ET_RW; nb++) {
    if (nb * ET_QN < N3) {
      const int nn_ = min(ET_QN, N3 - nb * ET_QN);
      float ok_ = 0.f;
      auto rows_of_w3 = [&](auto u_c, int nn) {
        constexpr int U = decltype(u_c)::value;
        float pr[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const float* w = W3s + (nb * ET_QN + nn + u) * 65;
          pr[u] = x2[0] * w[jc[0]];
#pragma unroll
          for (int i = 1; i < ET_RW; i++) pr[u] += x2[i] * w[jc[i]];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          const float tot = cell_sum<ET_RW>(pr[u]);
          ok_ = (nn + u == q) ? tot : ok_;
        }
      };
      int nn = 0;
      for (; nn + 4 <= nn_; nn += 4) rows_of_w3(std::integral_constant<int, 4>{}, nn);
      for (; nn < nn_; nn++) rows_of_w3(std::integral_constant<int, 1>{}, nn);
      o[nb] = ok_;
    }
  }
#pragma unroll
  for (int i = 0; i < ET_RW; i++) {
    const int n = ET_QN * i + q;
    if (n < N3 && livez) m.out[(long)rowz * m.ldo + n] = o[i] + W3s[64 * 65 + n];
  }
}

// BN: `seg` selects a segment of the backward -- 1: dOut -> final Linear, dropout 2 -> dy2 (left in the dz2 buffer);
// 2: dy2 -> BatchNorm 2, activation, LayerNorm 2 -> dz2, second Linear, dropout 1 -> dy1 (bn_dy1); 3: dy1 -> BatchNorm 1,
// activation, LayerNorm 1 -> dY1 / dYT.  The column-sum kernel (S1, S2 of the BatchNorm backward) runs between them.
template <int NCH, bool BN>
__global__ __launch_bounds__(64 * ET_WAVES) void enc_tail_bwd_kernel(TailSet S, int seg) {
  __shared__ float W3s[64 * 65 + 64];
  const ScpEncTail& m = S.b[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int H1 = m.H1, H2 = m.H2, N3 = m.N3, B = m.B;
  if (N3 > 0) stage_w3(m, W3s, tid, 64 * ET_WAVES);
  __syncthreads();
  const int rec_i = blockIdx.x * ET_WAVES + wv;
  const int row0 = rec_i * ET_RW;
  if (row0 >= B) return;                          // no barrier below
  const int NO = N3 > 0 ? N3 : H2;                 // width of dOut
  const TailOff off = tail_off(H1, H2, N3, m.C);
  const TailWs wsl = tail_ws(B, H1, H2, N3, m.C);
  [[maybe_unused]] const BnOff bo = bn_off(H1, H2);
  float* rec = m.ws + (long)rec_i * off.pv;
  float* dz2g = m.ws + wsl.o_dz2;
  const uint64_t sd1 = m.seed1 + (m.seed_dev ? m.seed_dev[0] : 0), sd2 = m.seed2 + (m.seed_dev ? m.seed_dev[0] : 0);
  const f4 zero4 = f4{0.f, 0.f, 0.f, 0.f};
  bool ok[NCH];
  int ko[NCH];
  f4 gam[NCH], bet[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    ok[c] = c * 256 + lane * 4 < H1;
    ko[c] = min(c * 256 + lane * 4, H1 - 4);
    gam[c] = *(const f4*)(m.g1 + ko[c]);
    bet[c] = *(const f4*)(m.be1 + ko[c]);
  }
  // ---- narrow part: cell (lane & (RW-1)), columns QN i + lane / RW --------------------------------------------
  const int zc = lane & (ET_RW - 1), q = lane >> ET_LG;
  const int rowz = min(row0 + zc, B - 1);
  const bool livez = row0 + zc < B;                // a dead cell has dOut = 0, hence every gradient of it is 0
  bool vz[ET_RW];
  int jc[ET_RW];
  float dor[ET_RW], xh2[ET_RW], g2v[ET_RW], zn[ET_RW], kp[ET_RW], a2[ET_RW];
  const float r2 = m.rstd2[rowz];
#pragma unroll
  for (int i = 0; i < ET_RW; i++) {
    const int j = ET_QN * i + q;
    vz[i] = j < H2;
    jc[i] = min(j, H2 - 1);
    dor[i] = (j < NO && livez) ? m.dOut[(long)rowz * m.lddo + min(j, NO - 1)] : 0.f;
    xh2[i] = vz[i] ? m.xhat2[(long)rowz * H2 + jc[i]] : 0.f;
    g2v[i] = vz[i] ? m.g2[jc[i]] : 0.f;
    zn[i] = xh2[i] * g2v[i] + m.be2[jc[i]];
    kp[i] = vz[i] ? keep_scale(sd2, (uint64_t)rowz * H2 + jc[i], m.p2) : 0.f;
    if constexpr (BN) a2[i] = (leaky_(zn[i], m.slope) * m.bn_ws[bo.sc2 + jc[i]] + m.bn_ws[bo.sh2 + jc[i]]) * kp[i];
    else a2[i] = leaky_(zn[i], m.slope) * kp[i];
  }
  float da2[ET_RW];
#pragma unroll
  for (int i = 0; i < ET_RW; i++) da2[i] = 0.f;
  f4 da[ET_RW][NCH];
  f4 xh[ET_RW][NCH];
  float d2[ET_RW];
  if (!BN || seg != 3) {
  if (BN && seg == 2) {
    // dy2 of the first segment (gradient at the BatchNorm output) -> gradient at the activation output
    const float invB = 1.f / (float)B;
#pragma unroll
    for (int i = 0; i < ET_RW; i++) {
      const float dy = (vz[i] && livez) ? dz2g[(long)rowz * H2 + jc[i]] : 0.f;
      const float xb = (leaky_(zn[i], m.slope) - m.bn_ws[bo.mu2 + jc[i]]) * m.bn_ws[bo.rs2 + jc[i]];
      const float mt = m.bn_train ? (m.bn_ws[bo.s1b + jc[i]] + xb * m.bn_ws[bo.s2b + jc[i]]) * invB : 0.f;
      da2[i] = vz[i] ? m.bn_ws[bo.sc2 + jc[i]] * (dy - (livez ? mt : 0.f)) : 0.f;
      kp[i] = vz[i] ? 1.f : 0.f;                   // the dropout mask is already in dy2
    }
  } else
  if (N3 > 0) {
    // da2[cell][j] = sum_n dOut[cell][n] W3[n][j];  dW3[n][j] += sum over the wave's cells of dOut[cell][n] a2[cell][j]
#pragma unroll
    for (int nb = 0; nb < ET_RW; nb++) {
      if (nb * ET_QN < N3) {
        const int nn_ = min(ET_QN, N3 - nb * ET_QN);
        for (int nn = 0; nn < nn_; nn++) {
          const int n = nb * ET_QN + nn;
          float dn = lane_bcast(dor[nb], ET_RW * nn);                  // dOut[my cell][n]: lane (cell, q = nn) holds it
#pragma unroll
          for (int cc = 1; cc < ET_RW; cc++) {
            const float dcc = lane_bcast(dor[nb], ET_RW * nn + cc);
            dn = (zc == cc) ? dcc : dn;
          }
          const float* w = W3s + n * 65;
#pragma unroll
          for (int i = 0; i < ET_RW; i++) da2[i] += dn * w[jc[i]];
          float pw[ET_RW];
#pragma unroll
          for (int i = 0; i < ET_RW; i++) pw[i] = dn * a2[i];
          const float t = low_tsum<ET_RW>(pw, lane);
          const int j = ET_QN * zc + q;                           // after the sum lane (i = zc, q) holds column QN i + q
          if (j < H2) rec[off.w3 + n * H2 + j] = t;
        }
      }
    }
    const float tb = low_tsum<ET_RW>(dor, lane);
    if (ET_QN * zc + q < N3) rec[off.b3 + ET_QN * zc + q] = tb;
  } else {
#pragma unroll
    for (int i = 0; i < ET_RW; i++) da2[i] = dor[i];
  }
  if (BN && seg == 1) {
#pragma unroll
    for (int i = 0; i < ET_RW; i++)
      if (vz[i] && livez) dz2g[(long)rowz * H2 + jc[i]] = da2[i] * kp[i];
    return;                                       // the column sums of dy2 come next
  }
  // LayerNorm 2 / activation / dropout backward
  float dz[ET_RW], dx[ET_RW];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < ET_RW; i++) {
    const float gq = da2[i] * kp[i];
    dz[i] = vz[i] ? (zn[i] > 0.f ? gq : gq * m.slope) : 0.f;
    dx[i] = dz[i] * g2v[i];
    s1 += dx[i];
    s2 += dx[i] * xh2[i];
  }
  s1 = cell_sum<ET_RW>(s1) / (float)H2;
  s2 = cell_sum<ET_RW>(s2) / (float)H2;
#pragma unroll
  for (int i = 0; i < ET_RW; i++) {
    d2[i] = vz[i] ? r2 * (dx[i] - s1 - xh2[i] * s2) : 0.f;
    if (vz[i] && livez) dz2g[(long)rowz * H2 + jc[i]] = d2[i];
  }
  {
    float pg[ET_RW];
#pragma unroll
    for (int i = 0; i < ET_RW; i++) pg[i] = dz[i] * xh2[i];
    const float tg = low_tsum<ET_RW>(pg, lane);
    const float tb = low_tsum<ET_RW>(dz, lane);
    const float td = low_tsum<ET_RW>(d2, lane);
    const int j = ET_QN * zc + q;
    if (j < H2) {
      rec[off.g2 + j] = tg;
      rec[off.be2 + j] = tb;
      rec[off.b2 + j] = td;
    }
  }
  // ---- wide part: gradient wrt a1: da[cell][k] = sum_j dz2[cell][j] W2[j][k] ------------------------------------
  if (!BN) {                                      // saved x-hat of layer 1: loaded here, used after the W2 loop
#pragma unroll
    for (int r = 0; r < ET_RW; r++)
#pragma unroll
      for (int c = 0; c < NCH; c++) xh[r][c] = *(const f4*)(m.xhat1 + (long)min(row0 + r, B - 1) * H1 + ko[c]);
  }
#pragma unroll
  for (int r = 0; r < ET_RW; r++)
#pragma unroll
    for (int c = 0; c < NCH; c++) da[r][c] = zero4;
#pragma unroll
  for (int jb = 0; jb < ET_RW; jb++) {
    if (jb * ET_QN < H2) {
      const int jn = min(ET_QN, H2 - jb * ET_QN);
      // same pipeline as the forward kernel: groups of four rows of W2, the next group's loads in flight; padding rows
      // carry dz2 = 0 (masked above)
      auto load4 = [&](f4 (&w)[4][NCH], int jj) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const float* row = m.W2 + (long)min(jb * ET_QN + jj + u, H2 - 1) * H1;
#pragma unroll
          for (int c = 0; c < NCH; c++) w[u][c] = *(const f4*)(row + ko[c]);
        }
      };
      auto mul4 = [&](const f4 (&w)[4][NCH], int jj) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
          for (int r = 0; r < ET_RW; r++) {
            const float dzj = lane_bcast(d2[jb], r + ET_RW * (jj + u));   // dz2[cell r][QN jb + jj + u]
#pragma unroll
            for (int c = 0; c < NCH; c++) da[r][c] += dzj * w[u][c];
          }
      };
      const int ng = (jn + 3) >> 2;
      f4 wa[4][NCH], wb[4][NCH];
      load4(wa, 0);
      for (int gq = 0; gq < ng; gq += 2) {
        load4(wb, 4 * min(gq + 1, ng - 1));
        mul4(wa, 4 * gq);
        load4(wa, 4 * min(gq + 2, ng - 1));
        if (gq + 1 < ng) mul4(wb, 4 * (gq + 1));
      }
    }
  }
  if (BN && seg == 2) {
    // gradient at the output of BatchNorm 1 (dropout mask applied): handed to the column sums and the last segment
#pragma unroll
    for (int r = 0; r < ET_RW; r++) {
      const int row = min(row0 + r, B - 1);
#pragma unroll
      for (int c = 0; c < NCH; c++)
        if (ok[c] && row0 + r < B)
          *(f4*)(m.bn_dy1 + (long)row * H1 + ko[c]) = da[r][c] * keep4(sd1, (uint64_t)row * H1 + ko[c], m.p1);
    }
    return;
  }
  }  // seg != 3
  if (BN && seg == 3) {
#pragma unroll
    for (int r = 0; r < ET_RW; r++)
#pragma unroll
      for (int c = 0; c < NCH; c++) xh[r][c] = *(const f4*)(m.xhat1 + (long)min(row0 + r, B - 1) * H1 + ko[c]);
  }
  // LayerNorm 1 / activation / dropout backward; column sums over the wave's cells stay in registers
  f4 sg[NCH], sb[NCH], sd[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++) sg[c] = sb[c] = sd[c] = zero4;
  float t1p[ET_RW], t2p[ET_RW];
#pragma unroll
  for (int r = 0; r < ET_RW; r++) {
    const int row = min(row0 + r, B - 1);
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      const f4 xl = xh[r][c];
      const f4 z = xl * gam[c] + bet[c];
      f4 gup;                                      // gradient at the activation output
      if constexpr (BN) {
        const f4 dy = (row0 + r < B) ? *(const f4*)(m.bn_dy1 + (long)row * H1 + ko[c]) : zero4;
        const f4 sc = *(const f4*)(m.bn_ws + bo.sc1 + ko[c]);
        if (m.bn_train) {
          const f4 xb = (leaky4(z, m.slope) - *(const f4*)(m.bn_ws + bo.mu1 + ko[c])) * *(const f4*)(m.bn_ws + bo.rs1 + ko[c]);
          const f4 mt = (*(const f4*)(m.bn_ws + bo.s1a + ko[c]) + xb * *(const f4*)(m.bn_ws + bo.s2a + ko[c])) * (1.f / (float)B);
          gup = (row0 + r < B) ? sc * (dy - mt) : zero4;
        } else {
          gup = sc * dy;
        }
      } else {
        gup = da[r][c] * keep4(sd1, (uint64_t)row * H1 + ko[c], m.p1);
      }
      const f4 dzl = dleaky4(z, gup, m.slope);
      const f4 dz1 = ok[c] ? dzl : zero4;          // dead cells: dz2 = 0, hence da = 0 and dz1 = 0
      xh[r][c] = ok[c] ? xl : zero4;
      sg[c] += dz1 * xh[r][c];
      sb[c] += dz1;
      da[r][c] = dz1 * gam[c];                      // from here on: d xhat
      t1 += f4sum(da[r][c]);
      t2 += f4sum(da[r][c] * xh[r][c]);
    }
    t1p[r] = t1;
    t2p[r] = t2;
  }
  const float t1s = reduce_rw<ET_RW>(t1p, lane);
  const float t2s = reduce_rw<ET_RW>(t2p, lane);
#pragma unroll
  for (int r = 0; r < ET_RW; r++) {
    const int row = min(row0 + r, B - 1);
    const bool live = row0 + r < B;
    const float r1 = m.rstd1[row];
    const float t1 = lane_bcast(t1s, r) / (float)H1, t2 = lane_bcast(t2s, r) / (float)H1;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      const f4 cd = ok[c] ? (da[r][c] - t1 - xh[r][c] * t2) * r1 : zero4;
      da[r][c] = cd;                                // from here on: dY1
      sd[c] += cd;
      if (m.dY1 && ok[c] && live) *(f4*)(m.dY1 + (long)row * m.ldy + ko[c]) = cd;
    }
  }
  // gradient of the covariate columns of W1: dW1c[k][j] = sum over cells of dY1[cell][k] cov[cell][j]; the wave's share
  // goes into its record as [j][k] (dead cells carry dY1 = 0)
  for (int j = 0; j < m.C; j++) {
    float cv[ET_RW];
#pragma unroll
    for (int r = 0; r < ET_RW; r++) cv[r] = m.cov[(long)min(row0 + r, B - 1) * m.ldcov + j];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      f4 t = cv[0] * da[0][c];
#pragma unroll
      for (int r = 1; r < ET_RW; r++) t += cv[r] * da[r][c];
      if (ok[c]) *(f4*)(rec + off.wc + (long)j * H1 + ko[c]) = t;
    }
  }
  if (m.dYT) {
    // transposed bf16 copy [H1][ldt]: the wave's cells are consecutive columns -> one store per row
    // of dY^T (cells past the batch end hold 0; ldt covers the batch rounded up to 64)
    __bf16* T = (__bf16*)m.dYT;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
      if (ok[c]) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          typedef __bf16 bfrw __attribute__((ext_vector_type(ET_RW == 1 ? 2 : ET_RW)));
          if constexpr (ET_RW == 1) {
            T[(long)(ko[c] + e) * m.ldt + row0] = (__bf16)da[0][c][e];
          } else {
            bfrw o;
#pragma unroll
            for (int r = 0; r < ET_RW; r++) o[r] = (__bf16)da[r][c][e];
            *(bfrw*)(T + (long)(ko[c] + e) * m.ldt + row0) = o;
          }
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    if (ok[c]) {
      *(f4*)(rec + off.g1 + ko[c]) = sg[c];
      *(f4*)(rec + off.be1 + ko[c]) = sb[c];
      *(f4*)(rec + off.b1 + ko[c]) = sd[c];
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// BatchNorm column kernels.  layer 1: activations a = leaky(xhat1 g1 + be1), [B][H1]; layer 2: the same of the second
// layer, [B][H2].  A workgroup owns 16 columns; thread (col = tid & 15, rg = tid >> 4) walks rows rg, rg + 16, ...
// and the 16 partial sums of a column are added in a fixed order.
// ---------------------------------------------------------------------------------------------------------------
struct BnLayer {
  const float *xhat, *g, *be, *w, *b;
  float *rm, *rv;
  float eps, mom;
  int H, sc, sh, mu, rs, s1, s2;
};
__device__ __forceinline__ BnLayer bn_layer(const ScpEncTail& m, int layer) {
  const BnOff o = bn_off(m.H1, m.H2);
  BnLayer L;
  if (layer == 1) {
    L.xhat = m.xhat1; L.g = m.g1; L.be = m.be1; L.w = m.bn1_w; L.b = m.bn1_b; L.rm = m.bn1_rm; L.rv = m.bn1_rv;
    L.eps = m.bn_eps1; L.mom = m.bn_mom1; L.H = m.H1; L.sc = o.sc1; L.sh = o.sh1; L.mu = o.mu1; L.rs = o.rs1; L.s1 = o.s1a; L.s2 = o.s2a;
  } else {
    L.xhat = m.xhat2; L.g = m.g2; L.be = m.be2; L.w = m.bn2_w; L.b = m.bn2_b; L.rm = m.bn2_rm; L.rv = m.bn2_rv;
    L.eps = m.bn_eps2; L.mom = m.bn_mom2; L.H = m.H2; L.sc = o.sc2; L.sh = o.sh2; L.mu = o.mu2; L.rs = o.rs2; L.s1 = o.s1b; L.s2 = o.s2b;
  }
  return L;
}

// forward: batch mean / biased variance of the activations (two passes, as nn.BatchNorm1d in training mode), running
// statistics updated in place (unbiased variance), scale = w * rstd and shift = b - mean * scale for the next segment;
// train = 0: the same vectors from the running statistics.
__global__ __launch_bounds__(256) void enc_tail_bnstat_kernel(TailSet S, int layer) {
  __shared__ float red[16][17];
  const ScpEncTail& m = S.b[blockIdx.y];
  const BnLayer L = bn_layer(m, layer);
  const int col = blockIdx.x * 16 + (threadIdx.x & 15), rg = threadIdx.x >> 4, cl = threadIdx.x & 15;
  if (blockIdx.x * 16 >= L.H) return;              // uniform: a narrower branch of the same launch
  const int cc = min(col, L.H - 1);
  const float w = L.w ? L.w[cc] : 1.f, b = L.b ? L.b[cc] : 0.f;
  float mean, var;
  if (m.bn_train) {
    const float g = L.g[cc], be = L.be[cc];
    float s = 0.f;
    for (int row = rg; row < m.B; row += 16) s += leaky_(L.xhat[(long)row * L.H + cc] * g + be, m.slope);
    red[rg][cl] = s;
    __syncthreads();
    s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += red[i][cl];
    mean = s / (float)m.B;
    __syncthreads();
    float q = 0.f;
    for (int row = rg; row < m.B; row += 16) {
      const float d = leaky_(L.xhat[(long)row * L.H + cc] * g + be, m.slope) - mean;
      q += d * d;
    }
    red[rg][cl] = q;
    __syncthreads();
    q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) q += red[i][cl];
    var = q / (float)m.B;
    if (rg == 0 && col < L.H && L.rm && L.rv) {
      const float unb = m.B > 1 ? var * ((float)m.B / (float)(m.B - 1)) : var;
      L.rm[col] = (1.f - L.mom) * L.rm[col] + L.mom * mean;
      L.rv[col] = (1.f - L.mom) * L.rv[col] + L.mom * unb;
    }
  } else {
    mean = L.rm ? L.rm[cc] : 0.f;
    var = L.rv ? L.rv[cc] : 1.f;
  }
  if (rg == 0 && col < L.H) {
    const float rs = 1.f / sqrtf(var + L.eps);
    const float sc = w * rs;
    m.bn_ws[L.sc + col] = sc;
    m.bn_ws[L.sh + col] = b - mean * sc;
    m.bn_ws[L.mu + col] = mean;
    m.bn_ws[L.rs + col] = rs;
  }
}

// backward: S1 = sum_cells dy, S2 = sum_cells dy * xhat_bn (xhat_bn = (a - mean) * rstd) of the gradient dy [B][ld] that
// reaches the BatchNorm output; they are the gradients of the affine bias / weight and the two mean terms of the
// training-mode input gradient.
__global__ __launch_bounds__(256) void enc_tail_bnsum_kernel(TailSet S, int layer) {
  __shared__ float red[2][16][17];
  const ScpEncTail& m = S.b[blockIdx.y];
  const BnLayer L = bn_layer(m, layer);
  const int col = blockIdx.x * 16 + (threadIdx.x & 15), rg = threadIdx.x >> 4, cl = threadIdx.x & 15;
  if (blockIdx.x * 16 >= L.H) return;              // uniform
  const int cc = min(col, L.H - 1);
  const TailWs wsl = tail_ws(m.B, m.H1, m.H2, m.N3, m.C);
  const float* dy = layer == 1 ? m.bn_dy1 : m.ws + wsl.o_dz2;      // layer 2: the first backward segment left dy2 in the dz2 buffer
  const float g = L.g[cc], be = L.be[cc], mu = m.bn_ws[L.mu + cc], rs = m.bn_ws[L.rs + cc];
  float s1 = 0.f, s2 = 0.f;
  for (int row = rg; row < m.B; row += 16) {
    const float d = dy[(long)row * L.H + cc];
    const float xb = (leaky_(L.xhat[(long)row * L.H + cc] * g + be, m.slope) - mu) * rs;
    s1 += d;
    s2 += d * xb;
  }
  red[0][rg][cl] = s1;
  red[1][rg][cl] = s2;
  __syncthreads();
  if (rg == 0 && col < L.H) {
    s1 = s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      s1 += red[0][i][cl];
      s2 += red[1][i][cl];
    }
    m.bn_ws[L.s1 + col] = s1;
    m.bn_ws[L.s2 + col] = s2;
    float* dw = layer == 1 ? m.dbn1_w : m.dbn2_w;
    float* db = layer == 1 ? m.dbn1_b : m.dbn2_b;
    if (dw) dw[col] = s2;
    if (db) db[col] = s1;
  }
}

// dW2 slab of one (128-column block, 64-row chunk): slab[chunk][j][k] = sum over the chunk's rows of dz2[row][j] a1[row][k]
__global__ __launch_bounds__(256) void enc_tail_w2_kernel(TailSet S) {
  __shared__ float dzs[ETW_ROWS][64];
  __shared__ __attribute__((aligned(16))) float red[8][8][128];
  const ScpEncTail& m = S.b[blockIdx.z];
  const int tid = threadIdx.x, kq = tid & 31, slot = tid >> 5;
  const int H1 = m.H1, H2 = m.H2;
  if (blockIdx.x * 128 >= H1 || !m.dW2) return;    // uniform: a narrower branch of the same launch
  const TailWs wsl = tail_ws(m.B, H1, H2, m.N3, m.C);
  const float* dz2g = m.ws + wsl.o_dz2;
  float* slab = m.ws + wsl.o_slab;
  const int k = blockIdx.x * 128 + kq * 4;
  const int r0 = blockIdx.y * ETW_ROWS;
  const uint64_t sd1 = m.seed1 + (m.seed_dev ? m.seed_dev[0] : 0);
  for (int idx = tid; idx < ETW_ROWS * 64; idx += 256) {
    const int rr = idx >> 6, j = idx & 63;
    dzs[rr][j] = (r0 + rr < m.B && j < H2) ? dz2g[(long)(r0 + rr) * H2 + j] : 0.f;
  }
  const int kc = min(k, H1 - 4);
  const f4 gam = *(const f4*)(m.g1 + kc), bet = *(const f4*)(m.be1 + kc);
  // BatchNorm 1 between activation and dropout (scale 1, shift 0 without it)
  const BnOff bo = bn_off(H1, H2);
  const f4 one4 = f4{1.f, 1.f, 1.f, 1.f}, nul4 = f4{0.f, 0.f, 0.f, 0.f};
  const f4 bsc = m.bn ? *(const f4*)(m.bn_ws + bo.sc1 + kc) : one4, bsh = m.bn ? *(const f4*)(m.bn_ws + bo.sh1 + kc) : nul4;
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
      const f4 a1 = (leaky4(xr[i] * gam + bet, m.slope) * bsc + bsh) * keep4(sd1, (uint64_t)min(row, m.B - 1) * H1 + kc, m.p1);
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
  int seg_beg[8]; int seg_end[8]; float* seg_dst[8];
  float* dW2;
  // covariate-weight segment [wc_beg, wc_beg + C * H1), element (j, k) -> column j of row k of the first-layer weight:
  // written as a gradient (dW1c [H1][lddw1c]) or, adam != 0, consumed by the AdamW step of those weights in place
  int wc_beg, wc_end, H1, adam;
  float* dW1c; long lddw1c;
  float* W1c; long ldw1c; float* adam_m; float* adam_v; __bf16* adam_sh; long adam_ldsh;
  float lr, wd, b1, b2, eps, bc1, bc2, gscale; const float* bc_dev;
};
struct TailRedSet {
  TailRed b[SCP_ENC_TAIL_MAX_BRANCHES];
};
__global__ __launch_bounds__(256) void enc_tail_reduce_kernel(TailRedSet R) {
  // 64 outputs per workgroup, 4 threads per output: thread (o, q) sums records q, q + 4, ... (eight loads in flight),
  // the four partial sums are combined in a fixed order
  __shared__ float red[4][64];
  const TailRed& r = R.b[blockIdx.y];
  const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + o;
  if ((long)blockIdx.x * 64 >= r.pv + r.wn) return;      // uniform: a smaller branch of the same launch
  const bool in_part = i < r.pv;
  const long w = i - r.pv;
  const float* src = in_part ? r.part + i : r.slab + w;
  const long stride = in_part ? r.pv : r.wn;
  const int n = in_part ? r.nblk : ((w < r.wn && r.dW2) ? r.nchunk : 0);
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
    if (i >= r.wc_beg && i < r.wc_end) {
      const int idx = (int)(i - r.wc_beg), j = idx / r.H1, k = idx - j * r.H1;
      if (r.adam) {
        const float bc1 = r.bc_dev ? r.bc_dev[0] : r.bc1, bc2 = r.bc_dev ? r.bc_dev[1] : r.bc2;
        const AdamHyper hy = adam_hyper(r.lr, r.wd, r.b1, r.b2, r.eps, bc1, bc2, r.gscale);
        const long o = (long)k * r.ldw1c + j;
        float p_ = r.W1c[o], m_ = r.adam_m[o], v_ = r.adam_v[o];
        adamw_update_fast(hy, p_, s, m_, v_);      // same arithmetic as the epilogue that updates the other columns
        r.W1c[o] = p_;
        r.adam_m[o] = m_;
        r.adam_v[o] = v_;
        if (r.adam_sh) r.adam_sh[(long)k * r.adam_ldsh + j] = (__bf16)p_;
      } else if (r.dW1c) {
        r.dW1c[(long)k * r.lddw1c + j] = s;
      }
      return;
    }
    for (int g_ = 0; g_ < 8; g_++)
      if (i >= r.seg_beg[g_] && i < r.seg_end[g_] && r.seg_dst[g_]) r.seg_dst[g_][i - r.seg_beg[g_]] = s;
  } else {
    r.dW2[w] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------
static int tail_check(const ScpEncTail* ms, int nb, const char* who) {
  SCP_CHECK(ms && nb >= 1 && nb <= SCP_ENC_TAIL_MAX_BRANCHES, SCP_ERR_ARG, "%s: needs 1..%d branch descriptions", who,
            SCP_ENC_TAIL_MAX_BRANCHES);
  for (int b = 0; b < nb; b++) {
    const ScpEncTail* m = ms + b;
    SCP_CHECK(m->B > 0 && m->H1 > 0 && m->H2 > 0 && m->N3 >= 0, SCP_ERR_ARG, "%s: sizes must be positive", who);
    SCP_CHECK(m->B == ms[0].B, SCP_ERR_ARG, "%s: the branches of one launch share the minibatch", who);
    SCP_CHECK(m->H1 % 4 == 0 && m->H1 >= 4 && m->H1 <= ET_NCH * 256 && m->H2 <= 64 && m->N3 <= 64, SCP_ERR_SHAPE,
              "%s: needs H1 %% 4 == 0, H1 <= %d, H2 <= 64, N3 <= 64 (got %d, %d, %d)", who, ET_NCH * 256, m->H1, m->H2, m->N3);
    SCP_CHECK(m->g1 && m->be1 && m->W2 && m->b2 && m->g2 && m->be2, SCP_ERR_ARG, "%s: null parameter pointer", who);
    SCP_CHECK(m->N3 == 0 || (m->W3 && m->b3), SCP_ERR_ARG, "%s: the final Linear needs weight and bias", who);
    SCP_CHECK((((uintptr_t)m->b1 | (uintptr_t)m->g1 | (uintptr_t)m->be1 | (uintptr_t)m->W2 | (uintptr_t)m->xhat1) & 15) == 0,
              SCP_ERR_SHAPE, "%s: H1-wide tensors must be 16-byte aligned", who);
    SCP_CHECK(m->p1 >= 0.f && m->p1 < 1.f && m->p2 >= 0.f && m->p2 < 1.f, SCP_ERR_ARG, "%s: dropout p must be in [0,1)", who);
    SCP_CHECK(m->C >= 0 && m->C <= 64 && (m->C == 0 || (m->cov && m->W1c && m->ldcov >= m->C && m->ldw1c >= m->C)), SCP_ERR_ARG,
              "%s: covariate block needs cov [B][ldcov >= C], W1c with row stride >= C, C <= 64 (got C = %d)", who, m->C);
  }
  return SCP_OK;
}

// BatchNorm encoders: every branch of the call with the same mode, the workspaces present, the hand-over buffers present
static int tail_check_bn(const ScpEncTail* ms, int nb, const char* who) {
  for (int b = 0; b < nb; b++) {
    const ScpEncTail* m = ms + b;
    SCP_CHECK(m->bn == ms[0].bn && (m->bn_train != 0) == (ms[0].bn_train != 0), SCP_ERR_ARG,
              "%s: the branches of one call share bn / bn_train", who);
    SCP_CHECK(m->bn_ws && ((uintptr_t)m->bn_ws & 15) == 0, SCP_ERR_ARG, "%s: BatchNorm needs bn_ws (16-byte aligned)", who);
    SCP_CHECK(!m->bn_train || (m->xhat1 && m->xhat2 && m->rstd1 && m->rstd2), SCP_ERR_ARG,
              "%s: batch statistics need the xhat1 / xhat2 / rstd buffers (the segments hand over through them)", who);
    SCP_CHECK(m->bn_train || ((m->bn1_rm != nullptr) == (m->bn1_rv != nullptr) && (m->bn2_rm != nullptr) == (m->bn2_rv != nullptr)),
              SCP_ERR_ARG, "%s: running mean and variance go together", who);
    SCP_CHECK(m->bn_eps1 > 0.f && m->bn_eps2 > 0.f, SCP_ERR_ARG, "%s: BatchNorm eps must be > 0", who);
  }
  return SCP_OK;
}

static int tail_nch(const ScpEncTail* ms, int nb) {
  int h = 0;
  for (int b = 0; b < nb; b++) h = ms[b].H1 > h ? ms[b].H1 : h;
  return (h + 255) / 256;
}

extern "C" int scp_enc_tail_fwd(const ScpEncTail* ms, int nbranch, void* stream_) {
  int rc = tail_check(ms, nbranch, "scp_enc_tail_fwd");
  if (rc) return rc;
  TailSet S = {};
  for (int b = 0; b < nbranch; b++) {
    const ScpEncTail* m = ms + b;
    SCP_CHECK(m->slabs && m->out && m->nslab > 0 && m->ldo >= (m->N3 > 0 ? m->N3 : m->H2) && m->ld_slab >= m->H1,
              SCP_ERR_ARG, "scp_enc_tail_fwd: bad slab / output argument");
    SCP_CHECK(((uintptr_t)m->slabs & 15) == 0 && m->slab_stride % 4 == 0 && m->ld_slab % 4 == 0, SCP_ERR_SHAPE,
              "scp_enc_tail_fwd: slabs must be 16-byte aligned with strides that are multiples of 4");
    SCP_CHECK((m->xhat1 == nullptr) == (m->rstd1 == nullptr) && (m->xhat2 == nullptr) == (m->rstd2 == nullptr) &&
                  (m->xhat1 == nullptr) == (m->xhat2 == nullptr),
              SCP_ERR_ARG, "scp_enc_tail_fwd: the four saved-for-backward buffers go together");
    S.b[b] = *m;
  }
  const int nrec = (ms[0].B + ET_RW - 1) / ET_RW;
  const dim3 grid((nrec + ET_WAVES - 1) / ET_WAVES, nbranch);
  hipStream_t stream = (hipStream_t)stream_;
  if (ms[0].bn) {
    rc = tail_check_bn(ms, nbranch, "scp_enc_tail_fwd");
    if (rc) return rc;
    int max_h1 = 0, max_h2 = 0;
    for (int b = 0; b < nbranch; b++) {
      max_h1 = ms[b].H1 > max_h1 ? ms[b].H1 : max_h1;
      max_h2 = ms[b].H2 > max_h2 ? ms[b].H2 : max_h2;
    }
    const dim3 gs1((max_h1 + 15) / 16, nbranch), gs2((max_h2 + 15) / 16, nbranch);
#define ET_FWD_BN(N, SEG)                                                                                     \
  case N:                                                                                                     \
    hipLaunchKernelGGL((enc_tail_fwd_kernel<N, true>), grid, dim3(64 * ET_WAVES), 0, stream, S, SEG);          \
    break;
#define ET_FWD_SEG(SEG)                                                                                        \
  switch (tail_nch(ms, nbranch)) { ET_FWD_BN(1, SEG) ET_FWD_BN(2, SEG) ET_FWD_BN(3, SEG) ET_FWD_BN(4, SEG) }  \
  SCP_LAUNCH_CHECK();
    if (!ms[0].bn_train) {                         // running statistics: scale / shift of both layers, then one pass
      hipLaunchKernelGGL(enc_tail_bnstat_kernel, gs1, dim3(256), 0, stream, S, 1);
      SCP_LAUNCH_CHECK();
      hipLaunchKernelGGL(enc_tail_bnstat_kernel, gs2, dim3(256), 0, stream, S, 2);
      SCP_LAUNCH_CHECK();
      ET_FWD_SEG(0)
      return SCP_OK;
    }
    ET_FWD_SEG(1)
    hipLaunchKernelGGL(enc_tail_bnstat_kernel, gs1, dim3(256), 0, stream, S, 1);
    SCP_LAUNCH_CHECK();
    ET_FWD_SEG(2)
    hipLaunchKernelGGL(enc_tail_bnstat_kernel, gs2, dim3(256), 0, stream, S, 2);
    SCP_LAUNCH_CHECK();
    ET_FWD_SEG(3)
#undef ET_FWD_SEG
#undef ET_FWD_BN
    return SCP_OK;
  }
#define ET_FWD(N)                                                                                          \
  case N:                                                                                                  \
    hipLaunchKernelGGL((enc_tail_fwd_kernel<N, false>), grid, dim3(64 * ET_WAVES), 0, stream, S, 0);       \
    break;
  switch (tail_nch(ms, nbranch)) { ET_FWD(1) ET_FWD(2) ET_FWD(3) ET_FWD(4) }
#undef ET_FWD
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}

extern "C" long scp_enc_tail_ws_floats(int B, int H1, int H2, int N3, int C) { return tail_ws(B, H1, H2, N3, C).total; }
extern "C" long scp_enc_tail_bn_ws_floats(int H1, int H2) { return bn_off(H1, H2).total + 4; }

extern "C" int scp_enc_tail_bwd(const ScpEncTail* ms, int nbranch, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = tail_check(ms, nbranch, "scp_enc_tail_bwd");
  if (rc) return rc;
  TailSet S = {};
  TailRedSet R = {};
  int max_h1 = 0, any_w2 = 0;
  long max_total = 0;
  const int B = ms[0].B;
  for (int b = 0; b < nbranch; b++) {
    const ScpEncTail* m = ms + b;
    SCP_CHECK(m->dOut && (m->dY1 || m->dYT) && m->ws && m->xhat1 && m->rstd1 && m->xhat2 && m->rstd2, SCP_ERR_ARG,
              "scp_enc_tail_bwd: null pointer");
    SCP_CHECK(!m->dY1 || (m->ldy >= m->H1 && m->ldy % 4 == 0 && ((uintptr_t)m->dY1 & 15) == 0), SCP_ERR_SHAPE,
              "scp_enc_tail_bwd: dY1 must be 16-byte aligned, ldy %% 4 == 0");
    SCP_CHECK(!m->dYT || (m->ldt % 4 == 0 && m->ldt >= (B + 63) / 64 * 64 && ((uintptr_t)m->dYT & 7) == 0), SCP_ERR_SHAPE,
              "scp_enc_tail_bwd: dYT needs an 8-byte aligned base, ldt %% 4 == 0 and ldt >= cells rounded up to 64");
    SCP_CHECK(((uintptr_t)m->ws & 15) == 0, SCP_ERR_SHAPE, "scp_enc_tail_bwd: ws must be 16-byte aligned");
    SCP_CHECK(m->lddo >= (m->N3 > 0 ? m->N3 : m->H2), SCP_ERR_ARG, "scp_enc_tail_bwd: dOut leading dimension too small");
    S.b[b] = *m;
    const TailOff off = tail_off(m->H1, m->H2, m->N3, m->C);
    const TailWs w = tail_ws(B, m->H1, m->H2, m->N3, m->C);
    TailRed& r = R.b[b];
    r.part = m->ws; r.nblk = w.nrec; r.pv = off.pv; r.slab = m->ws + w.o_slab; r.nchunk = w.nchunk; r.wn = w.wn;
    r.dW2 = m->dW2;
    const int begs[8] = {off.g1, off.be1, off.b1, off.g2, off.be2, off.b2, off.w3, off.b3};
    const int ends[8] = {off.be1, off.b1, off.g2, off.be2, off.b2, off.w3, off.b3, off.b3 + m->N3};
    float* dsts[8] = {m->dg1, m->dbe1, m->db1, m->dg2, m->dbe2, m->db2, m->dW3, m->db3};
    for (int q = 0; q < 8; q++) { r.seg_beg[q] = begs[q]; r.seg_end[q] = ends[q]; r.seg_dst[q] = dsts[q]; }
    r.wc_beg = off.wc; r.wc_end = off.wc + m->C * m->H1; r.H1 = m->H1; r.adam = m->C > 0 ? m->adam : 0;
    r.dW1c = m->dW1c; r.lddw1c = m->lddw1c;
    if (r.adam) {
      SCP_CHECK(m->adam_m && m->adam_v && (m->bc_dev || (m->bc1 > 0.f && m->bc2 > 0.f)), SCP_ERR_ARG,
                "scp_enc_tail_bwd: the AdamW step of the covariate weights needs moments and bias corrections > 0");
      r.W1c = const_cast<float*>(m->W1c); r.ldw1c = m->ldw1c; r.adam_m = m->adam_m; r.adam_v = m->adam_v;
      r.adam_sh = (__bf16*)m->adam_sh; r.adam_ldsh = m->adam_ldsh;
      r.lr = m->lr; r.wd = m->wd; r.b1 = m->beta1; r.b2 = m->beta2; r.eps = m->adam_eps; r.bc1 = m->bc1; r.bc2 = m->bc2;
      r.gscale = m->gscale; r.bc_dev = m->bc_dev;
    } else {
      SCP_CHECK(m->C == 0 || !m->dW1c || m->lddw1c >= m->C, SCP_ERR_ARG, "scp_enc_tail_bwd: dW1c leading dimension too small");
    }
    max_h1 = m->H1 > max_h1 ? m->H1 : max_h1;
    any_w2 |= m->dW2 != nullptr;
    const long total = off.pv + w.wn;
    max_total = total > max_total ? total : max_total;
  }
  const int nrec = (B + ET_RW - 1) / ET_RW, nchunk = (B + ETW_ROWS - 1) / ETW_ROWS;
  const dim3 grid((nrec + ET_WAVES - 1) / ET_WAVES, nbranch);
  for (int b = 0; b < nbranch; b++)               // cells past the batch end of the padded transposed copy read as 0
    if (ms[b].dYT && B % 64) {
      SCP_HIP(hipMemsetAsync(ms[b].dYT, 0, (size_t)ms[b].H1 * ms[b].ldt * sizeof(__bf16), stream));
    }
  if (ms[0].bn) {
    rc = tail_check_bn(ms, nbranch, "scp_enc_tail_bwd");
    if (rc) return rc;
    int max_h2 = 0;
    for (int b = 0; b < nbranch; b++) {
      SCP_CHECK(ms[b].bn_dy1 && ((uintptr_t)ms[b].bn_dy1 & 15) == 0, SCP_ERR_ARG,
                "scp_enc_tail_bwd: BatchNorm needs bn_dy1 [B][H1] (16-byte aligned)");
      max_h2 = ms[b].H2 > max_h2 ? ms[b].H2 : max_h2;
    }
    const dim3 gs1((max_h1 + 15) / 16, nbranch), gs2((max_h2 + 15) / 16, nbranch);
#define ET_BWD_BN(N, SEG)                                                                                     \
  case N:                                                                                                     \
    hipLaunchKernelGGL((enc_tail_bwd_kernel<N, true>), grid, dim3(64 * ET_WAVES), 0, stream, S, SEG);          \
    break;
#define ET_BWD_SEG(SEG)                                                                                        \
  switch (tail_nch(ms, nbranch)) { ET_BWD_BN(1, SEG) ET_BWD_BN(2, SEG) ET_BWD_BN(3, SEG) ET_BWD_BN(4, SEG) }  \
  SCP_LAUNCH_CHECK();
    ET_BWD_SEG(1)
    hipLaunchKernelGGL(enc_tail_bnsum_kernel, gs2, dim3(256), 0, stream, S, 2);
    SCP_LAUNCH_CHECK();
    ET_BWD_SEG(2)
    hipLaunchKernelGGL(enc_tail_bnsum_kernel, gs1, dim3(256), 0, stream, S, 1);
    SCP_LAUNCH_CHECK();
    ET_BWD_SEG(3)
#undef ET_BWD_SEG
#undef ET_BWD_BN
  } else {
#define ET_BWD(N)                                                                              \
  case N:                                                                                      \
    hipLaunchKernelGGL((enc_tail_bwd_kernel<N, false>), grid, dim3(64 * ET_WAVES), 0, stream, S, 0);       \
    break;
  switch (tail_nch(ms, nbranch)) { ET_BWD(1) ET_BWD(2) ET_BWD(3) ET_BWD(4) }
#undef ET_BWD
  SCP_LAUNCH_CHECK();
  }
  if (any_w2) {
    hipLaunchKernelGGL(enc_tail_w2_kernel, dim3((max_h1 + 127) / 128, nchunk, nbranch), dim3(256), 0, stream, S);
    SCP_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(enc_tail_reduce_kernel, dim3((unsigned)((max_total + 63) / 64), nbranch), dim3(256), 0, stream, R);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}
