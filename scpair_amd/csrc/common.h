// Shared device helpers for the scPair hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SCP_WAVE 64

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- error plumbing (host side) -----------------------------------------------------------
void scp_set_error(const char* fmt, ...);
#define SCP_CHECK(cond, code, ...)            \
  do {                                        \
    if (!(cond)) {                            \
      scp_set_error(__VA_ARGS__);             \
      return (code);                          \
    }                                         \
  } while (0)
#define SCP_HIP(expr)                                                        \
  do {                                                                       \
    hipError_t e_ = (expr);                                                  \
    if (e_ != hipSuccess) {                                                  \
      scp_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                    __FILE__, __LINE__);                                     \
      return -100;                                                           \
    }                                                                        \
  } while (0)
#define SCP_LAUNCH_CHECK() SCP_HIP(hipGetLastError())

enum {
  SCP_OK = 0,
  SCP_ERR_ARG = -1,      // bad argument (null pointer, negative size, unsupported flag)
  SCP_ERR_SHAPE = -2,    // shape outside what the kernel supports
  SCP_ERR_HIP = -100     // HIP runtime error, see scp_last_error()
};

// Opt a kernel in to `smem` bytes of dynamic LDS.  The attribute is per device: `done` is a bit mask over device
// indices (one static word per kernel instantiation at the call site).
static inline int scp_set_smem(const void* fn, size_t smem, uint64_t* done) {
  int dev = 0;
  SCP_HIP(hipGetDevice(&dev));
  uint64_t bit = 1ull << (dev & 63);
  if (!(*done & bit)) {
    SCP_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    *done |= bit;
  }
  return SCP_OK;
}

// ---- bf16 split: f = hi + mid + lo (each bf16), 24 significant bits in total ----------------
template <int NS>
__device__ __forceinline__ void bf16_split(float f, __bf16* out) {
  __bf16 h = (__bf16)f;
  out[0] = h;
  if (NS > 1) {
    float r = f - (float)h;
    __bf16 m = (__bf16)r;
    out[1] = m;
    if (NS > 2) {
      float r2 = r - (float)m;
      out[2] = (__bf16)r2;
    }
  }
}

// Which (a-term, b-term) products are kept for an NS-term split: all pairs with i + j < NS
// NS=1: (0,0)            -> bf16 inputs, fp32 accumulate
// NS=2: (0,0)(0,1)(1,0)  -> ~16 significant bits per operand
// NS=3: 6 products       -> fp32-equivalent (dropped terms <= 2^-24 relative)

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// C/D element mapping of the 32x32 MFMA accumulator (dtype independent on gfx950):
//   col = lane & 31 ; row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// ---- AdamW element update (torch.optim.AdamW, amsgrad False), shared by the multi-tensor optimiser kernel and
// the first-layer weight-gradient kernel that applies the update in its epilogue -----------------------------
struct AdamHyper {
  float decay;          // 1 - lr * weight_decay
  float step_size;      // lr / (1 - beta1^t)
  float inv_sqrt_bc2;   // 1 / sqrt(1 - beta2^t)
  float b1, b2, eps, grad_scale;
};
__device__ __forceinline__ AdamHyper adam_hyper(float lr, float wd, float b1, float b2, float eps, float bc1, float bc2,
                                                float grad_scale) {
  AdamHyper h;
  h.decay = 1.f - lr * wd;
  h.step_size = lr / bc1;
  h.inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  h.b1 = b1; h.b2 = b2; h.eps = eps; h.grad_scale = grad_scale;
  return h;
}
__device__ __forceinline__ void adamw_update(const AdamHyper& h, float& p, float g, float& m, float& v) {
  float gk = g * h.grad_scale;
  p *= h.decay;
  m = m + (gk - m) * (1.f - h.b1);
  v = v * h.b2 + gk * gk * (1.f - h.b2);
  float denom = sqrtf(v) * h.inv_sqrt_bc2 + h.eps;
  p -= h.step_size * (m / denom);
}

// The same update with the hardware square root and reciprocal (v_sqrt_f32 / v_rcp_f32, ~1 ulp each) instead of the
// correctly rounded sqrtf and division, whose expansions are two thirds of the exact update's ~37 instructions: for
// the epilogue of the first-layer weight-gradient kernel (precision mode 'bf16'), which spends a third of its
// vector-issue time here.  Relative difference to adamw_update per step: ~2e-7 of the step, i.e. ~1e-9 of a weight.
__device__ __forceinline__ void adamw_update_fast(const AdamHyper& h, float& p, float g, float& m, float& v) {
  float gk = g * h.grad_scale;
  p *= h.decay;
  m = m + (gk - m) * (1.f - h.b1);
  v = v * h.b2 + gk * gk * (1.f - h.b2);
  float denom = __builtin_amdgcn_sqrtf(v) * h.inv_sqrt_bc2 + h.eps;
  p -= h.step_size * (m * __builtin_amdgcn_rcpf(denom));
}

// ---- wave / block reductions -------------------------------------------------------------
// Cross-lane moves on the vector ALU (no LDS round trip like __shfl_xor's ds_bpermute): DPP within a row of 16 lanes,
// gfx950's permlane swaps across rows.  CTRL: 0xB1 = quad_perm [1,0,3,2] (lane ^ 1), 0x4E = quad_perm [2,3,0,1]
// (lane ^ 2), 0x120 + n = row_ror n.  All lanes of the wave must be active.
template <int CTRL> __device__ __forceinline__ float dppf(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float xor16_sum(float v) {      // v[lane] + v[lane ^ 16]
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {      // v[lane] + v[lane ^ 32]
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// sum over the 32 lanes that share (lane >> 5)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum (blockDim.x multiple of 64, <= 1024); result valid in thread 0. `red` >= 16 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x < 64) {
    r = (l < (int)(blockDim.x >> 6)) ? red[l] : 0.f;
    r = wave_sum(r);
  }
  return r;
}
__device__ __forceinline__ double block_sum_d(double v, double* red) {
  v = wave_sum_d(v);
  int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x < 64) {
    r = (l < (int)(blockDim.x >> 6)) ? red[l] : 0.0;
    r = wave_sum_d(r);
  }
  return r;
}

// p[r * stride] summed over r = y, y+16, y+32, ... < n, four loads in flight, fixed order
__device__ __forceinline__ float strided_sum16(const float* __restrict__ p, long stride, int n, int y) {
  float s = 0.f;
  int r = y;
  for (; r + 48 < n; r += 64) {
    float a0 = p[(long)r * stride], a1 = p[(long)(r + 16) * stride], a2 = p[(long)(r + 32) * stride],
          a3 = p[(long)(r + 48) * stride];
    s += a0;
    s += a1;
    s += a2;
    s += a3;
  }
  for (; r < n; r += 16) s += p[(long)r * stride];
  return s;
}

// ---- special functions -----------------------------------------------------------------------
// softplus with torch's threshold (F.softplus default beta=1, threshold=20)
__device__ __forceinline__ float softplus_t(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// digamma for x > 0 (the only domain the likelihood kernels need): recurrence up to x >= 6, then
// the asymptotic series  ln x - 1/(2x) - 1/(12x^2) + 1/(120x^4) - 1/(252x^6) + 1/(240x^8) - 1/(132x^10)
__device__ __forceinline__ float digammaf_pos(float x) {
  float acc = 0.f;
  while (x < 6.f) {
    acc -= 1.f / x;
    x += 1.f;
  }
  float r = 1.f / x, r2 = r * r;
  float ser = r2 * (1.f / 12.f - r2 * (1.f / 120.f - r2 * (1.f / 252.f - r2 * (1.f / 240.f - r2 * (1.f / 132.f)))));
  return acc + logf(x) - 0.5f * r - ser;
}

// lgamma(x+theta) - lgamma(theta) - lgamma(x+1) for integer-valued x >= 1, computed so that the
// catastrophic cancellation lgamma(x+theta)-lgamma(theta) at large theta is avoided for small x:
//   lgamma(x+theta)-lgamma(theta) = sum_{i<x} log(theta+i)   (exact identity for integer x)
// For large x fall back to lgammaf differences (what the reference computes).
__device__ __forceinline__ float lgamma_terms(float x, float theta) {
  return lgammaf(x + theta) - lgammaf(theta) - lgammaf(x + 1.f);
}

// math flavour: accurate libm in the fp32-equivalent mode (NS=3), hardware exp/log/rcp in bf16 mode (NS=1)
template <bool FAST> __device__ __forceinline__ float exp_(float x) { return FAST ? __expf(x) : expf(x); }
// FAST: the bare v_log_f32 (log2, ~1 ulp) times ln 2 -- two instructions; __logf() expands to twelve (denormal
// rescaling and an extended-precision product).  Denormal arguments flush to log(0) = -inf, which every caller clamps.
template <bool FAST> __device__ __forceinline__ float log_(float x) {
  return FAST ? __builtin_amdgcn_logf(x) * 0.69314718f : logf(x);
}
template <bool FAST> __device__ __forceinline__ float div_(float a, float b) { return FAST ? a * __builtin_amdgcn_rcpf(b) : a / b; }
template <bool FAST> __device__ __forceinline__ float sigm_(float x) { return div_<FAST>(1.f, 1.f + exp_<FAST>(-x)); }

// lgamma(y+th) - lgamma(th) - lgamma(y+1) and digamma(y+th) - digamma(th).  For small integer counts the exact
// identities  sum_{i<y} log((th+i)/(i+1))  and  sum_{i<y} 1/(th+i)  avoid both the slow lgammaf/digamma evaluations
// and the cancellation of lgamma(y+th)-lgamma(th) at large theta.  The sums are evaluated without a transcendental
// per term: the log of a running product (one log at the end; the product of <= 12 factors <= 1035 fits fp32) and
// a running fraction N/D (one division at the end).  For th > 1024 the factors are expanded around th instead
// (i/th < 0.011: three series terms leave < 4e-9 per factor).
template <bool FAST>
__device__ __forceinline__ float lgam3(float y, float th) {
  if (y <= 12.f && y == floorf(y)) {
    float den = 1.f;
    if (th > 1024.f) {
      for (float i = 1.f; i < y; i += 1.f) den *= i + 1.f;
      float it = div_<FAST>(1.f, th);
      float s1 = 0.5f * y * (y - 1.f);                          // sum i
      float s2 = (y - 1.f) * y * (2.f * y - 1.f) * (1.f / 6.f);  // sum i^2
      float s3 = s1 * s1;                                       // sum i^3
      return y * log_<FAST>(th) + it * (s1 - it * (0.5f * s2 - it * (1.f / 3.f) * s3)) - log_<FAST>(den);
    }
    float num = 1.f;
    for (float i = 0.f; i < y; i += 1.f) {
      num *= th + i;
      den *= i + 1.f;
    }
    return log_<FAST>(div_<FAST>(num, den));
  }
  return lgammaf(y + th) - lgammaf(th) - lgammaf(y + 1.f);
}
template <bool FAST>
__device__ __forceinline__ float digam2(float y, float th) {
  if (y <= 12.f && y == floorf(y)) {
    if (th > 1024.f) {
      float it = div_<FAST>(1.f, th);
      float s1 = 0.5f * y * (y - 1.f);
      float s2 = (y - 1.f) * y * (2.f * y - 1.f) * (1.f / 6.f);
      return it * (y - it * (s1 - it * s2));
    }
    float n = 0.f, d = 1.f;
    for (float i = 0.f; i < y; i += 1.f) {
      float x = th + i;
      n = n * x + d;
      d *= x;
    }
    return div_<FAST>(n, d);
  }
  return digammaf_pos(y + th) - digammaf_pos(th);
}

// counter-based RNG for dropout masks: one 32-bit hash per (seed, element index); reproducible in backward.
// The seed goes through a 64-bit mixer (splitmix64 finaliser) once -- it is uniform over the launch, so this runs on
// the scalar unit -- and every element then costs two 32-bit multiply rounds (the 'lowbias32' integer hash) instead of
// three 64-bit multiplies: full-width integer multiplies issue at a quarter of the vector rate, and the fused tail
// kernels hash ~1800 elements per cell in each direction.
__device__ __forceinline__ uint32_t hash_u32(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x632BE59BD9B4E019ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  uint32_t x = (uint32_t)idx + (uint32_t)z + (uint32_t)(idx >> 32) * 0x85EBCA6Bu;
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x ^ (uint32_t)(z >> 32);
}
// dropout seed of a launch = by-value base + an optional device-resident per-step offset (HIP-graph replays: the graph
// bakes the by-value arguments, the offset word is rewritten before every replay by scp_step_state_set)
struct SeedArg {
  uint64_t base;
  const uint64_t* dev;
};
__device__ __forceinline__ uint64_t seed_of(const SeedArg& s) { return s.dev ? s.base + s.dev[0] : s.base; }
__device__ __forceinline__ float keep_scale(uint64_t seed, uint64_t idx, float p) {
  if (p <= 0.f) return 1.f;
  float u = (float)(hash_u32(seed, idx) >> 8) * (1.0f / 16777216.0f);
  return u >= p ? 1.f / (1.f - p) : 0.f;
}
