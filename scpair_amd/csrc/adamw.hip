// Fused multi-tensor AdamW (torch.optim.AdamW defaults as the reference uses them, model.py:698-701:
// betas (0.9, 0.999), eps 1e-8, amsgrad False, decoupled decay p *= 1 - lr*wd).
// HBM-bound: 16 B read (p, g, m, v; 14 B with bf16 gradients) + 12 B written (p, m, v) per parameter = 28 B/param/step
// (+2 B/param for tensors that carry a bf16 shadow copy for the first-layer MFMA kernels).
// Up to SCP_ADAM_MAX_TENSORS tensors per launch, descriptors passed by value (no host->device copy).
#include "common.h"
#include "scpair_hip.h"

#define ADAM_CHUNK 4096   // elements per workgroup: 256 threads x 4 x float4

struct AdamPack {
  ScpAdamTensor t[SCP_ADAM_MAX_TENSORS];
  int chunk_start[SCP_ADAM_MAX_TENSORS + 1];
  int n;
  float beta1, beta2, eps;
};

__global__ __launch_bounds__(256) void adamw_kernel(AdamPack pk) {
  int blk = blockIdx.x;
  int ti = 0;
#pragma unroll 1
  while (ti + 1 < pk.n && blk >= pk.chunk_start[ti + 1]) ti++;
  const ScpAdamTensor t = pk.t[ti];
  long base = (long)(blk - pk.chunk_start[ti]) * ADAM_CHUNK;
  // bias corrections: by value, or (HIP-graph replays) read from the device words scp_step_state_set rewrites per step
  const float bc1 = t.bc_dev ? t.bc_dev[0] : t.bias_corr1, bc2 = t.bc_dev ? t.bc_dev[1] : t.bias_corr2;
  const AdamHyper hy = adam_hyper(t.lr, t.weight_decay, pk.beta1, pk.beta2, pk.eps, bc1, bc2, t.grad_scale);
  const bool g16 = t.g_bf16 != 0;
  const __bf16* gh = (const __bf16*)t.g;
  bool vec = ((((uintptr_t)t.p | (uintptr_t)t.m | (uintptr_t)t.v) & 15) == 0) &&
             (((uintptr_t)t.g & (g16 ? 7 : 15)) == 0);
  auto gload = [&](long i) { return g16 ? (float)gh[i] : t.g[i]; };
  auto update = [&](float& p, float g, float& m, float& v) { adamw_update(hy, p, g, m, v); };
  auto shadow4 = [&](long i, const float (&pa)[4]) {
    __bf16* sh = (__bf16*)t.p_bf16;
    if ((t.cols & 3) == 0) {
      long rr = i / t.cols, cc = i - rr * t.cols;
      bf16x4 o;
#pragma unroll
      for (int k = 0; k < 4; k++) o[k] = (__bf16)pa[k];
      *(bf16x4*)(sh + rr * t.ld_bf16 + cc) = o;
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        long rr = (i + k) / t.cols, cc = (i + k) - rr * t.cols;
        sh[rr * t.ld_bf16 + cc] = (__bf16)pa[k];
      }
    }
  };
  // p / m / v index of gradient element i: the same, or (rows padded to ld_p elements) row * ld_p + column
  const bool padded = t.ld_p > t.cols && t.cols > 0;
  auto pidx = [&](long i) {
    if (!padded) return i;
    long rr = i / t.cols;
    return rr * t.ld_p + (i - rr * t.cols);
  };
  if (padded) vec = vec && (t.cols & 3) == 0 && (t.ld_p & 3) == 0;   // a float4 never straddles a row
  if (vec && base + ADAM_CHUNK <= t.n) {
    // interior chunk: all 16 vector loads of the thread are issued before the first store (64 B x 4 streams in
    // flight per lane); the streams are read and written once, so they bypass the caches' retention (nontemporal)
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 p[4], g[4], m[4], v[4];
    long pi[4];
#pragma unroll
    for (int it = 0; it < 4; it++) {
      long i = base + (long)it * 1024 + threadIdx.x * 4;
      pi[it] = pidx(i);
      p[it] = __builtin_nontemporal_load((const f4*)(t.p + pi[it]));
      if (g16) {
        bf16x4 gb = __builtin_nontemporal_load((const bf16x4*)(gh + i));
        g[it] = f4{(float)gb[0], (float)gb[1], (float)gb[2], (float)gb[3]};
      } else {
        g[it] = __builtin_nontemporal_load((const f4*)(t.g + i));
      }
      m[it] = __builtin_nontemporal_load((const f4*)(t.m + pi[it]));
      v[it] = __builtin_nontemporal_load((const f4*)(t.v + pi[it]));
    }
#pragma unroll
    for (int it = 0; it < 4; it++) {
      long i = base + (long)it * 1024 + threadIdx.x * 4;
      float pa[4], ma[4], va[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        pa[k] = p[it][k]; ma[k] = m[it][k]; va[k] = v[it][k];
        update(pa[k], g[it][k], ma[k], va[k]);
      }
      f4 po = {pa[0], pa[1], pa[2], pa[3]}, mo = {ma[0], ma[1], ma[2], ma[3]}, vo = {va[0], va[1], va[2], va[3]};
      __builtin_nontemporal_store(po, (f4*)(t.p + pi[it]));
      if (t.p_bf16) shadow4(i, pa);
      __builtin_nontemporal_store(mo, (f4*)(t.m + pi[it]));
      __builtin_nontemporal_store(vo, (f4*)(t.v + pi[it]));
    }
    return;
  }
#pragma unroll 1
  for (int it = 0; it < 4; it++) {
    long i = base + (long)it * 1024 + threadIdx.x * 4;
    if (i >= t.n) break;
    if (vec && i + 4 <= t.n) {
      const long q = pidx(i);
      float4 p = *(float4*)(t.p + q), m = *(float4*)(t.m + q), v = *(float4*)(t.v + q);
      float pa[4] = {p.x, p.y, p.z, p.w}, ga[4] = {gload(i), gload(i + 1), gload(i + 2), gload(i + 3)}, ma[4] = {m.x, m.y, m.z, m.w},
            va[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; k++) update(pa[k], ga[k], ma[k], va[k]);
      *(float4*)(t.p + q) = make_float4(pa[0], pa[1], pa[2], pa[3]);
      if (t.p_bf16) shadow4(i, pa);
      *(float4*)(t.m + q) = make_float4(ma[0], ma[1], ma[2], ma[3]);
      *(float4*)(t.v + q) = make_float4(va[0], va[1], va[2], va[3]);
    } else {
      for (int k = 0; k < 4 && i + k < t.n; k++) {
        const long q = pidx(i + k);
        float p = t.p[q], m = t.m[q], v = t.v[q];
        update(p, gload(i + k), m, v);
        t.p[q] = p;
        if (t.p_bf16) {
          long rr = (i + k) / t.cols, cc = (i + k) - rr * t.cols;
          ((__bf16*)t.p_bf16)[rr * t.ld_bf16 + cc] = (__bf16)p;
        }
        t.m[q] = m;
        t.v[q] = v;
      }
    }
  }
}

extern "C" int scp_adamw_multi(const ScpAdamTensor* tensors, int n, float beta1, float beta2, float eps,
                               void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SCP_CHECK(tensors && n > 0, SCP_ERR_ARG, "scp_adamw_multi: no tensors");
  int i = 0;                                 // input cursor, carried across launches (zero-size tensors take no slot)
  while (i < n) {
    AdamPack pk;
    pk.n = 0;
    pk.beta1 = beta1; pk.beta2 = beta2; pk.eps = eps;
    int chunks = 0;
    for (; i < n && pk.n < SCP_ADAM_MAX_TENSORS; i++) {
      const ScpAdamTensor& t = tensors[i];
      SCP_CHECK(t.p && t.g && t.m && t.v, SCP_ERR_ARG, "scp_adamw_multi: null pointer in tensor %d", i);
      SCP_CHECK(t.n >= 0, SCP_ERR_ARG, "scp_adamw_multi: negative size in tensor %d", i);
      SCP_CHECK(!t.p_bf16 || (t.cols > 0 && t.ld_bf16 >= t.cols && t.ld_bf16 % 8 == 0), SCP_ERR_ARG,
                "scp_adamw_multi: bad bf16 shadow geometry in tensor %d", i);
      SCP_CHECK(t.bc_dev || (t.bias_corr1 > 0.f && t.bias_corr2 > 0.f), SCP_ERR_ARG,
                "scp_adamw_multi: bias corrections must be > 0");
      SCP_CHECK(t.ld_p == 0 || (t.cols > 0 && t.ld_p >= t.cols && t.n % t.cols == 0), SCP_ERR_ARG,
                "scp_adamw_multi: padded rows need cols > 0, ld_p >= cols and n a multiple of cols (tensor %d)", i);
      if (t.n == 0) continue;
      pk.t[pk.n] = t;
      pk.chunk_start[pk.n] = chunks;
      chunks += (int)((t.n + ADAM_CHUNK - 1) / ADAM_CHUNK);
      pk.n++;
    }
    pk.chunk_start[pk.n] = chunks;
    if (chunks == 0) continue;
    hipLaunchKernelGGL(adamw_kernel, dim3(chunks), dim3(256), 0, stream, pk);
    SCP_LAUNCH_CHECK();
  }
  return SCP_OK;
}

// ---- per-step state for HIP-graph replays ---------------------------------------------------------------------
// state: 16 bytes of device memory = { uint64 seed_offset ; float bias_corr1 ; float bias_corr2 }.  A captured training
// step bakes every by-value argument, so what changes from step to step (the dropout seed, AdamW's bias corrections)
// is read by the kernels from these words (seed_dev / bc_dev arguments) and rewritten here before each replay.
__global__ void step_state_kernel(uint64_t* __restrict__ state, uint64_t seed_offset, float bc1, float bc2) {
  state[0] = seed_offset;
  float* f = (float*)(state + 1);
  f[0] = bc1;
  f[1] = bc2;
}

extern "C" int scp_step_state_set(void* state, uint64_t seed_offset, float bias_corr1, float bias_corr2, void* stream_) {
  SCP_CHECK(state && ((uintptr_t)state % 8) == 0, SCP_ERR_ARG, "scp_step_state_set: state must be an 8-byte aligned device pointer");
  SCP_CHECK(bias_corr1 > 0.f && bias_corr2 > 0.f, SCP_ERR_ARG, "scp_step_state_set: bias corrections must be > 0");
  hipLaunchKernelGGL(step_state_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_, (uint64_t*)state, seed_offset,
                     bias_corr1, bias_corr2);
  SCP_LAUNCH_CHECK();
  return SCP_OK;
}
