// HBM access-pattern probe for the AdamW epilogue of the first-layer weight-gradient kernel: every workgroup (512
// threads) reads a TM x TN fp32 tile of three [H][F] arrays row by row (TN*4-byte segments, row stride F*4 bytes),
// updates and writes them back plus a bf16 copy: 26 B per element.  Reports TB/s for several tile shapes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int TM, int TN, int EPB>
__global__ __launch_bounds__(512) void tile_stream(float* __restrict__ W, float* __restrict__ M, float* __restrict__ V,
                                                   __bf16* __restrict__ S, int H, int F, int tiles_m, int tiles_n) {
  int b = blockIdx.x;
  int xcd = b & 7, j = b >> 3;
  int m_tile = j % tiles_m;
  int unit = (j / tiles_m) * 8 + xcd;
  if (unit >= tiles_n) return;
  const int m0 = m_tile * TM, n0 = unit * TN;
  constexpr int LPR = TN / 4;              // lanes per row segment
  constexpr int RPP = 512 / LPR;           // rows per pass
  const int tid = threadIdx.x, c4 = (tid % LPR) * 4, rw = tid / LPR;
  const int n = n0 + c4;
  const bool col_ok = n + 4 <= F;
  for (int pass = 0; pass < TM / RPP; pass += EPB) {
    f4 pp[EPB], mm[EPB], vv[EPB];
#pragma unroll
    for (int u = 0; u < EPB; u++) {
      int m = m0 + (pass + u) * RPP + rw;
      if (col_ok && m < H) {
        long o = (long)m * F + n;
        pp[u] = __builtin_nontemporal_load((const f4*)(W + o));
        mm[u] = __builtin_nontemporal_load((const f4*)(M + o));
        vv[u] = __builtin_nontemporal_load((const f4*)(V + o));
      }
    }
#pragma unroll
    for (int u = 0; u < EPB; u++) {
      int m = m0 + (pass + u) * RPP + rw;
      if (col_ok && m < H) {
        long o = (long)m * F + n;
        f4 po = pp[u] * 0.999f + mm[u] * 0.001f, mo = mm[u] * 0.9f + 0.1f, vo = vv[u] * 0.999f + 0.001f;
        __builtin_nontemporal_store(po, (f4*)(W + o));
        __builtin_nontemporal_store(mo, (f4*)(M + o));
        __builtin_nontemporal_store(vo, (f4*)(V + o));
        bf16x4 sh;
#pragma unroll
        for (int k = 0; k < 4; k++) sh[k] = (__bf16)po[k];
        *(bf16x4*)(S + o) = sh;
      }
    }
  }
}

template <int TM, int TN, int EPB>
static void run(const char* name, float* W, float* M, float* V, __bf16* S, int H, int F) {
  int tiles_m = (H + TM - 1) / TM, tiles_n = (F + TN - 1) / TN;
  int grid = 8 * tiles_m * ((tiles_n + 7) / 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((tile_stream<TM, TN, EPB>), dim3(grid), dim3(512), 0, 0, W, M, V, S, H, F, tiles_m, tiles_n);
  hipEventRecord(e0);
  const int rep = 10;
  for (int i = 0; i < rep; i++) hipLaunchKernelGGL((tile_stream<TM, TN, EPB>), dim3(grid), dim3(512), 0, 0, W, M, V, S, H, F, tiles_m, tiles_n);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double bytes = 26.0 * H * F;
  printf("%-28s grid %6d  %8.1f us  %6.2f TB/s\n", name, grid, ms / rep * 1e3, bytes / (ms / rep * 1e-3) / 1e12);
}

int main() {
  const int H = 900, F = 100000;
  float *W, *M, *V; __bf16* S;
  size_t n = (size_t)H * F;
  hipMalloc(&W, n * 4); hipMalloc(&M, n * 4); hipMalloc(&V, n * 4); hipMalloc(&S, n * 2);
  hipMemset(W, 0, n * 4); hipMemset(M, 0, n * 4); hipMemset(V, 0, n * 4);
  run<128, 256, 16>("128 x 256 (K2' today)", W, M, V, S, H, F);
  run<128, 256, 8>("128 x 256, EPB 8", W, M, V, S, H, F);
  run<64, 512, 16>("64 x 512", W, M, V, S, H, F);
  run<32, 1024, 16>("32 x 1024", W, M, V, S, H, F);
  run<16, 2048, 16>("16 x 2048", W, M, V, S, H, F);
  run<256, 128, 16>("256 x 128", W, M, V, S, H, F);
  run<128, 512, 16>("128 x 512", W, M, V, S, H, F);
  return 0;
}
