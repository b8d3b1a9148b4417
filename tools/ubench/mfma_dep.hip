// Issue rate of v_mfma_f32_32x32x16_bf16 when consecutive instructions accumulate into the SAME tile
// (dependent chain, as the 8 k-steps of one S tile) vs. round-robin over NACC independent tiles.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k(float* out, long long* cyc, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  f32x16 acc[NACC];
  for (int u = 0; u < NACC; ++u) for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u % NACC]) : "v"(a), "v"(b));
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int u = 0; u < NACC; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * WAVES + threadIdx.x / 64] = t1 - t0;
}

template <int NACC, int WAVES> void run() {
  float* out; long long* cyc; const int blocks = 256, iters = 2000;
  hipMalloc(&out, blocks * WAVES * 64 * 4); hipMalloc(&cyc, blocks * WAVES * 8);
  hipLaunchKernelGGL((k<NACC, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, out, cyc, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[WAVES]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double flops = 2.0 * 32 * 32 * 16 * 16.0 * iters * blocks * WAVES;
  printf("nacc=%d waves/CU=%d: %.3f ms  %.0f TF  clock64 ticks per MFMA (wave0) %.2f\n", NACC, WAVES, ms, flops / ms / 1e9, (double)h[0] / (16.0 * iters));
  hipFree(out); hipFree(cyc);
}
int main() {
  run<1, 4>(); run<2, 4>(); run<4, 4>(); run<1, 8>(); run<2, 8>(); run<4, 8>();
  return 0;
}
