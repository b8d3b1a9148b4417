// What shader clock does the chip sustain under a dense MFMA stream?  One wave per SIMD on every CU issues back-to-back
// v_mfma_f32_32x32x16_bf16 (8 independent accumulators) for a few milliseconds; every wave reads the shader clock counter
// (s_memtime) and the constant 100 MHz counter (s_memrealtime) before and after.  Operand data: constant (1.0 / 0.5), or
// pseudo-random bf16 in [-2, 2) refreshed from a small register pool -- switching activity (and so power) depends on it.
// Output: effective MHz, ticks per MFMA, TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int RANDOM, int FILL>
__global__ void __launch_bounds__(256, 1) k(float* out, long long* cyc, int iters) {
  u32x4 a[4], b[4];
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 4; ++i) {
      if (RANDOM) {
        s = s * 1664525u + 1013904223u; const unsigned x = s;
        s = s * 1664525u + 1013904223u; const unsigned y = s;
        // bf16 pairs: sign random, exponent 125..127 (|x| in [0.25, 2)), mantissa random
        a[j][i] = (x & 0x807f807fu) | 0x3e803e80u | ((x >> 3) & 0x01000100u);
        b[j][i] = (y & 0x807f807fu) | 0x3e803e80u | ((y >> 3) & 0x01000100u);
      } else { a[j][i] = 0x3f803f80u; b[j][i] = 0x3f003f00u; }
    }
  f32x16 acc[8];
  for (int u = 0; u < 8; ++u) for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = threadIdx.x * 0.001f + i;
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u % 8]) : "v"(a[u % 4]), "v"(b[(u / 4) % 4]));
#pragma unroll
      for (int j = 0; j < FILL; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[(u * FILL + j) % 8]) : "v"(f[(j + 1) % 8]));
    }
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  float sum = 0.f;
  for (int u = 0; u < 8; ++u) for (int r = 0; r < 16; ++r) sum += acc[u][r];
  for (int i = 0; i < 8; ++i) sum += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if ((threadIdx.x & 63) == 0) { cyc[(blockIdx.x * 4 + threadIdx.x / 64) * 2] = t1 - t0; cyc[(blockIdx.x * 4 + threadIdx.x / 64) * 2 + 1] = w1 - w0; }
}

template <int RANDOM, int FILL> void run(int blocks, int iters) {
  float* out; long long* cyc;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 4 * 16);
  hipLaunchKernelGGL((k<RANDOM, FILL>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<RANDOM, FILL>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double flops = 2.0 * 32 * 32 * 16 * 16.0 * iters * blocks * 4;
  printf("%s data, %d fillers/MFMA, %d workgroups: %.3f ms %5.0f TF  ticks/MFMA %.1f  shader clock %.0f MHz\n", RANDOM ? "random  " : "constant", FILL, blocks,
         ms, flops / ms / 1e9, (double)h[0] / (16.0 * iters), (double)h[0] / ((double)h[1] / 100.0));
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 0>(256, 20000); run<1, 0>(256, 20000); run<1, 3>(256, 20000); run<0, 3>(256, 20000);
    run<1, 0>(64, 20000);
  }
  return 0;
}
