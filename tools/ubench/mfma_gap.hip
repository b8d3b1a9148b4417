// One wave per SIMD: cycles per v_mfma_f32_32x32x16_bf16 as a function of (a) how many independent accumulators the
// MFMAs rotate over (dependent distance NACC), (b) how many single-issue fillers sit in each gap between two MFMAs
// (FILL plain v_add_f32 on private registers, of which TRANS are v_exp_f32), (c) accumulators in VGPRs or AGPRs.
// Question behind it: does a lone wave hide ~5 fillers per MFMA gap, and does the answer depend on the dependent distance?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC, int FILL, int TRANS, bool AG>
__global__ void __launch_bounds__(256, 1) k(float* out, long long* cyc, int iters) {
  u32x4 a, b;
  for (int i = 0; i < 4; ++i) { a[i] = 0x3f803f80u + threadIdx.x; b[i] = 0x3f003f00u + i; }
  f32x16 acc[NACC];
  for (int u = 0; u < NACC; ++u) {
    for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    if (AG) asm volatile("" : "=a"(acc[u]) : "0"(acc[u]));
  }
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = threadIdx.x * 0.001f + i;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[u % NACC]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u % NACC]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < FILL; ++j) {
        if (j < TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(f[(u * FILL + j) % 8]));
        else asm volatile("v_add_f32 %0, %0, %0" : "+v"(f[(u * FILL + j) % 8]));
      }
    }
  }
  long long t1 = clock64();
  float s = 0.f;
  if (AG) asm volatile("s_nop 15\n\ts_nop 3");
  for (int u = 0; u < NACC; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

template <int NACC, int FILL, int TRANS, bool AG> void run() {
  float* out; long long* cyc; const int blocks = 256, iters = 1000;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 4 * 8);
  hipLaunchKernelGGL((k<NACC, FILL, TRANS, AG>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, FILL, TRANS, AG>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double flops = 2.0 * 32 * 32 * 16 * 16.0 * iters * blocks * 4;
  printf("nacc=%d fill=%d trans=%d %s: %.3f ms %5.0f TF  ticks/MFMA %.1f\n", NACC, FILL, TRANS, AG ? "agpr" : "vgpr", ms, flops / ms / 1e9, (double)h[0] / (16.0 * iters));
  hipFree(out); hipFree(cyc);
}
template <int NACC, bool AG> void sweep() {
  run<NACC, 0, 0, AG>(); run<NACC, 1, 0, AG>(); run<NACC, 2, 0, AG>(); run<NACC, 3, 0, AG>(); run<NACC, 4, 0, AG>(); run<NACC, 5, 0, AG>();
  run<NACC, 6, 0, AG>(); run<NACC, 4, 1, AG>(); run<NACC, 5, 1, AG>(); run<NACC, 5, 2, AG>();
}
int main() {
  sweep<1, false>(); sweep<2, false>(); sweep<3, false>(); sweep<4, false>(); sweep<8, false>(); sweep<2, true>(); sweep<8, true>();
  return 0;
}
