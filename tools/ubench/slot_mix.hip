// What breaks MFMA/VALU overlap in an attention-like instruction slot?  One slot = [2 ds_read_b128] [s_waitcnt]
// [v_mfma 32x32x16] [K plain VALU] [T transcendental]; 16 slots per iteration, 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define LDSP __attribute__((address_space(3)))

// MODE bits: 1 = ds_reads, 2 = waitcnt before each MFMA, 4 = MFMA operands come from the LDS ring, 8 = VALU reads MFMA results of the previous iteration
template <int MODE, int K, int T, int WAVES, int RD4 = 8, int WC4 = 4>
__global__ void __launch_bounds__(WAVES * 64) kern(float* out, long long* cyc, int iters) {
  extern __shared__ char smem[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((float*)smem)[i] = 0.001f * i;
  __syncthreads();
  const unsigned base = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
  u32x4 ring[8];
  for (int i = 0; i < 8; ++i) ring[i] = u32x4{0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  f32x16 accS = {}, accO[4] = {};
  f32x16 prev;
  for (int r = 0; r < 16; ++r) prev[r] = 0.01f * r + threadIdx.x * 1e-4f;
  float vs = 0.f;
  const float cs = 1.0001f, nm = -0.5f;
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.01f * i); fb[i] = (__bf16)(0.02f * i); }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (MODE & 1) {  // RD4 reads per 4 slots
        constexpr int pat[4] = {(RD4 + 3) / 4, (RD4 + 1) / 4, (RD4 + 2) / 4, RD4 / 4};
        if (pat[s & 3] >= 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[(2 * s) % 8]) : "v"(base), "n"((s % 8) * 1024));
        if (pat[s & 3] >= 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[(2 * s + 1) % 8]) : "v"(base), "n"((s % 8) * 1024 + 512));
      }
      if ((MODE & 2) && ((s & 3) < WC4)) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
      bf16x8 a = fa, b = fb;
      if (MODE & 4) { a = __builtin_bit_cast(bf16x8, ring[(2 * s + 4) % 8]); b = __builtin_bit_cast(bf16x8, ring[(2 * s + 5) % 8]); }
      if (s < 8) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accS) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accO[s & 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float x = (MODE & 8) ? prev[(s + k) & 15] : vs;
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x) : "v"(x), "v"(cs), "v"(nm));
        vs += x;  // plain add (counts as a second VALU)
      }
#pragma unroll
      for (int t = 0; t < T; ++t) {
        float x = (MODE & 8) ? prev[(s + t + 7) & 15] : vs;
        asm volatile("v_exp_f32 %0, %1" : "=v"(x) : "v"(x));
        vs += x;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MODE & 8) prev = accS;
  }
  long long t1 = clock64();
  float sum = vs;
  for (int r = 0; r < 16; ++r) sum += accS[r] + accO[0][r] + accO[1][r] + accO[2][r] + accO[3][r];
  for (int i = 0; i < 8; ++i) sum += (float)ring[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * WAVES + threadIdx.x / 64] = t1 - t0;
}

template <int MODE, int K, int T, int WAVES, int RD4 = 8, int WC4 = 4> void run() {
  float* out; long long* cyc; const int blocks = 256, iters = 1000;
  (void)hipMalloc(&out, blocks * WAVES * 64 * 4); (void)hipMalloc(&cyc, blocks * WAVES * 8);
  auto k = kern<MODE, K, T, WAVES, RD4, WC4>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(WAVES * 64), 100 * 1024, 0, out, cyc, iters);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(WAVES * 64), 100 * 1024, 0, out, cyc, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_mfma_ns = ms * 1e6 / (16.0 * iters * (WAVES / 4));
  printf("mode=%2d K=%d(x2 valu) T=%d rd/4slots=%d wc/4slots=%d waves/SIMD=%d: %.3f ms  -> %.1f ns per MFMA per SIMD (32 cyc = %.1f ns at 2.1 GHz)  util %.0f%%\n", MODE, K, T, RD4, WC4, WAVES / 4, ms,
         per_mfma_ns, 32 / 2.1, 100.0 * (32 / 2.1) / per_mfma_ns);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<15, 2, 1, 4>(); run<15, 2, 1, 8>();
  run<15, 2, 1, 4, 4, 4>(); run<15, 2, 1, 4, 3, 2>(); run<15, 2, 1, 4, 2, 2>(); run<15, 2, 1, 4, 3, 1>();
  run<15, 2, 1, 8, 4, 4>(); run<15, 2, 1, 8, 3, 2>();
  run<15, 1, 1, 4, 3, 2>(); run<15, 1, 1, 4, 8, 4>(); run<15, 1, 1, 8, 8, 4>(); run<15, 3, 1, 4, 3, 2>();
  return 0;
}
