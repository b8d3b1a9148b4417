// Microbenchmark 2 (MI355X): MFMA / VALU overlap with explicit priorities and hand-interleaved streams (inline asm).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// 512 threads = 2 waves per SIMD.  MFMA_FIRST: waves 0-3 do MFMA and 4-7 VALU, else the reverse.
// PRIO: 0 none, 1 = VALU waves s_setprio 3, 2 = MFMA waves s_setprio 3.
template <bool MFMA_FIRST, int PRIO, int TRANS>
__global__ void __launch_bounds__(512, 2) cross_wave(float* out, int iters, long long* cyc) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool do_mfma = MFMA_FIRST ? (wave < 4) : (wave >= 4);
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x & 7); b[j] = (__bf16)1.0f; }
  f32x16 acc[4] = {};
  float v[16];
  for (int j = 0; j < 16; ++j) v[j] = 0.001f * (threadIdx.x + j);
  if (PRIO == 1 && !do_mfma) __builtin_amdgcn_s_setprio(3);
  if (PRIO == 2 && do_mfma) __builtin_amdgcn_s_setprio(3);
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  if (do_mfma) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
    }
  } else {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          if (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
          else asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(v[j]));
        }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
  for (int j = 0; j < 16; ++j) s += v[j];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

// one wave per SIMD; per MFMA exactly K VALU ops placed right behind it by inline asm (independent registers)
template <int K, int TRANS>
__global__ void __launch_bounds__(256) in_wave(float* out, int iters, long long* cyc) {
  __shared__ char pad[100 * 1024];
  if (threadIdx.x == 9999) pad[threadIdx.x] = 1;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x & 7); b[j] = (__bf16)1.0f; }
  f32x16 acc[4] = {};
  float v[16];
  for (int j = 0; j < 16; ++j) v[j] = 0.001f * (threadIdx.x + j);
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const int idx = (u * K + j) & 15;
        if (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(v[idx]));
        else asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(v[idx]));
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
  for (int j = 0; j < 16; ++j) s += v[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// two waves per SIMD, BOTH running the interleaved stream (MFMA + K VALU)
template <int K, int TRANS>
__global__ void __launch_bounds__(512, 2) in_wave2(float* out, int iters, long long* cyc) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x & 7); b[j] = (__bf16)1.0f; }
  f32x16 acc[4] = {};
  float v[16];
  for (int j = 0; j < 16; ++j) v[j] = 0.001f * (threadIdx.x + j);
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const int idx = (u * K + j) & 15;
        if (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(v[idx]));
        else asm volatile("v_fma_f32 %0, %0, 1.0, 0.5" : "+v"(v[idx]));
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
  for (int j = 0; j < 16; ++j) s += v[j];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <typename F> float time_ms(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  float* out; long long* cyc; CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 64));
  const int iters = 20000;
  long long h[8];
#define RUN_CW(F, P, T) { float ms = time_ms([&] { cross_wave<F, P, T><<<256, 512>>>(out, iters, cyc); }); CK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost)); \
    printf("cross_wave mfma_first=%d prio=%d trans=%d: %.3f ms  per-iter wave0 %.1f wave4 %.1f\n", (int)F, P, T, ms, (double)h[0] / iters, (double)h[4] / iters); }
  printf("cross_wave: MFMA waves 8 MFMA/iter (256 pipe cycles), VALU waves 32 ops/iter; ideal overlap = max, none = sum\n");
  RUN_CW(true, 0, 0) RUN_CW(true, 1, 0) RUN_CW(true, 2, 0) RUN_CW(false, 0, 0) RUN_CW(false, 1, 0) RUN_CW(false, 2, 0)
  RUN_CW(true, 0, 1) RUN_CW(true, 1, 1) RUN_CW(false, 0, 1) RUN_CW(false, 1, 1)
#define RUN_IW(K, T) { float ms = time_ms([&] { in_wave<K, T><<<256, 256>>>(out, iters, cyc); }); CK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost)); \
    printf("in_wave  (1 wave/SIMD) K=%2d trans=%d: %.3f ms  per MFMA %.1f\n", K, T, ms, (double)h[0] / iters / 8); }
  RUN_IW(0, 0) RUN_IW(1, 0) RUN_IW(2, 0) RUN_IW(3, 0) RUN_IW(4, 0) RUN_IW(5, 0) RUN_IW(6, 0) RUN_IW(7, 0) RUN_IW(8, 0) RUN_IW(10, 0) RUN_IW(12, 0) RUN_IW(16, 0)
  RUN_IW(2, 1) RUN_IW(4, 1) RUN_IW(6, 1) RUN_IW(8, 1)
#define RUN_IW2(K, T) { float ms = time_ms([&] { in_wave2<K, T><<<256, 512>>>(out, iters, cyc); }); CK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost)); \
    printf("in_wave2 (2 waves/SIMD) K=%2d trans=%d: %.3f ms  per MFMA wave0 %.1f wave4 %.1f\n", K, T, ms, (double)h[0] / iters / 8, (double)h[4] / iters / 8); }
  RUN_IW2(0, 0) RUN_IW2(2, 0) RUN_IW2(4, 0) RUN_IW2(5, 0) RUN_IW2(6, 0) RUN_IW2(8, 0) RUN_IW2(4, 1)
  return 0;
}
