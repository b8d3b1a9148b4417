// Microbenchmark (MI355X): can MFMA and VALU work overlap on one SIMD (a) across the two waves of a SIMD,
// (b) inside one wave with k independent VALU ops between consecutive MFMAs?
// build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MODE 0: all waves MFMA only. 1: all waves VALU only. 2: waves 0-3 MFMA, waves 4-7 VALU (same SIMDs, 512 threads).
// 3: waves 0-3 MFMA only, 4-7 idle (exit).  4: waves 0-3 idle, 4-7 VALU only.
template <int MODE, int TRANS>
__global__ void __launch_bounds__(512, 2) cross_wave(float* out, int iters, long long* cyc) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool do_mfma = (MODE == 0) || ((MODE == 2 || MODE == 3) && wave < 4);
  const bool do_valu = (MODE == 1) || ((MODE == 2 || MODE == 4) && wave >= 4);
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x & 7); b[j] = (__bf16)1.0f; }
  f32x16 acc[4] = {};
  float v[16];
  for (int j = 0; j < 16; ++j) v[j] = 0.001f * (threadIdx.x + j);
  long long t0 = __builtin_readcyclecounter();
  if (do_mfma) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
    }
  }
  if (do_valu) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          if (TRANS) v[j] = __builtin_amdgcn_exp2f(v[j]);
          else v[j] = __builtin_fmaf(v[j], 1.0001f, 0.5f);
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

// one wave per SIMD (256 threads, launch bounds keep 1 block/CU via LDS): per iteration 8 MFMAs, each followed by K VALU ops
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
      acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const int idx = (u * K + j) & 15;
        if (TRANS) v[idx] = __builtin_amdgcn_exp2f(v[idx]);
        else v[idx] = __builtin_fmaf(v[idx], 1.0001f, 0.5f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) s += acc[u][r];
  for (int j = 0; j < 16; ++j) s += v[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename F> float time_ms(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  float* out; long long* cyc; CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 64));
  const int iters = 20000;
  long long h[8];
#define RUN_CW(M, T) { float ms = time_ms([&] { cross_wave<M, T><<<256, 512>>>(out, iters, cyc); }); CK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost)); \
    printf("cross_wave mode %d trans %d: %.3f ms  cycles wave0 %lld wave4 %lld  (per iter %.1f / %.1f)\n", M, T, ms, h[0], h[4], (double)h[0] / iters, (double)h[4] / iters); }
  printf("per iter: 8 MFMA 32x32x16 (=256 pipe cycles) for MFMA waves, 32 VALU ops for VALU waves; s_memtime-like counter units\n");
  RUN_CW(0, 0) RUN_CW(3, 0) RUN_CW(1, 0) RUN_CW(4, 0) RUN_CW(2, 0) RUN_CW(1, 1) RUN_CW(4, 1) RUN_CW(2, 1)
#define RUN_IW(K, T) { float ms = time_ms([&] { in_wave<K, T><<<256, 256>>>(out, iters, cyc); }); CK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost)); \
    printf("in_wave K=%d trans %d: %.3f ms  per MFMA %.1f counter units\n", K, T, ms, (double)h[0] / iters / 8); }
  RUN_IW(0, 0) RUN_IW(1, 0) RUN_IW(2, 0) RUN_IW(3, 0) RUN_IW(4, 0) RUN_IW(5, 0) RUN_IW(6, 0) RUN_IW(7, 0) RUN_IW(8, 0) RUN_IW(10, 0) RUN_IW(12, 0)
  RUN_IW(1, 1) RUN_IW(2, 1) RUN_IW(3, 1) RUN_IW(4, 1) RUN_IW(6, 1) RUN_IW(8, 1)
  return 0;
}
