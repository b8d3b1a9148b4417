// What does one LDS-DMA piece (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction) cost a wave that is otherwise feeding the matrix
// pipe, as a function of WHO issues it WHEN?  Workgroup = 4 waves (one per SIMD), one workgroup per CU, a barrier + vmcnt(0) per
// iteration (= the tile loop of fa_fwd_w64_kernel); an iteration = 64 MFMA gaps, each with FILL-dependent fillers.
//   PLACE 0: no DMA                      1: every wave, 8 pieces at gaps 1,3,..,15 (all four waves in the same gaps = the kernel today)
//   PLACE 2: wave w at gaps 16w+1,16w+3,.. (scalar branch per slot)      3: as 2, but through EXEC = 0 instead of a branch (every wave issues 32)
//   PLACE 4: wave 0 issues all 32 pieces of the workgroup, one per odd gap; waves 1-3 none
//   PLACE 5: every wave, 8 pieces back to back in gap 1                  6: every wave, gaps 4w+1 + 16k (interleaved stagger, branch)
//   FILL 0: bare MFMAs   1: + v_exp, 2 v_add per gap   2: + a ds_read_b128 every other gap with an lgkmcnt wait two gaps later
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int PLACE, int FILL>
__global__ void __launch_bounds__(256, 1) kern(const char* src, float* out, long long* cyc, int iters) {
  extern __shared__ char smem[];
  for (int i = threadIdx.x; i < 40960; i += blockDim.x) ((float*)smem)[i] = 0.001f * (i & 1023);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned rbase = lane * 16 + wave * 4096;
  const unsigned long long a = (unsigned long long)(src + (size_t)blockIdx.x * (2u << 20));
  const u32x4 srd = {(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu, 2u << 20, 0x00020000u};
  const unsigned voff = lane * 16;
  u32x4 ring[3];
  for (int i = 0; i < 3; ++i) ring[i] = u32x4{0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  f32x16 acc[4] = {};
  float vs = 0.25f, l0 = 0.f, l1 = 0.f;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)((it & 31) * 32768));
    const unsigned dstb = __builtin_amdgcn_readfirstlane((unsigned)(65536 + (it & 1) * 32768));
#pragma unroll
    for (int g = 0; g < 64; ++g) {
      if (FILL >= 2 && (g & 1) == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[(g / 2) % 3]) : "v"(rbase), "n"(((g / 2) % 16) * 1024));
      if (FILL >= 2 && (g & 1) == 0 && g >= 4) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
      bf16x8 fa = __builtin_bit_cast(bf16x8, ring[(g / 2 + 1) % 3]), fb = __builtin_bit_cast(bf16x8, ring[(g / 2 + 2) % 3]);
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(fa), "v"(fb));
      // ---- the DMA slot of this gap
      int pc = -1, own = -1;   // piece index (LDS / source offset), owner wave (-1: every wave)
      if (PLACE == 1 && (g & 1) && g < 16) pc = g / 2;
      if ((PLACE == 2 || PLACE == 3) && (g & 1)) { own = g / 16; pc = (g % 16) / 2; }
      if (PLACE == 4 && (g & 1)) { own = 0; pc = g / 2; }
      if (PLACE == 6 && (g & 1)) { own = (g / 2) % 4; pc = g / 8; }
      if (pc >= 0) {
        const unsigned dst = dstb + wave * 8192 + (pc % 8) * 1024;
        const unsigned so = soff + pc * 1024 + wave * 262144;
        if (PLACE == 3) {
          unsigned long long keep;
          asm volatile("s_mov_b64 %0, exec\n\ts_cmp_eq_u32 %5, %6\n\ts_cselect_b64 exec, %0, 0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b64 exec, %0"
                       : "=&s"(keep) : "v"(voff), "s"(dst), "s"(srd), "s"(so), "s"(wave), "n"(own < 0 ? 0 : own) : "memory", "scc");
        } else if (own < 0 || wave == own) {
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" : : "v"(voff), "s"(dst), "s"(srd), "s"(so) : "memory");
        }
      }
      if (PLACE == 5 && g == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned dst = dstb + wave * 8192 + j * 1024;
          const unsigned so = soff + j * 1024 + wave * 262144;
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" : : "v"(voff), "s"(dst), "s"(srd), "s"(so) : "memory");
        }
      }
      if (FILL >= 1) {
        float x = vs;
        asm volatile("v_exp_f32 %0, %1\n\ts_nop 0" : "=v"(x) : "v"(x));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(l0) : "v"(x));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(l1) : "v"(x));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }
  long long t1 = clock64();
  float sum = vs + l0 + l1;
  for (int r = 0; r < 16; ++r) sum += acc[0][r] + acc[1][r] + acc[2][r] + acc[3][r];
  for (int i = 0; i < 3; ++i) sum += (float)ring[i][0];
  sum += ((float*)smem)[16384 + threadIdx.x];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int PLACE, int FILL> void run(const char* src, float* out, long long* cyc) {
  const int blocks = 256, iters = 400;
  auto k = kern<PLACE, FILL>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 160 * 1024, 0, src, out, cyc, iters);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 160 * 1024, 0, src, out, cyc, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long h[1024]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double w[4] = {0, 0, 0, 0};
  for (int b = 0; b < 256; ++b) for (int x = 0; x < 4; ++x) w[x] += (double)h[b * 4 + x] / 256.0 / iters;
  printf("place=%d fill=%d: %.3f ms; clocks per iteration (64 MFMAs = 2048 at the pipe's pace) wave0..3: %.0f %.0f %.0f %.0f -> %.1f clk/MFMA; %.2f GHz\n", PLACE, FILL, ms, w[0], w[1],
         w[2], w[3], w[0] / 64, w[0] * iters / (ms * 1e6));
}
int main() {
  char* src; float* out; long long* cyc;
  (void)hipMalloc(&src, (size_t)256 * (2u << 20) + (1u << 20)); (void)hipMemset(src, 1, (size_t)256 * (2u << 20));
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 1024 * 8);
#define ROW(F) run<0, F>(src, out, cyc); run<1, F>(src, out, cyc); run<2, F>(src, out, cyc); run<3, F>(src, out, cyc); run<4, F>(src, out, cyc); run<5, F>(src, out, cyc); run<6, F>(src, out, cyc);
  ROW(0) ROW(1) ROW(2)
  return 0;
}
