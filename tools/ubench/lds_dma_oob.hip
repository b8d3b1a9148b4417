// Probe: what does `buffer_load_dwordx4 ... lds` write into LDS for lanes whose offset is outside the buffer descriptor's range?
// LDS is pre-filled with a NaN pattern; lanes 0..31 read in range, lanes 32..63 out of range (and a second pass uses a null descriptor).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_dma_oob.hip -o tools/ubench/lds_dma_oob && tools/ubench/lds_dma_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* src, unsigned nbytes, unsigned* out, unsigned soff) {
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) smem[i] = 0x7fc01234u;
  __syncthreads();
  const unsigned long long a = (unsigned long long)src;
  u32x4 srd = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, nbytes, 0x00020000u};
  srd[0] = __builtin_amdgcn_readfirstlane(srd[0]); srd[1] = __builtin_amdgcn_readfirstlane(srd[1]); srd[2] = __builtin_amdgcn_readfirstlane(srd[2]); srd[3] = __builtin_amdgcn_readfirstlane(srd[3]);
  const unsigned vo = lane * 16;  // lanes >= nbytes/16 are out of range
  const unsigned dst = 0;
  soff = __builtin_amdgcn_readfirstlane(soff);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds\n\ts_waitcnt vmcnt(0)" : : "v"(vo), "s"(dst), "s"(srd), "s"(soff) : "memory");
  __syncthreads();
  for (int i = lane; i < 256; i += 64) out[i] = smem[i];
}
int main() {
  unsigned *src, *out, h[256], hs[256];
  for (int i = 0; i < 256; ++i) hs[i] = 0x3f800000u + i;
  hipMalloc(&src, 1024); hipMalloc(&out, 1024);
  hipMemcpy(src, hs, 1024, hipMemcpyHostToDevice);
  for (unsigned cfg : {512u, 0u, 768u + (256u << 16)}) {
    const unsigned nbytes = cfg & 0xffffu, soff = cfg >> 16;   // third pass: scalar offset 256 -- is it part of the range check?
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, src, nbytes, out, soff);
    hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
    if (soff) printf("soffset=%u: ", soff);
    printf("num_records=%u: lane0 %08x lane31 %08x | lane32 %08x %08x %08x %08x lane63 %08x\n", nbytes, h[0], h[31 * 4], h[32 * 4], h[32 * 4 + 1], h[32 * 4 + 2], h[32 * 4 + 3], h[63 * 4]);
  }
  return 0;
}
