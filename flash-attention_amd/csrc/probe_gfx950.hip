// Hardware-semantics probe for gfx950 (MI355X): verifies, on the device, every lane-layout
// assumption the attention kernels rely on.  Standalone: hipcc --offload-arch=gfx950 probe_gfx950.hip -o probe
// Prints PASS/FAIL per assumption and dumps raw lane data on failure.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2);} } while (0)

// A: 32x16 row-major bf16, B: 16x32 row-major bf16 -> D 32x32 f32 using the ASSUMED layouts:
//  A operand: lane l holds A[l&31][8*(l>>5) + j], j=0..7
//  B operand: lane l holds B[8*(l>>5) + j][l&31]
//  D: reg r of lane l = D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31]
__global__ void mfma32_bf16(const __bf16* A, const __bf16* B, float* D) {
  int l = threadIdx.x, hi = l >> 5, c = l & 31;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[c * 16 + 8 * hi + j]; b[j] = B[(8 * hi + j) * 32 + c]; }
  f32x16 acc = {};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + c] = acc[r];
}
__global__ void mfma32_f16(const _Float16* A, const _Float16* B, float* D) {
  int l = threadIdx.x, hi = l >> 5, c = l & 31;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[c * 16 + 8 * hi + j]; b[j] = B[(8 * hi + j) * 32 + c]; }
  f32x16 acc = {};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + c] = acc[r];
}
// 16x16x32: A 16x32, B 32x16; lane l: A[l&15][8*(l>>4)+j], B[8*(l>>4)+j][l&15]; D reg r = D[4*(l>>4)+r][l&15]
__global__ void mfma16_bf16(const __bf16* A, const __bf16* B, float* D) {
  int l = threadIdx.x, g = l >> 4, c = l & 15;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[c * 32 + 8 * g + j]; b[j] = B[(8 * g + j) * 16 + c]; }
  f32x4 acc = {};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + c] = acc[r];
}

// ds_read_b64_tr_b16: LDS holds M[r][c] = r*64 + c... we use a [8 rows][stride 72] short image with value = row*100+col.
// ASSUMPTION: within each 16-lane group, lane i supplies the address of (row i>>2, cols 4*(i&3)..+3) of a 4x16 block;
// the result in lane i, element j is M[j][i] of that block.
__global__ void tr_probe(short* out_hyp, short* out_raw) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 72];
  int l = threadIdx.x;
  for (int i = l; i < 64 * 72; i += 64) lds[i] = (short)((i / 72) * 100 + (i % 72));
  __syncthreads();
  int g = l >> 4, i = l & 15;
  // hypothesis addressing: group g reads the 4x16 block at rows 4g..4g+3, cols 16*(g&1)..+15
  int row = 4 * g + (i >> 2), col = 16 * (g & 1) + 4 * (i & 3);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + row * 72 + col));
  for (int j = 0; j < 4; ++j) out_hyp[l * 4 + j] = v[j];
  // raw addressing: lane l reads 4 shorts at row l (value l*100 + 0..3) -> tells who gets what
  s16x4 w = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + l * 72));
  for (int j = 0; j < 4; ++j) out_raw[l * 4 + j] = w[j];
}

__global__ void permlane_probe(unsigned* out) {
  unsigned l = threadIdx.x;
  unsigned a = 1000 + l, b = 2000 + l;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[l] = r[0]; out[64 + l] = r[1];
  out[128 + l] = __shfl_xor((int)a, 32);
}

// bf16 conversion rounding: (__bf16)float must be round-to-nearest-even
__global__ void cvt_probe(const float* in, unsigned short* out, int n) {
  int t = threadIdx.x;
  if (t < n) { __bf16 x = (__bf16)in[t]; out[t] = *(unsigned short*)&x; }
}

static float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short s; memcpy(&s, &h, 2); return s; }

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s arch %s CUs %d clock %d kHz LDS/block %zu regs/block %d L2 %d\n", prop.name, prop.gcnArchName,
         prop.multiProcessorCount, prop.clockRate, prop.sharedMemPerBlock, prop.regsPerBlock, prop.l2CacheSize);
  int fails = 0;
  srand(1);
  {  // mfma 32x32x16
    std::vector<float> A(32 * 16), B(16 * 32), Dref(32 * 32, 0.f);
    for (auto& x : A) x = (float)(rand() % 17 - 8);
    for (auto& x : B) x = (float)(rand() % 13 - 6);
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int k = 0; k < 16; ++k) s += A[m * 16 + k] * B[k * 32 + n]; Dref[m * 32 + n] = s; }
    std::vector<unsigned short> Ab(32 * 16), Bb(16 * 32), Ah(32 * 16), Bh(16 * 32);
    for (int i = 0; i < 512; ++i) { Ab[i] = f2bf(A[i]); Bb[i] = f2bf(B[i]); Ah[i] = f2h(A[i]); Bh[i] = f2h(B[i]); }
    void *dA, *dB; float* dD; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 4096));
    std::vector<float> D(1024);
    CK(hipMemcpy(dA, Ab.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bb.data(), 1024, hipMemcpyHostToDevice));
    mfma32_bf16<<<1, 64>>>((const __bf16*)dA, (const __bf16*)dB, dD); CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += (D[i] != Dref[i]);
    printf("[mfma_f32_32x32x16_bf16 layout] %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
    CK(hipMemcpy(dA, Ah.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bh.data(), 1024, hipMemcpyHostToDevice));
    mfma32_f16<<<1, 64>>>((const _Float16*)dA, (const _Float16*)dB, dD); CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
    bad = 0; for (int i = 0; i < 1024; ++i) bad += (D[i] != Dref[i]);
    printf("[mfma_f32_32x32x16_f16 layout] %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  {  // mfma 16x16x32
    std::vector<float> A(16 * 32), B(32 * 16), Dref(256, 0.f);
    for (auto& x : A) x = (float)(rand() % 17 - 8);
    for (auto& x : B) x = (float)(rand() % 13 - 6);
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float s = 0; for (int k = 0; k < 32; ++k) s += A[m * 32 + k] * B[k * 16 + n]; Dref[m * 16 + n] = s; }
    std::vector<unsigned short> Ab(512), Bb(512);
    for (int i = 0; i < 512; ++i) { Ab[i] = f2bf(A[i]); Bb[i] = f2bf(B[i]); }
    void *dA, *dB; float* dD; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 1024));
    CK(hipMemcpy(dA, Ab.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bb.data(), 1024, hipMemcpyHostToDevice));
    mfma16_bf16<<<1, 64>>>((const __bf16*)dA, (const __bf16*)dB, dD); CK(hipDeviceSynchronize());
    std::vector<float> D(256); CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 256; ++i) bad += (D[i] != Dref[i]);
    printf("[mfma_f32_16x16x32_bf16 layout] %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  {  // transpose read
    short *dh, *dr; CK(hipMalloc(&dh, 512)); CK(hipMalloc(&dr, 512));
    tr_probe<<<1, 64>>>(dh, dr); CK(hipDeviceSynchronize());
    std::vector<short> H(256), R(256);
    CK(hipMemcpy(H.data(), dh, 512, hipMemcpyDeviceToHost)); CK(hipMemcpy(R.data(), dr, 512, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
      int g = l >> 4, i = l & 15;
      int exp = (4 * g + j) * 100 + 16 * (g & 1) + i;
      bad += (H[l * 4 + j] != exp);
    }
    printf("[ds_read_b64_tr_b16 semantics] %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
    printf("  raw tr dump (lane: 4 values; source value = srclane*100 + elem):\n");
    for (int l = 0; l < 64; ++l) { printf("  %2d: %5d %5d %5d %5d%s", l, R[l * 4], R[l * 4 + 1], R[l * 4 + 2], R[l * 4 + 3], (l & 3) == 3 ? "\n" : " |"); }
  }
  {  // permlane32_swap
    unsigned* d; CK(hipMalloc(&d, 192 * 4)); permlane_probe<<<1, 64>>>(d); CK(hipDeviceSynchronize());
    std::vector<unsigned> P(192); CK(hipMemcpy(P.data(), d, 768, hipMemcpyDeviceToHost));
    // assumption (guide T21): r[0] (vdst=a): lanes 0-31 keep a, lanes 32-63 get b from lane-32; r[1] (src=b): lanes 0-31 get a from lane+32, lanes 32-63 keep b
    int bad = 0;
    for (unsigned l = 0; l < 64; ++l) {
      unsigned e0 = l < 32 ? 1000 + l : 2000 + (l - 32);
      unsigned e1 = l < 32 ? 1000 + (l + 32) : 2000 + l;
      bad += (P[l] != e0) + (P[64 + l] != e1);
      bad += (P[128 + l] != 1000 + (l ^ 32));
    }
    printf("[permlane32_swap / shfl_xor 32 semantics] %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
    if (bad) { for (int l = 0; l < 64; ++l) printf("  %2d: r0=%u r1=%u shfl=%u\n", l, P[l], P[64 + l], P[128 + l]); }
  }
  {  // bf16 RNE
    const int n = 8; float in[n] = {1.0f, 1.00390625f, 1.01171875f, 3.1415926f, -2.7182817f, 65504.f, 1e-8f, 0.3333333f};
    float* di; unsigned short* dout; CK(hipMalloc(&di, 64)); CK(hipMalloc(&dout, 32));
    CK(hipMemcpy(di, in, n * 4, hipMemcpyHostToDevice)); cvt_probe<<<1, 64>>>(di, dout, n); CK(hipDeviceSynchronize());
    unsigned short o[n]; CK(hipMemcpy(o, dout, n * 2, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < n; ++i) bad += (o[i] != f2bf(in[i]));
    printf("[float->bf16 is RNE] %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  printf("PROBE %s\n", fails ? "FAILED" : "ALL PASS");
  return fails ? 1 : 0;
}
