// Shared device-side helpers for the gfx950 attention kernels (wave64, MFMA 32x32x16, LDS tiles).
//
// Lane-layout facts used everywhere (verified on hardware by probe_gfx950.hip):
//   v_mfma_f32_32x32x16_{bf16,f16}:  D[m][n] = sum_k A[m][k] * B[k][n]
//     A operand : lane l holds A[l&31][8*(l>>5) + j],  j = 0..7   (8 x 16-bit = 4 VGPRs)
//     B operand : lane l holds B[8*(l>>5) + j][l&31]
//     C/D       : reg r of lane l = D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31],  r = 0..15
//   ds_read_b64_tr_b16: inside each 16-lane group, lane i supplies the address of row (i>>2),
//     columns 4*(i&3)..+3 of a 4x16 block of 16-bit elements; lane i receives column i
//     (4 elements, rows 0..3) of that block.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define FA_DEVINL __device__ __forceinline__
#define FA_LDS __attribute__((address_space(3)))

template <typename E> struct ElemTraits;
template <> struct ElemTraits<__bf16> {
  using v8 = bf16x8;
  using v4 = bf16x4;
  static FA_DEVINL f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct ElemTraits<_Float16> {
  using v8 = f16x8;
  using v4 = f16x4;
  static FA_DEVINL f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// Row index inside a 32x32 accumulator block held by register r of a lane in half hi (= lane>>5).
FA_DEVINL constexpr int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

FA_DEVINL float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// value held by lane (l ^ 32)
FA_DEVINL float xchg_half(float x) { return __shfl_xor(x, 32); }

// max / sum of a value with the one held by lane (l ^ 32), via v_permlane32_swap (no LDS round trip):
// with vdst = src = x the swap returns {r0, r1} where lanes 0-31 see (own, other) and lanes 32-63
// see (other, own) -- verified by probe_gfx950.hip -- so any symmetric combine needs no select.
FA_DEVINL float half_max(float x) {
  const unsigned u = __builtin_bit_cast(unsigned, x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
FA_DEVINL float half_sum(float x) {
  const unsigned u = __builtin_bit_cast(unsigned, x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

// acc + a.x * b.x + a.y * b.y on two packed 16-bit pairs (v_dot2_f32_bf16 / v_dot2_f32_f16: products exact in fp32)
template <typename E> FA_DEVINL float dot2_acc(unsigned a, unsigned b, float acc) {
  float r;
  if constexpr (std::is_same<E, __bf16>::value) asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(acc));
  else asm("v_dot2_f32_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(acc));
  return r;
}

// 16-byte global load of 8 consecutive 16-bit elements; zeros when !valid.
FA_DEVINL u32x4 ld_global_16B(const void* p, bool valid) {
  u32x4 z = {0u, 0u, 0u, 0u};
  if (valid) z = *reinterpret_cast<const u32x4*>(p);
  return z;
}

// 16-byte global store written through the XCD's L2 (sc1): for data another workgroup of the same launch reads back behind an agent-scope counter
// (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 payload -> s_waitcnt vmcnt(0) -> flag).  Invisible to hipcc's waitcnt pass, like lds_dma_16B.
FA_DEVINL void st_global_16B_sc1(void* p, u32x4 x) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory"); }

template <typename V> FA_DEVINL V bitcast_u32x4(u32x4 x) { return __builtin_bit_cast(V, x); }

// LDS transpose read: returns the 4 x 16-bit column this lane owns (see header comment).
FA_DEVINL s16x4 lds_read_tr16(const char FA_LDS* addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 FA_LDS*)addr);
}

template <typename V8> FA_DEVINL V8 combine_tr(s16x4 lo, s16x4 hi) {
  s16x8 x = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(V8, x);
}

// LDS-DMA: 16 bytes per lane, global -> LDS, destination = wave-uniform LDS byte address + 16 * lane.
// Issued from inline asm on purpose: hipcc's waitcnt pass treats an in-flight __builtin_amdgcn_global_load_lds
// as aliasing every later LDS read and drains it (s_waitcnt vmcnt(0)) at the first ds_read, which serialises
// the prefetch with the compute it was meant to overlap.  The asm form is invisible to that pass, so the
// kernel owns the wait: call lds_dma_wait_all() before the barrier that publishes the tile.
// M0 (DMA destination base) is saved and restored inside the statement (cdna_hip_programming.md 5.7).
FA_DEVINL void lds_dma_16B(const void* gsrc, const char FA_LDS* lds_dst_uniform) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)lds_dst_uniform);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}
FA_DEVINL void lds_dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// XCD-aware work mapping for 1-D grids.  Workgroup `bid` is observed to run on XCD bid % 8 (performance only,
// never correctness).  Work items are numbered unit * unit_size + item; a *unit* (e.g. all query blocks of the
// query heads sharing one KV head) must stay on one XCD so its K/V stay in that XCD's L2.  Units are dealt so
// that every XCD sees every sequence (packed batches have very uneven sequences):
//   hpx > 0 : units are (batch, kv-head) pairs with kv-heads-per-batch = 8 * hpx; XCD x owns kv heads
//             [x*hpx, (x+1)*hpx) of EVERY batch entry (adjacent heads stay together: their rows are adjacent in
//             HBM, which measurably helps the fabric);
//   hpx = 0 : units dealt round-robin (unit u -> XCD u % 8).
// Returns the work index or -1 for the padding workgroups of a partially filled last round.
FA_DEVINL int xcd_interleave(int bid, int n_units, int unit_size, int hpx) {
  constexpr int NX = 8;
  const int xcd = bid % NX, slot = bid / NX;
  const int j = slot / unit_size;
  const int item = slot - j * unit_size;
  int unit;
  if (hpx > 0) {
    const int bb = j / hpx;
    unit = bb * (NX * hpx) + xcd * hpx + (j - bb * hpx);
  } else {
    unit = xcd + NX * j;
  }
  return unit < n_units ? unit * unit_size + item : -1;
}

// Epilogue store of one wave's 32 x D tile held in the transposed accumulator layout (lane & 31 = row, registers = 4-element
// groups of the row at columns 32*db + 8*g + 4*hi): written directly, each store instruction puts 16 bytes into each of 32
// rows (512 partial-line writes per wave).  Staged through `stage` (32 padded rows of LDS private to the wave) every
// instruction writes whole rows.  Rows >= rows_valid are not stored.  DV < D (trimmed head dims): the tile has DV / 32
// blocks and only the first DV columns of a (pitch D) staging row are written out.
template <typename E, int D, int DV = D>
FA_DEVINL void store_tile_via_lds(char FA_LDS* stage, const f32x16 (&acc)[DV / 32], float scale, E* gtile, int64_t row_stride,
                                  int rows_valid, int lane, int chunks_valid = DV / 8) {
  using V4 = typename ElemTraits<E>::v4;
  constexpr int ROW_BYTES = D * 2, RS = ROW_BYTES + 16, CPR = D / 8, RPI = 64 / CPR;
  const int qi = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int db = 0; db < DV / 32; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      V4 ov;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) ov[jj] = (E)(acc[db][4 * g + jj] * scale);
      *reinterpret_cast<V4 FA_LDS*>(stage + qi * RS + (32 * db + 8 * g + 4 * hi) * 2) = ov;
    }
#pragma unroll
  for (int i = 0; i < 32 / RPI; ++i) {
    const int row = i * RPI + lane / CPR, ch = lane % CPR;
    const u32x4 x = *reinterpret_cast<const u32x4 FA_LDS*>(stage + row * RS + ch * 16);
    if (row < rows_valid && ch < chunks_valid) *reinterpret_cast<u32x4*>(gtile + (int64_t)row * row_stride + ch * 8) = x;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staging rows may be rewritten by the caller's next tile
}

// The same for head-packed rows (fa_fwd_kernel, FwdK::pack_g): tile row r is packed row row0 + r = query (row0 + r) / g of head
// (row0 + r) % g in the group, stored at gbase + query * row_stride + head * head_stride; rows >= rows_total do not exist.
template <typename E, int D, int DV = D>
FA_DEVINL void store_tile_via_lds_packed(char FA_LDS* stage, const f32x16 (&acc)[DV / 32], float scale, E* gbase, int64_t row_stride,
                                         int64_t head_stride, int g, int row0, int rows_total, int lane, int chunks_valid = DV / 8) {
  using V4 = typename ElemTraits<E>::v4;
  constexpr int ROW_BYTES = D * 2, RS = ROW_BYTES + 16, CPR = D / 8, RPI = 64 / CPR;
  const int qi = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int db = 0; db < DV / 32; ++db)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      V4 ov;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) ov[jj] = (E)(acc[db][4 * gq + jj] * scale);
      *reinterpret_cast<V4 FA_LDS*>(stage + qi * RS + (32 * db + 8 * gq + 4 * hi) * 2) = ov;
    }
#pragma unroll
  for (int i = 0; i < 32 / RPI; ++i) {
    const int row = i * RPI + lane / CPR, ch = lane % CPR;
    const u32x4 x = *reinterpret_cast<const u32x4 FA_LDS*>(stage + row * RS + ch * 16);
    const int prow = row0 + row, iq = prow / g, hh = prow - iq * g;
    if (prow < rows_total && ch < chunks_valid) *reinterpret_cast<u32x4*>(gbase + (int64_t)iq * row_stride + (int64_t)hh * head_stride + ch * 8) = x;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// Unified LDS tile layout (rows of D 16-bit elements, 16-B chunks XOR-swizzled) that is
// conflict-free for both access patterns used on the same tile:
//   - ds_read_b128 operand rows (16 distinct rows per lane group, same logical chunk),
//   - ds_read_b64_tr_b16 (a half-wave reads 4 consecutive rows x 64 contiguous logical bytes).
template <int D> FA_DEVINL int swz16(int row) {
  return D >= 128 ? (((row & 3) << 2) | ((row >> 2) & 3)) : ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
}
template <int D> FA_DEVINL int tile_off(int row, int chunk) { return row * (D * 2) + ((chunk ^ swz16<D>(row)) << 4); }

// ---- dS spill (BwdK::ds_ws) ------------------------------------------------------------------------------------------
// Does the dK/dV kernel compute (and write) the dS sub-tile of queries [q0, q0+32) x keys [k0, k0+32)?  The dQ contraction
// reads exactly the sub-tiles for which this holds -- anything else in the workspace is uninitialised.
FA_DEVINL bool ds_tile_active(int q0, int k0, int sq, int sk, int shift, int wl, int wr) {
  if (q0 >= sq || k0 >= sk) return false;
  const int k1 = min(k0 + 31, sk - 1);
  if (wr >= 0 && k0 > q0 + 31 + shift + wr) return false;
  if (wl >= 0 && k1 < q0 + shift - wl) return false;
  return true;
}
// Image of a sub-tile: two 1-KiB halves (queries 0-15 / 16-31), each the dK/dV kernel's B-operand fragment (lane = key,
// lane half hi, 8 queries acc_row(j, hi)) stored one 16-byte slot per lane.  The slot order is chosen for the reader, which
// pulls the transposed fragment (lane = query) out of LDS with ds_read_b64_tr_b16: the 16 lanes of one transpose group vary
// key bits 0-1, hi and the 8-byte half of the slot -- with those in address bits 3-6 the group covers all 32 banks once.
FA_DEVINL int ds_slot(int key, int hi) { return (key & 3) | (hi << 2) | ((key >> 2) << 3); }
// Packed rows of the 5-contraction backward's dS workspace (fa_bwd_dkdv_w64.hip, DS).  The unit of a row is the PAIR of key sub-tiles (64 keys) a dK/dV wave
// writes together: row block i (32 queries) of a head holds pairs 0 .. pc(i) - 1, pc(i) = min((i + a) >> 1, np64) -- a - 1 = the key sub-tiles row block 0 can see
// under a right-bounded mask (sk >= sq, no left window), so (i + a - 2) >> 1 is the pair of the last sub-tile row block i sees; a = 2 * np64 without a right bound --
// and starts at sub-tile ds_row_start(i) = 2 * sum_{j < i} pc(j); with F(n) = sum_{m < n} (m >> 1) = ((n - 1)^2) >> 2 that is F(a + i) - F(a) up to row block
// jb = max(0, 2 * np64 - a), the first one that holds every pair.  A head occupies ds_row_start(nq32) sub-tiles of 2 KB.
__host__ __device__ inline int ds_row_start(int i, int a, int jb, int np64) {
  const int m = i < jb ? i : jb;
  const int n1 = a + m - 1, n0 = a - 1;
  return 2 * (((n1 * n1) >> 2) - ((n0 * n0) >> 2) + (i > jb ? i - jb : 0) * np64);
}

// Score-transform features of the non-plain kernel variants (template int FEAT): softcap, ALiBi, dropout.  A variant
// with exactly one feature carries only that feature's code and registers; FEAT_ALL checks the parameters at run time.
enum { FEAT_NONE = 0, FEAT_CAP = 1, FEAT_ALIBI = 2, FEAT_DROP = 4, FEAT_ALL = 7,
       FEAT_EXACT = 8 };   // dK/dV kernel only: no feature, and no pre-scaled K either (fa_bwd.hip: PRE) -- the default plain variant since round 4
inline int feat_code(bool cap, bool alibi, bool drop) {
  const int f = (cap ? FEAT_CAP : 0) | (alibi ? FEAT_ALIBI : 0) | (drop ? FEAT_DROP : 0);
  return (f == (FEAT_CAP | FEAT_ALIBI)) ? FEAT_ALL : f;  // every combination but softcap+ALiBi(+dropout) has its own variant
}
// tanh(x) = 1 - 2 / (2^(2x log2 e) + 1): two transcendentals instead of libm's polynomial; relative error ~1e-4 near 0,
// exact limits at +-inf (softcap: reference utils.h:395-409 uses the hardware tanh approximation as well)
FA_DEVINL float fast_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);
  return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}

// ---- varlen work list ---------------------------------------------------------------------------------
// A packed batch with uneven lengths leaves most (batch, block) slots of a max_seqlen-sized grid empty.  The
// schedule kernel writes the non-empty query (or key) blocks, heaviest first, as {batch, block} pairs after a
// header {count, 0}; workgroup `bid` then takes (head, item): KV heads are dealt to the XCDs (all blocks and all query
// heads of one KV head share an L2) when there are >= 8 of them, heads round-robin otherwise.
FA_DEVINL bool work_list_item(const int2* __restrict__ list, int bid, int h, int h_k, int& b, int& head, int& blk) {
  constexpr int NX = 8;
  const int count = list[0].x;
  int item;
  if (h_k % NX == 0) {
    const int ratio = h / h_k, per = h_k / NX;
    const int x = bid % NX, r = bid / NX;
    const int g = r % ratio, hl = (r / ratio) % per;
    item = r / (ratio * per);
    head = (x * per + hl) * ratio + g;
  } else {
    head = bid % h;
    item = bid / h;
  }
  if (item >= count) return false;
  const int2 e = list[1 + item];
  b = e.x;
  blk = e.y;
  return true;
}

// ---- dropout random stream --------------------------------------------------------------------------
// Counter-based and keyed: the 4 random bytes of keys 4g .. 4g+3 of (batch b, query head h, query row i) are word 0 of
// Philox2x32-7 with counter (g, i) under the 32-bit key K(seed, offset) ^ (b*H + h) * odd.  For a fixed key the block
// cipher is a bijection of the 64-bit counter, and the key is injective in (b*H + h), so no (batch, head) stream can be a
// shifted copy of another one (the first version -- one multiply/xorshift hash over an ADDITIVE 32-bit counter, streams
// distinguished only by a start offset -- overlapped by construction once B*H*Sq*Sk/4 approached 2^32).  A pure function of
// the element, so the forward and both backward kernels regenerate the same mask in their own tilings; the reference does
// the same with Philox4x32-7 keyed on its tile coordinates (csrc/flash_attn/src/dropout.h:31-90, philox.cuh).
// Cost: 7 x (v_mul_hi, v_mul_lo, 3-input xor) = 21 VALU per 4 elements.
FA_DEVINL uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
FA_DEVINL uint32_t drop_bh_key(const uint64_t* rng, int bh) {
  const uint64_t seed = rng[0], off = rng[1];
  uint32_t k = hash32((uint32_t)seed ^ 0x9E3779B9u);
  k = hash32(k ^ (uint32_t)(seed >> 32));
  k = hash32(k ^ (uint32_t)off);
  k = hash32(k ^ (uint32_t)(off >> 32));
  return k ^ ((uint32_t)bh * 0x9E3779B1u);   // odd multiplier: distinct (batch, head) -> distinct key
}
// bytes of keys 4g..4g+3 (byte c <-> key 4g+c) of query row i
FA_DEVINL uint32_t drop_bytes(uint32_t bh_key, int i, int g) {
  constexpr uint32_t M = 0xD256D193u, W = 0x9E3779B9u;   // Philox2x32 multiplier / Weyl key increment (Random123)
  uint32_t c0 = (uint32_t)g, c1 = (uint32_t)i, key = bh_key;
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const uint32_t hi = __umulhi(M, c0), lo = M * c0;
    c0 = hi ^ key ^ c1;
    c1 = lo;
    key += W;
  }
  return c0;
}
// value held by lane j of the caller's quad (4 adjacent lanes); j must fold to a constant
template <int J> FA_DEVINL uint32_t quad_bcast_c(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, J * 0x55, 0xf, 0xf, true); }
FA_DEVINL uint32_t quad_bcast(uint32_t x, int j) {
  return j == 0 ? quad_bcast_c<0>(x) : j == 1 ? quad_bcast_c<1>(x) : j == 2 ? quad_bcast_c<2>(x) : quad_bcast_c<3>(x);
}

}  // namespace fa
