// Forward attention for gfx950, "64 query rows per wave, one wave per SIMD" schedule.
//
// Why this shape (profiles/r01_fwd_issue_analysis.txt, MI355X_MICROARCH.md "Per-instruction cycle constants"): beside one
// v_mfma_f32_32x32x16 (32 cycles on its SIMD) the issue port hides about five other instructions.  The 32-rows-per-wave
// kernels (fa_fwd_il.hip) need ~11 per MFMA: every K fragment and every transposed V fragment read from LDS feeds ONE
// MFMA.  Here a wave owns two 32-row query blocks, so each LDS fragment feeds TWO MFMAs, and three more per-element VALU
// ops are removed at the source:
//   * Q is multiplied by softmax_scale*log2(e) once per block (rounded once to the input dtype) -- no per-score multiply;
//   * the running row maximum is subtracted BY THE MATRIX PIPE: the first MFMA of a score chain takes C = -m (a 16-register
//     broadcast that only changes when the deferred rescale fires) instead of C = 0, so scores leave the pipe as s - m;
//   * accumulators never move: O (128 regs) and the Q fragments (64 regs) live in the accumulator half of the 512-entry
//     register file as MFMA C/D and B operands, the scores come out of the pipe in arch VGPRs where the VALU reads them.
// Left per score: v_exp, one add (row sum), half a v_cvt_pk, half a v_max3  => ~3.4 VALU + 0.75 LDS reads per MFMA.
// The MFMAs are inline asm (register classes as above); the steady-state step is a hand-placed sequence of MFMA "gaps"
// pinned with sched_barrier, each gap carrying its share of the VALU / LDS work.  Software pipeline as in fa_fwd_il.hip:
// step i = 16 MFMAs S_{i+1} = K_{i+1}.Q^T, 16 MFMAs O += V_{i-1}^T.P_{i-1}, VALU P_i = exp2(S_i); lagged O rescale.
//
// Workgroup = 4 waves x 64 rows = 256 query rows; registers allow one workgroup per CU, so LDS is spent freely:
// K/V double buffers (64 KB) + the staged Q block (64 KB).  K/V tiles arrive by LDS-DMA through a buffer descriptor
// (buffer_load_dwordx4 ... lds: one M0 write per four 1-KiB pieces, immediate offsets walk both LDS and memory).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "fa_device.h"
#include "fa_kernel_params.h"
#include "fa_launch.h"
#include "fa_fwd_w64_regs.h"
#define FA_W64_CLOB FA_W64_ACC_CLOBBERS
#include "fa_w64_asm.h"

#ifndef FA_W64_ABL
#define FA_W64_ABL 0  // timing ablations of the steady-state step, bit mask (results become wrong): tools/ablate_w64.sh
#endif                // 1 no exp2, 2 no row-sum adds, 4 no packing, 8 no row-max tree, 16 LDS operand reads only once per step,
                      // 32 no K/V DMA after the first tiles, 64 row-sum adds lag their exp2 by one gap, 128 no LDS operand reads at all,
                      // 256 LSE output = shader clocks per MFMA of the wave's tile loop, 512 no DMA wait / barrier per tile,
                      // 1024 no decision (row-max finish + branch), 2048 LSE output = this wave's clock stamps (lane i = stamp i: 0 prologue
                      // barrier passed, 1 Q converted, 2 K_0 landed, 3 tile loop starts, 4+u iteration u done, 62 O stored), tools/w64_stamps.py

namespace fa {

template <int D> FA_DEVINL constexpr int k_swz_w(int row) { return D == 128 ? (row & 15) : ((row >> 1) & 7); }
template <int D> FA_DEVINL constexpr int v_swz_w(int row) { return D == 128 ? (row & 3) : ((row >> 1) & 1); }

// ---- MFMA in inline asm: operand register classes are part of the design (see header) ----------------------------
// O and the Q fragments are NOT C++ values: they live in accumulator registers named literally in the asm (fa_fwd_w64_regs.h).
// (As "+a" operands they worked, but every join of the cold rescale path with the hot loop made hipcc shuffle whole
// accumulator tuples through VGPRs -- 144 v_accvgpr moves in a 32-MFMA step.)
#define FA_W64_MFMA(E_) (std::is_same<E_, __bf16>::value ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x16_f16")
constexpr int W64_Q_BASE = 128;   // first accumulator register of the Q fragments
// d(VGPR) = a(VGPR) . Qfrag(AGPR, fragment F) + c(VGPR)       first k-step of a score chain (c = -m broadcast)
template <typename E, int F> FA_DEVINL void mfma_s_first(f32x16& d, u32x4 a, const f32x16& c) {
  if constexpr (std::is_same<E, __bf16>::value)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c3:%c4], %2" : "=&v"(d) : "v"(a), "v"(c), "i"(W64_Q_BASE + 4 * F), "i"(W64_Q_BASE + 4 * F + 3) : FA_W64_ACC_CLOBBERS);
  else
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%c3:%c4], %2" : "=&v"(d) : "v"(a), "v"(c), "i"(W64_Q_BASE + 4 * F), "i"(W64_Q_BASE + 4 * F + 3) : FA_W64_ACC_CLOBBERS);
}
// d(VGPR) += a(VGPR) . Qfrag(AGPR)
template <typename E, int F> FA_DEVINL void mfma_s_acc(f32x16& d, u32x4 a) {
  if constexpr (std::is_same<E, __bf16>::value)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(d) : "v"(a), "i"(W64_Q_BASE + 4 * F), "i"(W64_Q_BASE + 4 * F + 3) : FA_W64_ACC_CLOBBERS);
  else
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%c2:%c3], %0" : "+v"(d) : "v"(a), "i"(W64_Q_BASE + 4 * F), "i"(W64_Q_BASE + 4 * F + 3) : FA_W64_ACC_CLOBBERS);
}
// O tuple T (AGPR a[16T:16T+15]) += a(VGPR) . b(VGPR)
template <typename E, int T> FA_DEVINL void mfma_o_acc(u32x4 a, u32x4 b) {
  if constexpr (std::is_same<E, __bf16>::value)
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" : : "v"(a), "v"(b), "i"(16 * T), "i"(16 * T + 15) : FA_W64_ACC_CLOBBERS);
  else
    asm volatile("v_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" : : "v"(a), "v"(b), "i"(16 * T), "i"(16 * T + 15) : FA_W64_ACC_CLOBBERS);
}
template <typename E, int D>
__global__ void __launch_bounds__(256, 1) fa_fwd_w64_kernel(const FwdK p) {
  using T = ElemTraits<E>;
  using V8 = typename T::v8;
  constexpr int NW = 4, QB = 2, BM = NW * 64, BN = 64, CPR = D / 8;
  constexpr int ROW_BYTES = D * 2;
  constexpr int TILE_BYTES = BN * ROW_BYTES;
  constexpr int KS = D / 16;
  constexpr int DB = D / 32;
  static_assert(D == 64 || D == 128, "head dims built natively: 64, 128");
  constexpr float kLn2 = 0.6931471805599453f;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS: K0 | K1 | V0 | V1 (the O tile of the epilogue is staged over them, 256 padded rows) | Q block (K-style swizzle; each
  // wave loads and reads only its own 64 rows, so the NEXT block's Q can be prefetched here as soon as this block's
  // fragments sit in their registers)
  char FA_LDS* lds = (char FA_LDS*)smem;
  constexpr int STAGE_BYTES = BM * (ROW_BYTES + 16);
  constexpr int Q_OFF = (4 * TILE_BYTES > STAGE_BYTES ? 4 * TILE_BYTES : STAGE_BYTES + 1023) / 1024 * 1024;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, qi = lane & 31;

  const int d_row = lane / CPR, d_pc = lane % CPR;   // row inside a 1-KiB DMA piece, physical 16-byte chunk
  constexpr int RPD = 1024 / ROW_BYTES;              // tile rows per DMA instruction

  // ---- persistent workgroup: blocks vb = blockIdx.x, + gridDim.x, ...  (grid = one workgroup per CU when the dense grid is
  // larger; a grid as large as the block count degenerates to one block per workgroup).  vb -> (batch, head, query block) is
  // the XCD-aware, heavy-first order of xcd_interleave; under a right-bounded mask every other round walks the blocks of its
  // units lightest-first instead, so that a CU's rounds add up to the same work (the hardware dispatcher balanced this
  // dynamically; a static walk has to do it by construction).
  struct Blk { int b, h, m_block, sq, sk; int64_t q_row0, k_row0, q_boff, k_boff, v_boff, o_boff; };
  const int n_virtual = p.persist_total > 0 ? p.persist_total : (int)gridDim.x;
  const bool snake = p.persist_total > 0 && p.wr >= 0 && p.unit_size > 1 && ((int)(gridDim.x / 8) % p.unit_size) == 0;
  auto decode = [&](int vb, int round, Blk& k) __attribute__((always_inline)) -> bool {
    if (vb >= n_virtual) return false;
    if (p.work_list) {  // varlen: non-empty blocks only, heaviest first (fa_varlen_schedule_kernel)
      if (!work_list_item(p.work_list, vb, p.h, p.h_k, k.b, k.h, k.m_block)) return false;
    } else {
      int bid = vb;
      if (snake && (round & 1)) {   // same unit, mirrored item
        const int slot = vb / 8, item = slot % p.unit_size;
        bid = vb + 8 * (p.unit_size - 1 - 2 * item);
      }
      const int w = xcd_interleave(bid, p.n_units, p.unit_size, p.unit_hpx);
      if (w < 0) return false;
      const int bh = w / p.nmb;
      const int mbr = w - bh * p.nmb;
      k.m_block = (p.wr >= 0) ? (p.nmb - 1 - mbr) : mbr;
      k.b = bh / p.h;
      k.h = bh - k.b * p.h;
    }
    k.sq = p.sq; k.sk = p.sk; k.q_row0 = 0; k.k_row0 = 0;
    const int bkv = p.kv_batch_idx ? p.kv_batch_idx[k.b] : k.b;
    k.q_boff = (int64_t)k.b * p.q_bs; k.k_boff = (int64_t)bkv * p.k_bs; k.v_boff = (int64_t)bkv * p.v_bs; k.o_boff = (int64_t)k.b * p.o_bs;
    if (p.cu_q) { const int c0 = p.cu_q[k.b]; k.sq = p.cu_q[k.b + 1] - c0; k.q_row0 = c0; k.q_boff = 0; k.o_boff = 0; }
    if (p.cu_k) { const int c0 = p.cu_k[k.b]; k.sk = p.cu_k[k.b + 1] - c0; k.k_row0 = c0; k.k_boff = 0; k.v_boff = 0; }
    if (p.seqused_k) k.sk = min(p.seqused_k[k.b] + p.seqused_add, p.sk);
    if (p.leftpad_k) {
      const int lp = p.leftpad_k[k.b];
      k.sk = max(0, k.sk - lp);
      k.k_row0 += lp;
    }
    return k.m_block * BM < k.sq;
  };
  // this wave's 64 rows of a block's Q -> its rows of the LDS Q region (coalesced DMA, K-style swizzle on the source chunk)
  constexpr int QDMA = (BM * ROW_BYTES) / 1024 / NW;   // 1-KiB pieces of a block's Q per wave
  auto q_base = [&](const Blk& k) __attribute__((always_inline)) { return (const E*)p.q + k.q_boff + k.q_row0 * p.q_rs + (int64_t)k.h * p.q_hs; };
  auto dma_q_piece = [&](const E* qsrc, int row0, int sq_, int i) __attribute__((always_inline)) {   // piece i of this wave's rows; row0 = first row of the block
    const int row = (wave * QDMA + i) * RPD + d_row;
    const int grow = min(row0 + row, sq_ - 1);
    const int c = d_pc ^ k_swz_w<D>(row);
    lds_dma_16B(qsrc + (int64_t)grow * p.q_rs + c * 8, lds + Q_OFF + (wave * QDMA + i) * 1024);
  };
  auto dma_q = [&](const Blk& k) __attribute__((always_inline)) {
    const E* qsrc = q_base(k);
#pragma unroll
    for (int i = 0; i < QDMA; ++i) dma_q_piece(qsrc, k.m_block * BM, k.sq, i);
  };
  int q_in_lds = -1;   // virtual block whose Q this wave has prefetched into the LDS Q region

  for (int vb = blockIdx.x, round = 0; vb < n_virtual; vb += gridDim.x, ++round) {
#if FA_W64_ABL & (256 | 2048)
  const long long abl_tk = clock64();
#endif
#if FA_W64_ABL & 2048
  int abl_st = 0, abl_n = 4;
#define FA_W64_STAMP(idx_) do { const int sv_ = (int)(clock64() - abl_tk); abl_st = (lane == (idx_)) ? sv_ : abl_st; } while (0)
#else
#define FA_W64_STAMP(idx_) ((void)0)
#endif
  Blk blk;
  if (!decode(vb, round, blk)) continue;   // (uniform over the workgroup: no barrier is skipped by part of it)
  const int b = blk.b, h = blk.h, m_block = blk.m_block, sq = blk.sq, sk = blk.sk;
  const int hk = h / p.hk_ratio;
  const int64_t q_row0 = blk.q_row0, k_row0 = blk.k_row0, q_boff = blk.q_boff, k_boff = blk.k_boff, v_boff = blk.v_boff, o_boff = blk.o_boff;
  const int m0 = m_block * BM;

  const E* __restrict__ kp = (const E*)p.k + k_boff + k_row0 * p.k_rs + (int64_t)hk * p.k_hs;
  const E* __restrict__ vp = (const E*)p.v + v_boff + k_row0 * p.v_rs + (int64_t)hk * p.v_hs;
  E* __restrict__ op = (E*)p.o + o_boff + q_row0 * p.o_rs + (int64_t)h * p.o_hs;
  float* __restrict__ lsep = p.cu_q ? (p.lse + (int64_t)h * p.total_q + q_row0) : (p.lse + ((int64_t)b * p.h + h) * p.sq);

  const int shift = sk - sq;
  const int blk_last = min(m0 + BM, sq) - 1;
  int kmax = sk - 1, kmin = 0;
  if (p.wr >= 0) kmax = min(kmax, blk_last + shift + p.wr);
  if (p.wl >= 0) kmin = max(0, m0 + shift - p.wl);
  const int n_min = kmin / BN;
  const int n_max = (kmax >= kmin) ? (kmax / BN + 1) : n_min;
  const int n_tiles = n_max - n_min;
  const int n_steps = 2 * n_tiles;
  const int key_base = n_min * BN;  // first key of step 0

  const int w_row0 = m0 + wave * 64;
  const int w_row1 = min(w_row0 + 63, sq - 1);
  const bool wave_valid = w_row0 < sq;
  const int w_kmax = (p.wr >= 0) ? min(sk - 1, w_row1 + shift + p.wr) : sk - 1;
  const int w_kmin = (p.wl >= 0) ? max(0, w_row0 + shift - p.wl) : 0;
  const int w_full_hi = (p.wr >= 0) ? min(sk - 1, w_row0 + shift + p.wr) : sk - 1;
  const int w_full_lo = (p.wl >= 0) ? (w_row1 + shift - p.wl) : 0;
  int lim_hi[QB], lim_lo[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int my_row = w_row0 + 32 * qb + qi;
    lim_hi[qb] = (p.wr >= 0) ? min(sk - 1, my_row + shift + p.wr) : sk - 1;
    lim_lo[qb] = (p.wl >= 0) ? (my_row + shift - p.wl) : 0;
  }
  const float thr = p.rescale_thr;

  auto step_active = [&](int i) __attribute__((always_inline)) {
    const int k0 = key_base + 32 * i;
    return wave_valid && (i >= 0) && (i < n_steps) && (k0 <= w_kmax) && (k0 + 31 >= w_kmin);
  };
  auto step_needs_mask = [&](int i) __attribute__((always_inline)) {
    const int k0 = key_base + 32 * i;
    return (k0 + 31 > w_full_hi) || (k0 < w_full_lo);
  };

  // ---- K/V tiles: global -> LDS by DMA through a buffer descriptor.  The LDS image is lane-linear, so the XOR swizzles
  // are applied to the per-lane SOURCE chunk.  A wave issues its DPW pieces of a tile from ONE statement: M0 = LDS base
  // of piece 0, pieces 1.. by the instruction offset (added to the LDS AND the memory address, hence the -1024*i folded
  // into each piece's lane offset).  Rows past the last key are clamped to the last key (finite data, masked to -inf).
  constexpr int NDMA = TILE_BYTES / 1024;          // DMA instructions per tile
  constexpr int DPW = NDMA / NW;                   // per wave: 4 (D = 128) or 2 (D = 64)
  unsigned koff_l[DPW], voff_l[DPW];
#pragma unroll
  for (int i = 0; i < DPW; ++i) {
    const int row = (wave * DPW + i) * RPD + d_row;
    const int kc = d_pc ^ k_swz_w<D>(row);
    const int vc = ((((d_pc >> 2) ^ v_swz_w<D>(row)) << 2) | (d_pc & 3));
    koff_l[i] = (unsigned)(row * (int)p.k_rs + kc * 8) * 2u - (unsigned)(i * 1024);
    voff_l[i] = (unsigned)(row * (int)p.v_rs + vc * 8) * 2u - (unsigned)(i * 1024);
  }
  auto make_srd = [&](const void* base, int64_t row_stride) __attribute__((always_inline)) {
    const unsigned long long a = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi16 = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu;
    const unsigned long long bytes = sk > 0 ? ((unsigned long long)(sk - 1) * (unsigned long long)row_stride + D) * 2ull : 0ull;
    const unsigned nrec = __builtin_amdgcn_readfirstlane((unsigned)(bytes > 0xffffffffull ? 0xffffffffull : bytes));
    u32x4 s = {lo, hi16, nrec, 0x00020000u};
    return s;
  };
  const u32x4 k_srd = make_srd(kp, p.k_rs), v_srd = make_srd(vp, p.v_rs);
  auto dma_pieces = [&](const u32x4& srd, const unsigned (&vo)[DPW], unsigned lds_dst) __attribute__((always_inline)) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
    if constexpr (DPW == 4) {
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                   "buffer_load_dwordx4 %1, %6, 0 offen lds\n\t"
                   "buffer_load_dwordx4 %2, %6, 0 offen offset:1024 lds\n\t"
                   "buffer_load_dwordx4 %3, %6, 0 offen offset:2048 lds\n\t"
                   "buffer_load_dwordx4 %4, %6, 0 offen offset:3072 lds\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(vo[0]), "v"(vo[1]), "v"(vo[2]), "v"(vo[3]), "s"(dst), "s"(srd) : "memory");
    } else {
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                   "buffer_load_dwordx4 %1, %4, 0 offen lds\n\t"
                   "buffer_load_dwordx4 %2, %4, 0 offen offset:1024 lds\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(vo[0]), "v"(vo[DPW - 1]), "s"(dst), "s"(srd) : "memory");
    }
  };
  auto dma_tile = [&](auto isvc, int buf, int t) __attribute__((always_inline)) {  // t relative to n_min
    constexpr bool ISV = decltype(isvc)::value != 0;
    const int n = n_min + t;
    const int64_t rs = ISV ? p.v_rs : p.k_rs;
    const unsigned lds_dst = (unsigned)((ISV ? 2 + buf : buf) * TILE_BYTES + wave * DPW * 1024);
    unsigned vo[DPW];
    if (n * BN + BN <= sk) {
      const unsigned toff = (unsigned)n * (unsigned)(BN * 2) * (unsigned)rs;
#pragma unroll
      for (int i = 0; i < DPW; ++i) vo[i] = (ISV ? voff_l[i] : koff_l[i]) + toff;
    } else {  // last, partial tile
#pragma unroll
      for (int i = 0; i < DPW; ++i) {
        const int row = (wave * DPW + i) * RPD + d_row;
        const int grow = min(n * BN + row, sk - 1);
        const int c = ISV ? ((((d_pc >> 2) ^ v_swz_w<D>(row)) << 2) | (d_pc & 3)) : (d_pc ^ k_swz_w<D>(row));
        vo[i] = ((unsigned)grow * (unsigned)rs + (unsigned)(c * 8)) * 2u - (unsigned)(i * 1024);
      }
    }
    dma_pieces(ISV ? v_srd : k_srd, vo, lds_dst);
  };

  // ---- prologue.  The previous block's epilogue staged its O tile over the K/V buffers: nobody may refill them before every
  // wave is through with it.  Q: already prefetched by this wave during the previous block (its own rows, so its own
  // vmcnt wait at the tile barriers made them visible), else loaded now.  K_0 rides under the Q conversion.
  // The first three tiles (K_0, V_0, K_1: everything iteration 0 and 1 read) are requested here, in front of the Q conversion,
  // so that no iteration of the tile loop waits for a tile it has only just asked for.
  __syncthreads();
  FA_W64_STAMP(0);
  if (q_in_lds != vb) dma_q(blk);
  if (n_tiles > 0) { dma_tile(ICw<0>{}, 0, 0); dma_tile(ICw<1>{}, 0, 0); }
  if (n_tiles > 1) dma_tile(ICw<0>{}, 1, 1);
  auto wait_pieces = [&](int tiles_left) __attribute__((always_inline)) {   // all but the last `tiles_left` tile requests have landed
    if (tiles_left >= 2) { if constexpr (DPW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else if (tiles_left == 1) { if constexpr (DPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
    else lds_dma_wait_all();
  };
  if (q_in_lds != vb) {   // Q was not prefetched (first block of this workgroup): its pieces were requested first
    if (n_tiles > 1) { if constexpr (DPW == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else wait_pieces(n_tiles > 0 ? 2 : 0);
  }
  // this wave's B-operand fragments of Q, pre-multiplied by softmax_scale*log2(e) and rounded once to the input dtype, into
  // accumulator registers for the whole block
  {
    const float cq = p.scale_log2;
    auto load_q = [&](auto qbc, auto ksc) __attribute__((always_inline)) {
      constexpr int qb = decltype(qbc)::value, ks = decltype(ksc)::value;
      const int qbase = Q_OFF + (wave * 64 + qb * 32 + qi) * ROW_BYTES + ((hi ^ k_swz_w<D>(qi)) << 4);
      const V8 raw = bitcast_u32x4<V8>(*(const u32x4 FA_LDS*)(unsigned long)(unsigned)(qbase ^ (ks << 5)));
      V8 sc;
#pragma unroll
      for (int j = 0; j < 8; ++j) sc[j] = (E)((float)raw[j] * cq);
      acc_write_frag<W64_Q_BASE + 4 * (qb * KS + ks)>(__builtin_bit_cast(u32x4, sc));
    };
    auto load_q_all = [&](auto qbc) __attribute__((always_inline)) {
      load_q(qbc, ICw<0>{}); load_q(qbc, ICw<1>{}); load_q(qbc, ICw<2>{}); load_q(qbc, ICw<3>{});
      if constexpr (KS == 8) { load_q(qbc, ICw<4>{}); load_q(qbc, ICw<5>{}); load_q(qbc, ICw<6>{}); load_q(qbc, ICw<7>{}); }
    };
    load_q_all(ICw<0>{});
    load_q_all(ICw<1>{});
  }
  FA_W64_STAMP(1);
  wait_pieces(n_tiles > 1 ? 2 : n_tiles > 0 ? 1 : 0);   // K_0 (requested before the conversion); V_0 and K_1 may still be on their way
  __syncthreads();
  FA_W64_STAMP(2);
  // Next block's Q rows of this wave -> LDS under this block's tile loop, a few 1-KiB pieces per iteration (each iteration's
  // tile barrier waits for the pieces requested in it).  All of them at once, as in round 2, put 64 KB per CU -- 16 MB across
  // the chip, every workgroup at the same moment -- in front of the first tiles of the loop: 16k clocks to get the requests
  // out and another 10k of iteration 0 waiting behind them (profiles/r03_fwd_w64_stamps.txt).
  const E* qn_src = nullptr;
  int qn_row0 = 0, qn_sq = 1, qn_done = QDMA, qn_per_iter = QDMA;
  {
    Blk nxt;
    q_in_lds = -1;
    if (p.persist_total > 0 && decode(vb + (int)gridDim.x, round + 1, nxt)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the Q region have returned
      qn_src = q_base(nxt); qn_row0 = nxt.m_block * BM; qn_sq = nxt.sq; qn_done = 0;
      qn_per_iter = n_tiles >= QDMA ? 1 : n_tiles >= QDMA / 2 ? 2 : n_tiles >= QDMA / 4 ? 4 : QDMA;
      if (n_tiles == 0) { dma_q(nxt); lds_dma_wait_all(); qn_done = QDMA; }   // no tile barrier will wait for it
      q_in_lds = vb + (int)gridDim.x;
    }
  }
  auto q_trickle = [&]() __attribute__((always_inline)) {   // called once per iteration, outside the steps (M0 is theirs inside)
    if (qn_done < QDMA) {
      for (int j = 0; j < qn_per_iter; ++j) dma_q_piece(qn_src, qn_row0, qn_sq, qn_done + j);
      qn_done += qn_per_iter;
    }
  };

  // per-lane LDS read bases: K fragment of k-step ks at ka[ks] (+ buffer / half offsets as immediates), V d-block db at va[db]
  const int kbase = qi * ROW_BYTES + ((hi ^ k_swz_w<D>(qi)) << 4);
  const int tr_i = lane & 15, tr_half = (lane >> 4) & 1;
  const int tr_rr = tr_i >> 2, tr_cc = tr_i & 3;
  const int vbase = (4 * hi + tr_rr) * ROW_BYTES + (v_swz_w<D>(tr_rr) << 6) + tr_half * 32 + tr_cc * 8;
  int ka[KS], va[DB];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) ka[ks] = kbase ^ (ks << 5);
#pragma unroll
  for (int db = 0; db < DB; ++db) va[db] = vbase ^ (db << 6);

  acc_zero_range<0>(std::make_integer_sequence<int, 32 * DB>{});   // O = 2*DB tuples: a[0 : 32*DB) (query block qb, d-block db at tuple qb*DB + db)
  float m_run[QB], l_run[QB][2], o_lag[QB];
  float thr_l[QB];          // per-lane decision threshold: -inf until the row has seen a key (any finite score moves m), then rescale_thr
  bool lag_pending = false; // wave-uniform: some o_lag != 1 is waiting to be applied to O
  f32x16 negm[QB];     // the C operand of every score chain's first MFMA: -m broadcast (0 while m = -inf), and -inf in the elements the
                       // mask hides in a step that straddles a mask boundary (prep_c) -- the matrix pipe applies the mask
  float nbase[QB];     // the value the visible elements of negm hold
  bool negm_masked = false;   // wave-uniform: negm currently carries a step's mask
  f32x16 sA[QB], sB[QB];
  u32x4 pfA[QB][2], pfB[QB][2];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = -INFINITY; l_run[qb][0] = 0.f; l_run[qb][1] = 0.f; o_lag[qb] = 1.f; thr_l[qb] = -INFINITY; nbase[qb] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { negm[qb][r] = 0.f; sA[qb][r] = 0.f; sB[qb][r] = 0.f; }
#pragma unroll
    for (int t = 0; t < 2; ++t) { pfA[qb][t] = u32x4{0u, 0u, 0u, 0u}; pfB[qb][t] = u32x4{0u, 0u, 0u, 0u}; }
  }
  bool have_cur = false, have_prev = false;

  // C operand of the score chain of step i (called right before its first MFMA): a step that straddles a mask boundary gets
  // -inf in its hidden elements, so the scores leave the matrix pipe masked (s + (-inf) = -inf: exp2 gives 0, the row maximum
  // ignores it) and the masked step is otherwise the plain step; the first plain step after it puts the broadcast back.
  // Element by element through tied asm operands, as in rescale(): the tuple must stay in its registers.
  const bool two_sided = p.wl >= 0;
  auto prep_c = [&](int i) __attribute__((always_inline)) {
    if (step_needs_mask(i)) {
      const int k0 = key_base + 32 * i;
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        const int rel_hi = lim_hi[qb] - k0 - 4 * hi;
        const int rel_lo = lim_lo[qb] - k0 - 4 * hi;
        const float nb = nbase[qb];
        float ninf = -INFINITY;
        asm volatile("" : "+v"(ninf));   // a register: the lane-mask form of v_cndmask takes no literal
        if (two_sided) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int off = acc_row(r, 0);
            const unsigned long long vis = __builtin_amdgcn_ballot_w64((off <= rel_hi) && (off >= rel_lo));
            float nv = negm[qb][r];
            asm volatile("v_cndmask_b32 %0, %3, %1, %2" : "+v"(nv) : "v"(nb), "s"(vis), "v"(ninf));
            negm[qb][r] = nv;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int off = acc_row(r, 0);
            const unsigned long long vis = __builtin_amdgcn_ballot_w64(off <= rel_hi);
            float nv = negm[qb][r];
            asm volatile("v_cndmask_b32 %0, %3, %1, %2" : "+v"(nv) : "v"(nb), "s"(vis), "v"(ninf));
            negm[qb][r] = nv;
          }
        }
      }
      negm_masked = true;
    } else if (negm_masked) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        const float nb = nbase[qb];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float nv = negm[qb][r];
          asm volatile("v_mov_b32 %0, %1" : "+v"(nv) : "v"(nb));
          negm[qb][r] = nv;
        }
      }
      negm_masked = false;
    }
  };
  // Decision on the NEXT step's scores, held as s' = s - m_base (m_base = m, or 0 while m = -inf): the row moves its maximum
  // when max(s') > thr_l, i.e. when it grew by more than rescale_thr -- or, for a row that has not seen a key yet (thr_l = -inf),
  // as soon as any score is finite (same rule as fa_fwd_il.hip: (m_new - m_run) > thr with m_new = max(m_run, max s)).
  // The cold path moves m and l at once, re-bases the pending scores and the C broadcast, and rescales O one step later
  // (after P_i.V, computed at the old scale, has been accumulated) -- the lagged rescale of fa_fwd_il.hip.
  auto rescale = [&](auto qbc, bool grow, float tmax, f32x16& s_nxt) __attribute__((always_inline)) {
    constexpr int qb = decltype(qbc)::value;
    const float m_upd = grow ? (tmax - nbase[qb]) : m_run[qb];   // grow => tmax finite
    const float m_safe = (m_upd == -INFINITY) ? 0.f : m_upd;
    const float alpha = grow ? fast_exp2(m_run[qb] - m_safe) : 1.f;
    const float delta = m_safe + nbase[qb];   // new base - old base (0 where the row did not move)
    m_run[qb] = m_upd;
    thr_l[qb] = grow ? thr : thr_l[qb];
    l_run[qb][0] *= alpha;
    l_run[qb][1] *= alpha;
    // in place, element by element through tied asm operands: a recomputed tuple would live in NEW registers and cost the
    // common path a 16-register copy at the join
    const float neg = -m_safe;
    nbase[qb] = neg;
#pragma unroll
    for (int r = 0; r < 16; ++r) {   // (writes the plain broadcast: a mask negm carried is dropped, see negm_masked below)
      float sv = s_nxt[r], nv = negm[qb][r];
      asm volatile("v_sub_f32 %0, %0, %2\n\tv_mov_b32 %1, %3" : "+v"(sv), "+v"(nv) : "v"(delta), "v"(neg));
      s_nxt[r] = sv;
      negm[qb][r] = nv;
    }
    // O of this query block (accumulator registers 16*DB*qb ..): one register at a time through one temporary; skipped when
    // no factor is pending (the first decision of every block: m = -inf -> finite, O still 0)
    if (lag_pending) acc_scale_range<16 * DB * qb>(o_lag[qb], std::make_integer_sequence<int, 16 * DB>{});
    o_lag[qb] = alpha;
  };
  // tmax = row maxima of s' (already combined across the lane halves)
  auto decide_and_rescale = [&](const float (&tmax)[QB], f32x16 (&s_nxt)[QB]) __attribute__((always_inline)) {
    const bool g0 = tmax[0] > thr_l[0], g1 = tmax[1] > thr_l[1];
    const bool any_grow = __builtin_amdgcn_ballot_w64(g0 || g1) != 0ull;
    // cold: ~4 KB of straight-line code per call site.  Left in line it sat between the steps of the tile loop and every
    // step paid an instruction-fetch miss jumping over it (18 of 63 clocks per MFMA, profiles/r02_w64_ablations.txt);
    // __builtin_expect moves it behind the loop.
    if (__builtin_expect(any_grow || lag_pending, 0)) {
      mfma_drain_acc();  // O is about to be read by the VALU
      rescale(ICw<0>{}, g0, tmax[0], s_nxt[0]);
      rescale(ICw<1>{}, g1, tmax[1], s_nxt[1]);
      negm_masked = false;
      lag_pending = any_grow;
    }
  };
  auto row_max = [&](const f32x16& s) __attribute__((always_inline)) {
    float t = vmax3(s[0], s[1], s[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) t = vmax3(t, s[r], s[r + 1]);
    t = vmax2(t, s[15]);
    return vhalf_max(t);
  };

  // ---- generic (head / tail) step: compiler-ordered, drains the matrix pipe before the VALU touches MFMA results ------
  auto generic_step = [&](int par, auto halfc, int i, f32x16 (&s_cur)[QB], f32x16 (&s_nxt)[QB], const u32x4 (&pf_prev)[QB][2],
                          u32x4 (&pf_cur)[QB][2]) __attribute__((always_inline)) {
    constexpr int half = decltype(halfc)::value;
    const bool do_qk = step_active(i + 1);
    const bool do_sm = have_cur;
    const bool do_pv = have_prev;
    const int koff = par * TILE_BYTES + half * 32 * ROW_BYTES;
    const int voff = (2 + (par ^ 1)) * TILE_BYTES + half * 32 * ROW_BYTES;
    if (do_qk) {
      prep_c(i + 1);
      u32x4 kf_nxt = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)(ka[0] + koff);
      static_for<KS>([&](auto ksc) __attribute__((always_inline)) {
        constexpr int ks = decltype(ksc)::value;
        const u32x4 kf = kf_nxt;
        if constexpr (ks + 1 < KS) kf_nxt = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)(ka[ks + 1] + koff);
        if constexpr (ks == 0) {
          mfma_s_first<E, 0>(s_nxt[0], kf, negm[0]);
          mfma_s_first<E, KS>(s_nxt[1], kf, negm[1]);
        } else {
          mfma_s_acc<E, ks>(s_nxt[0], kf);
          mfma_s_acc<E, KS + ks>(s_nxt[1], kf);
        }
      });
    }
    if (do_sm) {  // s_cur is at least one whole step old: safe to read
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float p0 = fast_exp2(s_cur[qb][r]), p1 = fast_exp2(s_cur[qb][r + 1]);
          s_cur[qb][r] = p0; s_cur[qb][r + 1] = p1;
          ps0 += p0; ps1 += p1;
        }
        l_run[qb][0] += ps0;
        l_run[qb][1] += ps1;
      }
    }
    if (do_pv) {
      auto rd_v = [&](int g) __attribute__((always_inline)) {
        const char FA_LDS* a0 = (const char FA_LDS*)(unsigned long)(unsigned)(va[g % DB] + voff + (16 * (g / DB)) * ROW_BYTES);
        const s16x4 vlo = lds_read_tr16(a0), vhi = lds_read_tr16(a0 + 8 * ROW_BYTES);
        return __builtin_bit_cast(u32x4, combine_tr<V8>(vlo, vhi));
      };
      u32x4 vf_nxt = rd_v(0);
      static_for<2 * DB>([&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        const u32x4 vf = vf_nxt;
        if constexpr (g + 1 < 2 * DB) vf_nxt = rd_v(g + 1);
        mfma_o_acc<E, g % DB>(vf, pf_prev[0][g / DB]);
        mfma_o_acc<E, DB + g % DB>(vf, pf_prev[1][g / DB]);
      });
    }
    {
      float tmax[QB] = {-INFINITY, -INFINITY};
      if (do_qk) {
        mfma_drain_v(s_nxt[0], s_nxt[1]);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) tmax[qb] = row_max(s_nxt[qb]);
      }
      decide_and_rescale(tmax, s_nxt);   // without fresh scores tmax = -inf never grows; a pending O factor is still applied
    }
    if (do_sm) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          V8 x;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) x[jj] = (E)s_cur[qb][8 * t + jj];
          pf_cur[qb][t] = __builtin_bit_cast(u32x4, x);
        }
    }
    have_prev = do_sm;
    have_cur = do_qk;
  };

  // ---- steady-state step: NG MFMA gaps, everything else hand-assigned to a gap ---------------------------------------
  //   gaps 0 .. 2KS-1      : S_{i+1}[qb] chain, k-step g/2 (the K fragment read once, used by both query blocks)
  //   gaps 2KS .. NG-1     : O[qb][db] += V^T.P_{i-1}[qb], op (g - 2KS)/2 (the V fragment read once, used by both)
  //   VALU per gap: exp2 + row-sum of P_i elements (3/4 of them under the score chain), bf16/f16 packing of P_i, the
  //   row-max tree of S_{i+1} and its cross-half combine under the PV half; only two compares and the branch are left for
  //   the step boundary (an all-at-the-end decision measured 5.8 of 52 clocks per MFMA: it runs after the last MFMA has
  //   been issued, i.e. with nothing to hide behind).  The step's K or V tile DMA pieces sit in the odd gaps 1, 3, ...
  //   (DPW pieces: M0 is written with the first one and must survive until the last -- hipcc emits no M0 use in this kernel,
  //   checked in the ISA by tools/isa_blocks.py --m0).
  auto fast_step = [&](auto parc, auto halfc, int i_nxt, f32x16 (&s_cur)[QB], f32x16 (&s_nxt)[QB],
                       const u32x4 (&pf_prev)[QB][2], u32x4 (&pf_cur)[QB][2], const u32x4& dma_srd, const unsigned (&dma_off)[DPW],
                       unsigned dma_toff, unsigned dma_dst) __attribute__((always_inline)) {
    constexpr int par = decltype(parc)::value, half = decltype(halfc)::value;
    constexpr int QKG = 2 * KS, PVG = 4 * DB, NG = QKG + PVG;
    constexpr int KOFF = par * TILE_BYTES + half * 32 * ROW_BYTES;
    constexpr int VOFF = (2 + (par ^ 1)) * TILE_BYTES + half * 32 * ROW_BYTES;
    constexpr int AH = 2, RING = AH + 1;  // operand reads run AH fragment slots (2 gaps each) ahead of their MFMAs
    constexpr int NF = KS + 2 * DB;       // fragment slots per step: KS K fragments, then 2*DB V fragments
    u32x4 kfr[RING];
    s16x4 vlo[RING], vhi[RING];
    if (FA_W64_ABL & 128) {
#pragma unroll
      for (int f = 0; f < RING; ++f) { kfr[f] = pf_prev[0][0]; vlo[f] = __builtin_bit_cast(s16x4, u32x2{pf_prev[0][1][0], pf_prev[0][1][1]}); vhi[f] = vlo[f]; }
    }
    auto rd_frag = [&](int f) __attribute__((always_inline)) {
      if (FA_W64_ABL & 128) return;
      if ((FA_W64_ABL & 16) && f != 0 && f != KS) {
        if (f < KS) kfr[f % RING] = kfr[0];
        else if (f < NF) { vlo[f % RING] = vlo[KS % RING]; vhi[f % RING] = vhi[KS % RING]; }
        return;
      }
      if (f < KS) {
        kfr[f % RING] = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)(ka[f] + KOFF);
      } else if (f < NF) {
        const int op = f - KS;
        const char FA_LDS* a0 = (const char FA_LDS*)(unsigned long)(unsigned)(va[op % DB] + VOFF + (16 * (op / DB)) * ROW_BYTES);
        vlo[f % RING] = lds_read_tr16(a0);
        vhi[f % RING] = lds_read_tr16(a0 + 8 * ROW_BYTES);
      }
    };
    // element e of P_i (e = 16*qb + r): done-by-gap schedule
    auto el_end = [](int x) constexpr { return x <= QKG ? (24 * x) / QKG : (24 + ((x - QKG) * 16) / PVG > 32 ? 32 : 24 + ((x - QKG) * 16) / PVG); };
    float pe[QB][16];   // P_i as scalars (writing them back into the score tuples makes hipcc copy whole 16-register tuples)
    float tmax[QB] = {-INFINITY, -INFINITY};
    prep_c(i_nxt);   // (a step that straddles a mask boundary: the mask goes into the chain's C operand; nothing else differs)
    // gap (inside the PV half) schedule of the row-max work of query block mq
    constexpr int UPG = PVG >= 16 ? 1 : 2;        // max3 units per gap
    auto tree_g0 = [](int mq) constexpr { return 1 + mq; };   // the chain of block mq retired at gap QKG - 2 + mq
    auto hm_gap = [&](int mq) constexpr { return tree_g0(mq) + 8 / UPG; };           // cross-half combine right after the tree
#pragma unroll
    for (int f = 0; f < AH; ++f) rd_frag(f);
    __builtin_amdgcn_sched_barrier(0);
    static_for<NG>([&](auto xc) __attribute__((always_inline)) {
      constexpr int x = decltype(xc)::value;
      constexpr int f = x / 2, qb = x & 1;
      if constexpr (qb == 0) rd_frag(f + AH);
      if constexpr (x < QKG) {
        if constexpr (f == 0) mfma_s_first<E, qb * KS>(s_nxt[qb], kfr[f % RING], negm[qb]);
        else mfma_s_acc<E, qb * KS + f>(s_nxt[qb], kfr[f % RING]);
      } else {
        constexpr int op = f - KS;
        const u32x4 vf = __builtin_bit_cast(u32x4, combine_tr<V8>(vlo[f % RING], vhi[f % RING]));
        mfma_o_acc<E, qb * DB + op % DB>(vf, pf_prev[qb][op / DB]);
      }
      // this step's DMA pieces (K_{u+1} in the first step of an iteration, V_u in the second): odd gaps 1, 3, ..
      if constexpr ((x & 1) && (x / 2) < DPW && !(FA_W64_ABL & 32)) {
        constexpr int pc = x / 2;
        const unsigned vo = dma_off[pc] + dma_toff;
        if constexpr (pc == 0)
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" : : "v"(vo), "s"(dma_dst), "s"(dma_srd) : "memory");
        else
          asm volatile("buffer_load_dwordx4 %0, %1, 0 offen offset:%c2 lds" : : "v"(vo), "s"(dma_srd), "i"(1024 * pc) : "memory");
      }
      // exp2 + row sums (two running sums per query block, carried across steps)
#pragma unroll
      for (int e = el_end(x); e < el_end(x + 1); ++e) {
        const int eq = e >> 4, r = e & 15;
        pe[eq][r] = (FA_W64_ABL & 1) ? s_cur[eq][r] : fast_exp2(s_cur[eq][r]);
        if (!(FA_W64_ABL & (2 | 64))) l_run[eq][r & 1] += pe[eq][r];
      }
      if ((FA_W64_ABL & 64) && x > 0) {  // the adds of the PREVIOUS gap's elements
#pragma unroll
        for (int e = el_end(x - 1); e < el_end(x); ++e) l_run[e >> 4][e & 1] += pe[e >> 4][e & 15];
      }
      if constexpr (x >= QKG) {
        constexpr int y = x - QKG;  // gap inside the PV half
        // packing of P_i: PVG gaps, 16 conversions (one packed register each)
        constexpr int CPG = 16 / PVG > 0 ? 16 / PVG : 1;
#pragma unroll
        for (int c = y * CPG; c < (y + 1) * CPG && c < 16 && !(FA_W64_ABL & 4); ++c) {
          const int j = c >> 2, m = c & 3, cq = j >> 1, t = j & 1;
          using V2 = __attribute__((ext_vector_type(2))) E;
          V2 pr;
          pr[0] = (E)pe[cq][8 * t + 2 * m];
          pr[1] = (E)pe[cq][8 * t + 2 * m + 1];
          unsigned pw = __builtin_bit_cast(unsigned, pr);
          asm volatile("" : "+v"(pw));   // pinned to this gap (hipcc otherwise sinks the conversions into the next step's head)
          pf_cur[cq][t][m] = pw;
        }
        // the row-max tree of the fresh scores, the cross-half combine
#pragma unroll
        for (int mq = 0; mq < QB; ++mq) {
          const int g0 = tree_g0(mq);
          if (y >= g0 && y < g0 + 8 / UPG && !(FA_W64_ABL & 8)) {
#pragma unroll
            for (int u = (y - g0) * UPG; u < (y - g0 + 1) * UPG; ++u)  // 16 values in 8 ops: max3(s0,s1,s2), 6 x max3(t,.,.), max(t,s15)
              tmax[mq] = u == 0 ? vmax3(s_nxt[mq][0], s_nxt[mq][1], s_nxt[mq][2])
                                : u < 7 ? vmax3(tmax[mq], s_nxt[mq][2 * u + 1], s_nxt[mq][2 * u + 2]) : vmax2(tmax[mq], s_nxt[mq][15]);
          }
          if (y == hm_gap(mq) && !(FA_W64_ABL & (8 | 1024))) tmax[mq] = vhalf_max(tmax[mq]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if (!(FA_W64_ABL & 1024)) {
#pragma unroll
      for (int mq = 0; mq < QB; ++mq)
        if (hm_gap(mq) >= PVG && !(FA_W64_ABL & 8)) tmax[mq] = vhalf_max(tmax[mq]);   // (no gap left for it)
      decide_and_rescale(tmax, s_nxt);
    }
  };

  // iteration u (0..n_tiles): steps 2u-1 and 2u read K_u (kbuf[u&1]) and V_{u-1} (vbuf[(u-1)&1]); K_{u+1} and V_u are
  // DMA'd during the iteration into the buffers it does not read.  Head / steady-state / tail split as in fa_fwd_il.hip.
  // (A wave's two drain steps after its last scored step -- {exp2 + P.V}, {P.V} -- stay on the generic path.  Tried and
  // measured slower or wrong: running them as MASKED steady-state steps, whose discarded score chains cost more than the
  // generic step saves (config 3: 949 -> 858 TFLOP/s); compile-time variants of the steady-state step without the score
  // chain, which were right one at a time and wrong -- rows without any visible key -- when both were instantiated.)
  int uf_lo = 1, uf_hi = 0;
  if (wave_valid && n_tiles > 0) {
    const int a_lo = max(0, (w_kmin - key_base) >> 5);
    const int a_hi = min(n_steps - 1, (w_kmax - key_base) >> 5);
    uf_lo = (a_lo + 3) >> 1;
    uf_hi = (a_hi - 1) >> 1;
    // the steady-state iteration u DMAs the tiles n_min+u+1 (K) and n_min+u (V) with unclamped row addresses: keep the
    // partial last tile of a sequence (sk % 64 != 0) out of it -- the generic iterations clamp its rows
    if (sk % BN != 0) uf_hi = min(uf_hi, sk / BN - n_min - 2);
  }
#ifdef FA_W64_NOFAST   // debugging: every iteration through the generic step
  uf_lo = 1; uf_hi = 0;
#endif
  auto iter_head = [&](int u) __attribute__((always_inline)) {
    if ((FA_W64_ABL & 32) && u > 1) return;
    q_trickle();
    if (u == 0) return;   // K_1 and V_0 were requested in the prologue
    const int par = u & 1;
    if (u + 1 < n_tiles) dma_tile(ICw<0>{}, par ^ 1, u + 1);
    if (u < n_tiles) dma_tile(ICw<1>{}, par, u);
  };
  auto iter_tail = [&]() __attribute__((always_inline)) {
    if (FA_W64_ABL & 512) return;
    lds_dma_wait_all();
    __syncthreads();
#if FA_W64_ABL & 2048
    if (abl_n < 62) { FA_W64_STAMP(abl_n); ++abl_n; }
#endif
  };
  auto generic_iter = [&](int u) __attribute__((always_inline)) {
    iter_head(u);
    generic_step(u & 1, ICw<0>{}, 2 * u - 1, sA, sB, pfA, pfB);
    generic_step(u & 1, ICw<1>{}, 2 * u, sB, sA, pfB, pfA);
    iter_tail();
  };
#if FA_W64_ABL & 256
  const long long abl_t0 = clock64();
#endif
  FA_W64_STAMP(3);
  if (n_tiles > 0) {
    int u = 0;
    const int head_end = min(max(uf_lo, 0), n_tiles + 1);
    for (; u < head_end; ++u) generic_iter(u);
    const unsigned wave_dst = (unsigned)(wave * DPW * 1024);
    auto fast_iter = [&](auto parc, int uu) __attribute__((always_inline)) {
      constexpr int par = decltype(parc)::value;
      // K_{u+1} rides in the first step, V_u in the second (full tiles only, see uf_hi; a K tile past the last one lands in
      // the buffer nobody reads again: rows past the end of the buffer descriptor are out of range, never a fault)
      q_trickle();
      const unsigned toff_k = (unsigned)(n_min + uu + 1) * (unsigned)(BN * 2) * (unsigned)p.k_rs;
      const unsigned toff_v = (unsigned)(n_min + uu) * (unsigned)(BN * 2) * (unsigned)p.v_rs;
      fast_step(parc, ICw<0>{}, 2 * uu, sA, sB, pfA, pfB, k_srd, koff_l, toff_k, __builtin_amdgcn_readfirstlane((unsigned)((par ^ 1) * TILE_BYTES) + wave_dst));
      fast_step(parc, ICw<1>{}, 2 * uu + 1, sB, sA, pfB, pfA, v_srd, voff_l, toff_v, __builtin_amdgcn_readfirstlane((unsigned)((2 + par) * TILE_BYTES) + wave_dst));
      have_prev = step_active(2 * uu);      // P_{2u} is packed and S_{2u+1} is pending -- what the generic steps that follow
      have_cur = step_active(2 * uu + 1);   // expect; past the wave's last scored step they are 0 / -inf and simply dropped
      iter_tail();
    };
    if (u <= uf_hi && (u & 1)) { fast_iter(ICw<1>{}, u); ++u; }
    for (; u + 1 <= uf_hi; u += 2) {
      fast_iter(ICw<0>{}, u);
      fast_iter(ICw<1>{}, u + 1);
    }
    if (u <= uf_hi) { fast_iter(ICw<0>{}, u); ++u; }
    for (; u <= n_tiles; ++u) generic_iter(u);
  }

#if FA_W64_ABL & 256
  const long long abl_t1 = clock64();
  const float abl_ticks = (float)(abl_t1 - abl_t0) / (float)(n_tiles > 0 ? n_tiles * 16 * DB : 1);
#endif
  if (wave_valid) {
  mfma_drain_acc();
  // O tile through LDS (the K/V buffers are free after the last tile barrier; the Q region may already hold the next block's
  // rows and is not touched): whole-row stores
  static_for<QB>([&](auto qbc) __attribute__((always_inline)) {
    constexpr int qb = decltype(qbc)::value;
    f32x16 o_v[DB];
    static_for<DB>([&](auto dbc) __attribute__((always_inline)) {
      constexpr int db = decltype(dbc)::value;
      acc_read_tuple<16 * (qb * DB + db)>(o_v[db]);
#pragma unroll
      for (int r = 0; r < 16; ++r) o_v[db][r] *= o_lag[qb];
    });
    const float l_tot = half_sum(l_run[qb][0] + l_run[qb][1]);
    const bool dead = (l_tot == 0.f) || (l_tot != l_tot);
    const float inv = dead ? 1.f : 1.f / l_tot;
    const int row0 = w_row0 + 32 * qb;
    if (row0 < sq) {
      store_tile_via_lds<E, D>(lds + (wave * 64 + qb * 32) * (ROW_BYTES + 16), o_v, inv, op + (int64_t)row0 * p.o_rs, p.o_rs, sq - row0, lane);
      const int my_row = row0 + qi;
#if FA_W64_ABL & 256   // rows = 0 mod 4: clocks per MFMA of the tile loop; 1: prologue clocks; 2: tile-loop clocks; 3: epilogue clocks so far
      if (my_row < sq && hi == 0) {
        const int sel = my_row & 3;
        lsep[my_row] = sel == 0 ? abl_ticks : sel == 1 ? (float)(abl_t0 - abl_tk) : sel == 2 ? (float)(abl_t1 - abl_t0) : (float)(clock64() - abl_t1);
      }
#elif FA_W64_ABL & 2048
      if (qb == QB - 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FA_W64_STAMP(62);
        if (w_row0 + lane < sq) lsep[w_row0 + lane] = (float)abl_st;
      }
#else
      if (my_row < sq && hi == 0) lsep[my_row] = dead ? INFINITY : (m_run[qb] * kLn2 + __logf(l_tot));
#endif
    }
  });
  }  // wave_valid
  }  // persistent block loop
}

template <typename E, int D>
static int launch_fwd_w64_t(const FwdK& p, hipStream_t stream) {
  constexpr int STAGE = 256 * (D * 2 + 16);
  constexpr int smem = ((4 * 64 * D * 2 > STAGE ? 4 * 64 * D * 2 : STAGE + 1023) / 1024 * 1024) + 256 * D * 2;  // K/V buffers | O staging, then the Q block
  auto kern = fa_fwd_w64_kernel<E, D>;
  static std::atomic<unsigned long long> attr_mask{0};  // the kernel addresses LDS by byte offset: the dynamic segment must start at 0
  if (ensure_dyn_lds(attr_mask, (const void*)kern, smem, true) != 0) return -1;
  const long long total = p.work_list ? (long long)p.work_bound * p.h : units_grid(p.n_units, p.unit_size);
  if (total <= 0) return 0;
  // dense grids larger than the chip: one persistent workgroup per CU walks the blocks (multiple of 8 keeps vb % 8 = XCD);
  // work lists keep one workgroup per block (their order is already heavy-first, the dispatcher balances the tail)
  FwdK pp = p;
  long long grid = total;
  const int cus = device_cu_count();
  // (under a right-bounded mask the static walk is balanced by mirroring every other round inside its units: needs rounds
  // made of whole units)
  const bool balanced = p.wr < 0 || (p.unit_size > 1 && (cus / 8) % p.unit_size == 0);
  if (!p.work_list && cus >= 8 && total > cus && balanced && knobs().w64_persist != 0) {
    grid = cus / 8 * 8;
    pp.persist_total = (int)total;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), smem, stream, pp);
  if (hipGetLastError() != hipSuccess) return -1;
  LastSchedule& ls = last_schedule();
  ls.fwd_kernel = 3; ls.fwd_nw = 4; ls.fwd_feat = 0; ls.fwd_splits = 1; ls.fwd_list = p.work_list != nullptr; ls.d = D;
  ls.bf16 = std::is_same<E, __bf16>::value;
  snprintf(ls.name, sizeof(ls.name), "fa::fa_fwd_w64_kernel<%s,%d>", ls.bf16 ? "bf16" : "f16", D);
  return 0;
}

// build.py compiles this file twice side by side (-DFA_W64_PART=1: bf16, =2: fp16; 0 = one object): the hand-unrolled steps make it
// the slowest unit of the build.
#ifndef FA_W64_PART
#define FA_W64_PART 0
#endif
int launch_fwd_w64_bf16(const FwdK& p, int d, hipStream_t stream);
int launch_fwd_w64_f16(const FwdK& p, int d, hipStream_t stream);
#if FA_W64_PART != 1
int launch_fwd_w64_f16(const FwdK& p, int d, hipStream_t stream) {
  if (d == 128) return launch_fwd_w64_t<_Float16, 128>(p, stream);
  if (d == 64) return launch_fwd_w64_t<_Float16, 64>(p, stream);
  return -2;
}
#endif
#if FA_W64_PART != 2
int launch_fwd_w64_bf16(const FwdK& p, int d, hipStream_t stream) {
  if (d == 128) return launch_fwd_w64_t<__bf16, 128>(p, stream);
  if (d == 64) return launch_fwd_w64_t<__bf16, 64>(p, stream);
  return -2;
}
// 4 waves x 64 query rows per workgroup.  Plain attention only (no softcap / ALiBi / dropout / split keys / paged KV).
int launch_fwd_w64(const FwdK& p, int dtype_bf16, int d, hipStream_t stream) {
  if (p.softcap > 0.f || p.alibi != nullptr || p.rng != nullptr || p.n_splits > 1 || p.block_table != nullptr) return -2;
  // buffer addressing: 32-bit byte offsets from the (batch, kv-head) base
  const uint64_t span = ((uint64_t)(p.sk > 0 ? p.sk : 1) + 128) * (uint64_t)(p.k_rs > p.v_rs ? p.k_rs : p.v_rs) * 2u;
  if (span >= (1ull << 32)) return -3;
  return dtype_bf16 ? launch_fwd_w64_bf16(p, d, stream) : launch_fwd_w64_f16(p, d, stream);
}
#endif

}  // namespace fa
