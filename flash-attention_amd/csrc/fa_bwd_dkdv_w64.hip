// dK/dV kernel of the backward pass for gfx950, "64 keys per wave, one wave per SIMD" schedule (reference: the dK/dV half of
// compute_dq_dk_dv_1colblock, csrc/flash_attn/src/flash_bwd_kernel.h:457-733; same contractions and arithmetic as fa_bwd_dkdv_kernel in
// fa_bwd.hip, which keeps the feature variants, head dim 256 and the trimmed head dims).
//
// Why (profiles/r04_bwd_sq_counters.txt): the eight-wave kernel issues 2.0 LDS + 5.5 VALU + 2.4 SALU per MFMA -- each LDS fragment feeds ONE MFMA, the
// 256-register budget of two waves per SIMD forces the operand addresses to be recomputed at every read, and a wave's matrix and vector phases do
// not overlap.  Here a wave owns TWO 32-key blocks and the whole 512-entry register file:
//   * dV^T and dK^T of its 64 keys (2 x 2 x D/32 tiles of 32 x 32) are the accumulator half of the file, named literally in the asm (fa_w64_asm.h);
//   * every A operand read from LDS -- Q / dO row fragments for S = Q.K^T and dP = dO.V^T, transposed dO / Q fragments for dV^T += dO^T.P and
//     dK^T += Q^T.dS -- feeds two MFMAs, one per key block; K fragments stay in registers, V fragments (B operands of dP) are read from the V block in LDS;
//   * the wave software-pipelines its own stream over 32-query tiles.  Step s = phase A: the 4*KS MFMAs of S_s and dP_s (LDS reads, the tile DMAs; the vector
//     ALU is idle), then phase B: the 8*DB MFMAs of dV / dK of tile s-1 with the softmax arithmetic of tile s (fma, exp2, mul, packing: 4 VALU per element,
//     one element per MFMA gap) hand-placed into its gaps.  One S / dP register set, two P / dS sets: that is what fits 256 + 256 registers at D = 128.
//   * Q / dO tiles of 32 queries live in a ring of FOUR slots, DMA'd two tiles ahead: tile s+1 was published by the barrier at the end of step s-1, so the
//     first operand reads of the next step and its dP chains' C operand are requested BEFORE the end-of-step barrier (no LDS round trip behind it).
// Arithmetic (as FEAT_EXACT of fa_bwd.hip): P = exp2(S*scale*log2e - LSE*log2e) with the scale applied in fp32, dP - delta by the matrix pipe (the dP chains
// start from C = -delta), dS = P*(dP - delta), P and dS rounded to the input dtype, softmax_scale applied to dK once at the end.
#include <cstdlib>
#include <type_traits>

#include "fa_device.h"
#include "fa_kernel_params.h"
#include "fa_launch.h"
#include "fa_fwd_w64_regs.h"
#define FA_W64_CLOB FA_W64_ACC_CLOBBERS_256
#include "fa_w64_asm.h"

// LDS operand reads run ahead of their MFMAs with one explicit wait per k-step (phase A) / per two fragments (phase B).  The timing ablations and the
// one-wait-per-MFMA variant live in experiments/ablations/fa_bwd_dkdv_w64.patch (tools/ablate_dkdv64.sh).
#define FA_DKDV64_AHJ 6     // phase A: row-fragment reads run this many fragments (= MFMA gaps) ahead of their first MFMA
#define FA_DKDV64_AHT 3     // phase B: transposed-fragment reads run this many fragments (two gaps each) ahead

#ifndef FA_DKDV64_PART
#define FA_DKDV64_PART 0    // build.py compiles this file twice side by side: 1 = the dK/dV kernels, 2 = the 5-contraction backward's mixed kernel; 0 = everything
#endif

namespace fa {
namespace {

// d (arch VGPR tuple) = a . b (+ d): asm, so that hipcc cannot place a chain's accumulator in the accumulator registers the tiles below live in
template <typename E> FA_DEVINL void kv_mfma_c0(f32x16& d, u32x4 a, u32x4 b) {
  if constexpr (std::is_same<E, __bf16>::value)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b) : FA_W64_CLOB);
  else
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b) : FA_W64_CLOB);
}
template <typename E> FA_DEVINL void kv_mfma_acc(f32x16& d, u32x4 a, u32x4 b) {
  if constexpr (std::is_same<E, __bf16>::value)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b) : FA_W64_CLOB);
  else
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b) : FA_W64_CLOB);
}
// d = a . b + c, c another tuple
template <typename E> FA_DEVINL void kv_mfma_from(f32x16& d, u32x4 a, u32x4 b, const f32x16& c) {
  if constexpr (std::is_same<E, __bf16>::value)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c) : FA_W64_CLOB);
  else
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c) : FA_W64_CLOB);
}
// accumulator tile T (a[16T : 16T+15]) += a . b
template <typename E, int T> FA_DEVINL void kv_mfma_tile(u32x4 a, u32x4 b) {
  if constexpr (std::is_same<E, __bf16>::value)
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" : : "v"(a), "v"(b), "i"(16 * T), "i"(16 * T + 15) : FA_W64_CLOB);
  else
    asm volatile("v_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" : : "v"(a), "v"(b), "i"(16 * T), "i"(16 * T + 15) : FA_W64_CLOB);
}

}  // namespace

// ALIBI (under a causal right bound, where the bias -slope * |key - row - shift| is linear in the key): the exponent of P gets slope*log2e * (key - row - shift).
// With q0 = the tile's first row: the row's part, -slope * (row - q0), rides with -LSE in the score chains' C operand (stream_aux), the key's part,
// slope*log2e * (key - q0 - shift) -- one value per lane, key block and tile, from an integer difference --, is the addend of the fused multiply-add that
// applies scale*log2e (a plain multiply without ALiBi): no instruction per element.  The slopes of the group's query heads wait in a small LDS table.
//
// FEAT_CAP (softcap, round 5; reference flash_bwd_kernel.h:588 + utils.h:395-409): the cap is not linear, so LSE cannot ride in the score chains' C operand: the chains
// start from C = 0, the tile stream stores c - LSE*log2e (c = softcap*log2e) per row instead of -LSE/scale, and phase B reads those rows four at a time one group of
// elements ahead.  With y = score * scale/softcap * 2*log2e and r = 1/(2^y + 1): P = 2^(c - LSE*log2e - 2c*r), 1 - tanh^2 = 4*(r - r^2) -- twelve vector instructions
// per element instead of five, in three stages a gap apart; the 4 meets softmax_scale in dK's epilogue.
//
// DS (round 6, the 5-contraction backward, fa_bwd_c5_kernel below): the wave also hands every dS sub-tile it forms (32 queries x 32 keys, rounded to the input dtype as it
// enters the dK contraction) to the workspace BwdK::ds_ws in the image fa_device.h ds_slot describes -- four 16-byte stores per lane and step, issued in phase A's gaps of
// the step AFTER the one that formed them (their registers are rewritten only by that step's phase B), through a buffer descriptor whose range is zero for a step without
// a previous tile: no branch in the step.  `bid` = the workgroup's number in the dK/dV grid (the mixed launch passes its own).
template <typename E, int D, int FEAT, bool DS>
static __device__ __forceinline__ void dkdv_w64_body(const BwdK p, const int bid) {   // (p by value: through a reference hipcc no longer proves the descriptor words and tile offsets uniform -- asm "s" operands arrive in vector registers)
  constexpr bool ALIBI = FEAT == FEAT_ALIBI, CAP = FEAT == FEAT_CAP;
  static_assert(FEAT == 0 || ALIBI || CAP, "feature variants of this schedule: none, causal ALiBi, softcap");
  static_assert(!DS || FEAT == 0, "the dS hand-over: plain attention");
  using T = ElemTraits<E>;
  using V8 = typename T::v8;
  constexpr int NW = 4, KB = 2, BNK = NW * 64, TQ = 32;
  constexpr int CPR = D / 8, ROW_BYTES = D * 2, KS = D / 16, DB = D / 32;
  constexpr int QT = TQ * ROW_BYTES;          // one operand of a tile (Q or dO)
  constexpr int SLOT = 2 * QT;                // ring slot: Q rows | dO rows
  constexpr int NSLOT = 4;
  constexpr int OFF_V = NSLOT * SLOT;         // the V block of this workgroup's 256 keys
  constexpr int OFF_AUX = OFF_V + BNK * ROW_BYTES;
  constexpr int AUX_SLOT = 2 * TQ * 4;        // TQ x -LSE/scale, then TQ x -delta
  // Register relief at D = 128 (the kernel sits exactly on the 256 + 256 budget): the K fragments of key block 1, k-steps 1 .. KS-1, wait in a wave-private,
  // lane-linear LDS region and are read back one k-step ahead (KL fragments, 1 KiB each per wave)
  constexpr int KL = D == 128 ? KS - 1 : 0;
  constexpr int OFF_KX = OFF_AUX + NSLOT * AUX_SLOT;
  constexpr int OFF_TAB = OFF_KX + NW * KL * 1024;   // (ALIBI) slope*log2e of the group's query heads, hk_ratio + 1 words
  constexpr int RPD = 1024 / ROW_BYTES;       // tile rows per 1-KiB DMA piece
  constexpr int NP = QT / 1024, PW = NP / NW; // pieces per operand / per wave
  static_assert(D == 128 || D == 64, "head dims of this schedule: 64, 128");
  static_assert(PW == 1 || PW == 2, "DMA pieces per wave and operand");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char FA_LDS* lds = (char FA_LDS*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, ki = lane & 31;

  int b, hk, n_block;
  if (p.k_list) {  // varlen: non-empty key blocks only, heaviest first
    if (!work_list_item(p.k_list, bid, p.h_k, p.h_k, b, hk, n_block)) return;
  } else {
    const int w = xcd_interleave(bid, p.k_units, p.k_unit_size, p.k_unit_hpx);
    if (w < 0) return;
    const int bhk = w / p.nnb;
    n_block = w - bhk * p.nnb;
    b = bhk / p.h_k;
    hk = bhk - b * p.h_k;
  }
  int sq = p.sq, sk = p.sk;
  int64_t q_row0 = 0, k_row0 = 0;
  int64_t q_boff = (int64_t)b * p.q_bs, do_boff = (int64_t)b * p.do_bs;
  int64_t k_boff = (int64_t)b * p.k_bs, v_boff = (int64_t)b * p.v_bs, dk_boff = (int64_t)b * p.dk_bs, dv_boff = (int64_t)b * p.dv_bs;
  if (p.cu_q) { const int c0 = p.cu_q[b]; sq = p.cu_q[b + 1] - c0; q_row0 = c0; q_boff = 0; do_boff = 0; }
  if (p.cu_k) { const int c0 = p.cu_k[b]; sk = p.cu_k[b + 1] - c0; k_row0 = c0; k_boff = 0; v_boff = 0; dk_boff = 0; dv_boff = 0; }
  if (p.seqused_q) sq = min(sq, p.seqused_q[b]);
  if (p.seqused_k) sk = min(p.seqused_k[b], sk);   // (include/fa_gfx950.h: in the backward seqused_k can only shorten a sequence)
  const int n0 = n_block * BNK;
  if (n0 >= sk) return;
  const int n1 = min(n0 + BNK, sk);
  const int shift = sk - sq;

  const E* __restrict__ kp = (const E*)p.k + k_boff + k_row0 * p.k_rs + (int64_t)(hk >> p.kv_in_shift) * p.k_hs;   // (kv_in_shift: a GQA group split into virtual kv heads, fa_kernel_params.h)
  const E* __restrict__ vp = (const E*)p.v + v_boff + k_row0 * p.v_rs + (int64_t)(hk >> p.kv_in_shift) * p.v_hs;

  // query range that can see this key block, in tiles of 32
  int q_lo = 0, q_hi = sq - 1;
  if (p.wr >= 0) q_lo = max(0, n0 - shift - p.wr);
  if (p.wl >= 0) q_hi = min(sq - 1, n1 - 1 - shift + p.wl);
  const int m_lo = q_lo / TQ;
  const int nm = (q_hi >= q_lo) ? (q_hi / TQ + 1 - m_lo) : 0;  // tiles per query head
  const int n_steps = __builtin_amdgcn_readfirstlane(nm * p.hk_ratio);
  // Walk order of the tiles (fa_bwd.hip): downwards under a right-bounded mask, so that the key blocks of a head -- co-resident on an XCD -- read the same tile together
  const bool walk_down = p.wr >= 0 && p.wl < 0;

  // this wave's keys: key block kb = keys wk0 + 32 kb .. + 31, lane & 31 = key
  const int wk0 = n0 + wave * 64;
  const int wk_last = wk0 + 63;
  const int wk_hi = min(wk_last, sk - 1);   // last key of this wave that exists

  // K fragments (B operands of S = Q.K^T): lane = key, 8 consecutive d per k-step.  kf0: key block 0; kf1: key block 1, k-steps 0 .. KS-KL-1 (the rest in LDS)
  u32x4 kf0[KS], kf1[KS - KL];
  {
    const E* krow0 = kp + (int64_t)(wk0 + ki) * p.k_rs + 8 * hi;
    const E* krow1 = kp + (int64_t)(wk0 + 32 + ki) * p.k_rs + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kf0[ks] = ld_global_16B(krow0 + 16 * ks, wk0 + ki < sk);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u32x4 x = ld_global_16B(krow1 + 16 * ks, wk0 + 32 + ki < sk);
      if (ks < KS - KL) kf1[ks < KS - KL ? ks : 0] = x;
      else *(u32x4 FA_LDS*)(lds + OFF_KX + (wave * KL + (ks - (KS - KL))) * 1024 + lane * 16) = x;
    }
  }
  // V block -> LDS (B operands of dP = dO.V^T)
  {
    constexpr int LDV = (BNK * CPR) / 256;
#pragma unroll
    for (int i = 0; i < LDV; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / CPR, ch = idx % CPR;
      const u32x4 x = ld_global_16B(vp + (int64_t)(n0 + row) * p.v_rs + ch * 8, n0 + row < sk);
      *(u32x4 FA_LDS*)(lds + OFF_V + tile_off<D>(row, ch)) = x;
    }
  }
  if constexpr (ALIBI) {
    for (int i = tid; i <= p.hk_ratio; i += 256)   // (one word past the last head: the drain step's preload reads it and nobody uses it)
      *(float FA_LDS*)(lds + OFF_TAB + 4 * i) = p.alibi[(int64_t)b * p.alibi_bs + min(hk * p.hk_ratio + i, p.h - 1)] * 1.4426950408889634f;
  }

  // ---- tile stream: Q / dO tiles by LDS-DMA through buffer descriptors of the head's rows (rows past the sequence end are outside the range and arrive as
  // zeros), swizzle on the source chunk (the LDS image is lane-linear); -LSE*log2e and -delta of the tile's rows through a register of wave 0.
  unsigned qo[PW], doo[PW];
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int row = (wave * PW + i) * RPD + lane / CPR;
    const int c = (lane % CPR) ^ swz16<D>(row);
    qo[i] = (unsigned)(row * (int)p.q_rs + c * 8) * 2u - (unsigned)(i * 1024);
    doo[i] = (unsigned)(row * (int)p.do_rs + c * 8) * 2u - (unsigned)(i * 1024);
  }
  auto make_srd = [&](const E* base, int64_t rs) __attribute__((always_inline)) {
    const unsigned long long a = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi16 = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu;
    const unsigned long long bytes = sq > 0 ? ((unsigned long long)(sq - 1) * (unsigned long long)rs + D) * 2ull : 0ull;
    const unsigned nrec = __builtin_amdgcn_readfirstlane((unsigned)(bytes > 0xffffffffull ? 0xffffffffull : bytes));
    u32x4 sd = {lo, hi16, nrec, 0x00020000u};
    return sd;
  };
  // Tile walk, in tile indices mt (tile = rows 32*mt ..): a head's tiles are m_lo .. m_lo + nm - 1, walked downwards or upwards; the scalar bookkeeping of a
  // step is a handful of adds and compares on mt (the first version evaluated the mask geometry and the addresses from scratch: ~250 instructions per step,
  // none of them under an MFMA -- the kernel ran at 88 clocks per MFMA).
  const int mt_first = __builtin_amdgcn_readfirstlane(walk_down ? m_lo + nm - 1 : m_lo), mt_dir = walk_down ? -1 : 1;
  // stream state: tile t2 = tile s_mt of query head s_h of the group, s_left tiles of that head still to request; descriptors and LSE / delta rows of its head
  int t2 = 0, s_h = 0, s_mt = mt_first, s_left = __builtin_amdgcn_readfirstlane(nm);
  u32x4 q_srd = {0u, 0u, 0u, 0x00020000u}, do_srd = q_srd;
  const float* aux_h = p.lse;   // per lane: lanes 0-31 the head's LSE rows, lanes 32-63 its delta rows
  float s_slope = 0.f;          // (ALIBI) slope of the stream's head
  const float kif = (float)ki;
  auto stream_head = [&]() __attribute__((always_inline)) {
    const int h = hk * p.hk_ratio + s_h;
    if constexpr (ALIBI) s_slope = p.alibi[(int64_t)b * p.alibi_bs + h];
    q_srd = make_srd((const E*)p.q + q_boff + q_row0 * p.q_rs + (int64_t)h * p.q_hs, p.q_rs);
    do_srd = make_srd((const E*)p.dout + do_boff + q_row0 * p.do_rs + (int64_t)h * p.do_hs, p.do_rs);
    const int64_t base = p.cu_q ? ((int64_t)h * p.total_q + q_row0) : (((int64_t)b * p.h + h) * p.sq);
    aux_h = (hi ? p.delta : p.lse) + base;
  };
  float aux_reg = 0.f;
  const unsigned wave_dst = (unsigned)(wave * PW * 1024);
  const float rscale = 1.f / p.scale;
  const float cap_c = CAP ? p.softcap * 1.4426950408889634f : 0.f, cap_m2c = -2.f * cap_c;   // (CAP) c = softcap*log2e: the capped scores live in [-c, c] log2 units
  // Request of tile t2 into ring slot `slot` in three parts: the scalars (before phase A), the two DMA statements (in phase A's gaps 1 and 3; a tile past the
  // last one is zero-filled through an empty descriptor), and -- after phase A -- wave 0's load of the tile's LSE / delta rows + the stream's advance.
  struct Strm { u32x4 srd_q, srd_d; unsigned toff_q, toff_d, dst_q, dst_d; int m0; bool real; };
  auto stream_prep = [&](int slot) __attribute__((always_inline)) {
    Strm z;
    z.real = t2 < n_steps;
    z.m0 = s_mt * TQ;
    z.toff_q = (unsigned)z.m0 * (unsigned)p.q_rs * 2u;
    z.toff_d = (unsigned)z.m0 * (unsigned)p.do_rs * 2u;
    z.srd_q = q_srd; z.srd_d = do_srd;
    if (!z.real) { z.srd_q[2] = 0u; z.srd_d[2] = 0u; }
    z.dst_q = (unsigned)(slot * SLOT) + wave_dst;
    z.dst_d = z.dst_q + QT;
    return z;
  };
  auto dma_pair = [&](const unsigned (&vo)[PW], unsigned dst, const u32x4& srd, unsigned toff) __attribute__((always_inline)) {
    unsigned keep;
    if constexpr (PW == 2)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                   "buffer_load_dwordx4 %1, %4, %5 offen lds\n\t"
                   "buffer_load_dwordx4 %2, %4, %5 offen offset:1024 lds\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(vo[0]), "v"(vo[PW - 1]), "s"(dst), "s"(srd), "s"(toff) : "memory");
    else
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "buffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(vo[0]), "s"(dst), "s"(srd), "s"(toff) : "memory");
  };
  auto stream_dma_q = [&](const Strm& z) __attribute__((always_inline)) { dma_pair(qo, z.dst_q, z.srd_q, z.toff_q); };
  auto stream_dma_d = [&](const Strm& z) __attribute__((always_inline)) { dma_pair(doo, z.dst_d, z.srd_d, z.toff_d); };
  // -LSE/softmax_scale (lanes 0-31) and -delta (lanes 32-63) of the tile's rows.  EVERY wave loads the 64 values (row clamped into the sequence, the value of a
  // row past the end replaced: -inf => P = 0) and later stores the same 64 words: no wave-dependent branch in the step.
  auto stream_aux = [&](const Strm& z) __attribute__((always_inline)) {
    const int r = z.m0 + ki;
    const int lim = z.real ? sq : 0;
    float x = aux_h[max(min(r, sq - 1), 0)];
    const bool ok = r < lim;
    if constexpr (ALIBI) x = hi ? x : __builtin_fmaf(s_slope, kif, x);   // LSE + slope * (row - q0)
    if constexpr (CAP) aux_reg = hi ? (ok ? -x : 0.f) : (ok ? __builtin_fmaf(x, -1.4426950408889634f, cap_c) : -INFINITY);   // c - LSE*log2e: the exponent's row part
    else aux_reg = hi ? (ok ? -x : 0.f) : (ok ? -x * rscale : -INFINITY);
  };
  auto stream_advance = [&]() __attribute__((always_inline)) {
    ++t2;
    s_mt += mt_dir;
    if (__builtin_expect(--s_left == 0, 0)) { s_left = nm; s_mt = mt_first; ++s_h; if (t2 < n_steps) stream_head(); }
  };
  auto stream_issue = [&](int slot) __attribute__((always_inline)) {   // all of it at once (prologue, idle steps)
    const Strm z = stream_prep(slot);
    stream_aux(z);
    stream_dma_q(z);
    stream_dma_d(z);
    stream_advance();
  };
  auto stream_store_aux = [&](int slot) __attribute__((always_inline)) {
    *(float FA_LDS*)(lds + OFF_AUX + slot * AUX_SLOT + lane * 4) = aux_reg;
  };

  // ---- per-lane LDS read offsets (fa_bwd.hip): row fragment of k-step ks of row ki = k0 ^ (ks << 5); transposed operand block db = tr_base[s] ^ (db << 6)
  const int rswz = swz16<D>(ki);
  const int k0 = ki * ROW_BYTES + ((hi ^ rswz) << 4);
  const int kv0 = k0 + OFF_V + wave * 64 * ROW_BYTES;
  const int tr_i = lane & 15, tr_half = (lane >> 4) & 1, tr_rr = tr_i >> 2, tr_cc = tr_i & 3;
  int tr_base[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const int row = 8 * s2 + 4 * hi + tr_rr;
    tr_base[s2] = tile_off<D>(row, 2 * tr_half + (tr_cc >> 1)) + (tr_cc & 1) * 8;
  }
  const int aux_lane = OFF_AUX + 16 * hi;   // rows 4*hi .. of a group of eight
  const int kx_lane = OFF_KX + wave * KL * 1024 + lane * 16;
  auto opaque = [](int x) __attribute__((always_inline)) { asm volatile("" : "+v"(x)); return x; };
  // (DS) the hand-over: descriptor of this launch's dS slot, this lane's 16-byte slot inside a half image, and the sub-tile index (2 KB each) of this wave's first
  // key sub-tile in row 0 of the group's first query head.  Sub-tile (head g of the group, query row block i, key block k) lives at
  // ds_unit0 + g * ds_head_tiles + ds_row_start(i) + k (fa_device.h): rows are packed -- under a right-bounded mask row i holds only the key blocks it can see.
  u32x4 ds_srd = {0u, 0u, 0u, 0x00020000u};
  unsigned ds_voff = 0u, ds_unit0 = 0u, ds_prev_off = 0u;
  if constexpr (DS) {
    const unsigned long long a = (unsigned long long)p.ds_ws;
    ds_srd[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    ds_srd[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu;
    ds_srd[2] = __builtin_amdgcn_readfirstlane(p.c5_slot_bytes);
    ds_voff = (unsigned)ds_slot(ki, hi) * 16u;
    const int xcd = bid & 7, jr = (bid >> 3) / p.k_unit_size - p.c5_pj0;
    ds_unit0 = __builtin_amdgcn_readfirstlane((unsigned)((jr * 8 + xcd) * p.hk_ratio) * (unsigned)p.ds_head_tiles + (unsigned)(wk0 >> 5));
  }

  // dV^T / dK^T tiles: dV (kb, db) = tile kb*DB + db, dK (kb, db) = tile 2*DB + kb*DB + db; zeroed by the matrix pipe
  {
    u32x4 zf = {0u, 0u, 0u, 0u};
    asm volatile("" : "+v"(zf));
    acc_zero_tuples_mfma<0>(zf, std::make_integer_sequence<int, 4 * DB>{});
  }
  // (softcap: y = score * scale/softcap * 2*log2e; the quotient comes out of the vector ALU: back into a scalar register, said uniform)
  const float cs = CAP ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.scale * 2.885390081777927f / p.softcap))) : p.scale_log2;

  // ---- prologue: tiles 0 and 1 --------------------------------------------------------------------------------------------
  if (n_steps > 0) stream_head();
  stream_issue(0);
  stream_store_aux(0);
  stream_issue(1);
  stream_store_aux(1);
  lds_dma_wait_all();
  __syncthreads();

  // Tiles this wave takes part in, and the ones among them it sees completely (no mask), as tile-index ranges [lo, lo + span]: a tile is tested with one
  // subtraction and one unsigned compare.  Rows past the sequence end need no mask (their LSE is +inf: P = 0).
  int ma_lo = m_lo, ma_hi = m_lo + nm - 1;
  if (wk0 >= sk) ma_hi = ma_lo - 1;
  if (p.wr >= 0) ma_lo = max(ma_lo, (wk0 - 31 - shift - p.wr + 31) >> 5);     // some key <= some row + shift + wr
  if (p.wl >= 0) ma_hi = min(ma_hi, (wk_hi - shift + p.wl) >> 5);             // some key >= some row + shift - wl
  int mf_lo = ma_lo, mf_hi = ma_hi;
  if (wk_last >= sk) mf_hi = mf_lo - 1;                                        // keys past the end: always masked
  if (p.wr >= 0) mf_lo = max(mf_lo, (wk_hi - shift - p.wr + 31) >> 5);        // every key <= every row + shift + wr
  if (p.wl >= 0) mf_hi = min(mf_hi, (wk0 - 31 - shift + p.wl) >> 5);          // every key >= every row + shift - wl
  const bool a_any = ma_hi >= ma_lo, f_any = mf_hi >= mf_lo;
  const int a_lo = __builtin_amdgcn_readfirstlane(a_any ? ma_lo : 0x3fffffff), f_lo = __builtin_amdgcn_readfirstlane(f_any ? mf_lo : 0x3fffffff);
  const unsigned a_span = __builtin_amdgcn_readfirstlane((unsigned)(a_any ? ma_hi - ma_lo : 0)), f_span = __builtin_amdgcn_readfirstlane((unsigned)(f_any ? mf_hi - mf_lo : 0));

  f32x16 s[KB], dp[KB];                           // S / dP of the tile in flight (both chains start from a C operand loaded by preload_a)
  u32x4 pc[KB][2], dc[KB][2], pn[KB][2], dn[KB][2];   // P / dS (B operands: [kb][16-query half]): pc / dc of the previous tile (read by phase B), pn / dn of the
                                                          // current one (written by SM); pn -> pc in the idle vector slots of the next phase A -- one copy of the step's code
  constexpr int AHJ = FA_DKDV64_AHJ, AHT = FA_DKDV64_AHT;
  constexpr int NR = AHJ + 3, NT = AHT + 1;       // fragment rings of phase A / phase B (see the gap loops): the same registers
  static_assert(NT <= NR, "phase B's ring lives in phase A's");
  u32x4 ring[NR];
  u32x4 kx[2];   // key block 1's K fragment of the k-step in flight and of the next one (KL > 0)
  float slope_v = 0.f;   // (ALIBI) slope*log2e of the head of the tile in flight (uniform; loaded with the chains' C operands)

  // phase A fragment stream: j = 4*ks + kind, kind 0 = Q row fragment (S chains), 1 = dO row fragment (dP chains), 2 / 3 = V fragment of key block 0 / 1
  auto rd_a = [&](auto jc, int qa, int kva) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value, ks = j >> 2, kind = j & 3;
    if constexpr (kind == 0) ring[j % NR] = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)(qa ^ (ks << 5));
    else if constexpr (kind == 1) ring[j % NR] = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)((qa ^ (ks << 5)) + QT);
    else if constexpr (kind == 2) ring[j % NR] = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)(kva ^ (ks << 5));
    else ring[j % NR] = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)((kva ^ (ks << 5)) + 32 * ROW_BYTES);
  };
  // What a step's phase A needs before its first MFMA: the chains' C operands -- -LSE/scale of the tile's rows for S (so that the scores leave the pipe as
  // S - LSE/scale and the exponent of P is ONE multiply by scale*log2e away: (S - LSE/scale) * scale * log2e = S*scale*log2e - LSE*log2e, the scale applied
  // in fp32), -delta for dP, the same sixteen rows for both key blocks -- and the first AHJ fragments.  Requested at the end of the step before (the tile
  // was published one barrier earlier).
  auto preload_a = [&](int slot, int hq) __attribute__((always_inline)) {
    if constexpr (ALIBI) slope_v = *(const float FA_LDS*)(unsigned long)(unsigned)opaque(OFF_TAB + 4 * hq);
    {
      const int auxp = opaque(aux_lane) + slot * AUX_SLOT;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 d4 = *(const f32x4 FA_LDS*)(unsigned long)(unsigned)(auxp + 32 * g + TQ * 4);
        if constexpr (!CAP) {
          const f32x4 l4 = *(const f32x4 FA_LDS*)(unsigned long)(unsigned)(auxp + 32 * g);
#pragma unroll
          for (int j = 0; j < 4; ++j) s[0][4 * g + j] = l4[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) dp[0][4 * g + j] = d4[j];
      }
    }
    const int qa = opaque(k0) + slot * SLOT, kva = opaque(kv0);
    static_for<AHJ>([&](auto jc) __attribute__((always_inline)) { rd_a(jc, qa, kva); });
    if constexpr (KL > 0) kx[(KS - KL) & 1] = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)opaque(kx_lane);
  };

  // ---- phase A: S[kb] = Q.K[kb]^T - LSE/scale, dP[kb] = dO.V[kb]^T - delta, 4*KS MFMA gaps; the stream's DMA statements ride in gaps 1 and 3 -----------------------
  auto phase_a = [&](int slot, auto&& in_gap) __attribute__((always_inline)) {
    const int qa = opaque(k0) + slot * SLOT, kva = opaque(kv0), kxa = opaque(kx_lane);
    (void)kxa;
    __builtin_amdgcn_sched_barrier(0);
    static_for<4 * KS>([&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value, ks = g >> 2, kind = g & 3;
      if constexpr (g + AHJ < 4 * KS) rd_a(ICw<g + AHJ>{}, qa, kva);
      // one wait per k-step: its four fragments (and key block 1's LDS-resident K fragment, requested between them) have landed; the AHJ - 3 fragments
      // requested behind them may stay in flight.  hipcc models the explicit wait and drops its own in front of the k-step's other three MFMAs.
      if constexpr (kind == 0 && ks >= 1 && AHJ >= 3) __builtin_amdgcn_s_waitcnt(0xC07F | ((g + AHJ < 4 * KS ? AHJ - 3 : (4 * KS - 4 - g > 0 ? 4 * KS - 4 - g : 0)) << 8));
      // k-step 0: key block 1's chain goes FIRST and takes key block 0's preloaded tuple as its C operand (one copy of -LSE/scale and -delta is loaded per step,
      // not two); block 0's chain then accumulates onto it in place.  The matrix pipe is in order: the second MFMA's write follows the first one's read.
      if constexpr (kind == 0) {
        if constexpr (ks == 0 && CAP) kv_mfma_c0<E>(s[1], ring[0], kf1[0]);
        else if constexpr (ks == 0) kv_mfma_from<E>(s[1], ring[0], kf1[0], s[0]);
        else kv_mfma_acc<E>(s[0], ring[(4 * ks) % NR], kf0[ks]);
      } else if constexpr (kind == 1) {
        if constexpr (ks == 0 && CAP) kv_mfma_c0<E>(s[0], ring[0], kf0[0]);
        else if constexpr (ks == 0) kv_mfma_acc<E>(s[0], ring[0], kf0[0]);
        else if constexpr (ks < KS - KL) kv_mfma_acc<E>(s[1], ring[(4 * ks) % NR], kf1[ks]);
        else kv_mfma_acc<E>(s[1], ring[(4 * ks) % NR], kx[ks & 1]);
      } else if constexpr (kind == 2) {
        if constexpr (ks == 0) kv_mfma_from<E>(dp[1], ring[1], ring[3], dp[0]);
        else kv_mfma_acc<E>(dp[0], ring[(4 * ks + 1) % NR], ring[(4 * ks + 2) % NR]);
      } else {
        if constexpr (ks == 0) kv_mfma_acc<E>(dp[0], ring[1], ring[2]);
        else kv_mfma_acc<E>(dp[1], ring[(4 * ks + 1) % NR], ring[(4 * ks + 3) % NR]);
      }
      // (the next k-step's LDS-resident K fragment: requested one k-step ahead, its register was last read three gaps ago)
      if constexpr (kind == 0 && ks + 1 >= KS - KL && ks + 1 < KS && ks >= KS - KL)
        kx[(ks + 1) & 1] = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)(kxa + (ks + 1 - (KS - KL)) * 1024);
      // the previous step's P / dS become "current": 32 registers over the phase's gaps (the vector ALU is idle in this phase)
      static_for<32 / (4 * KS)>([&](auto cc) __attribute__((always_inline)) {
        constexpr int c = g * (32 / (4 * KS)) + decltype(cc)::value;
        constexpr int kb_ = c >> 4, t_ = (c >> 3) & 1, pd_ = (c >> 2) & 1, i_ = c & 3;
        unsigned cv = pd_ == 0 ? pn[kb_][t_][i_] : dn[kb_][t_][i_];
        asm volatile("" : "+v"(cv));   // (a real move in THIS gap: the registers of pn / dn are rewritten by this step's SM)
        if constexpr (pd_ == 0) pc[kb_][t_][i_] = cv; else dc[kb_][t_][i_] = cv;
      });
      in_gap(gc);   // the step's tile DMA, LSE / delta load and scalar bookkeeping ride here (the vector and scalar ALUs are idle in this phase)
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // transposed-fragment stream of phase B: F = 4*db + 2*t + src, db = 32-wide block of the head dim, t = 16-query half, src 0 = dO^T -> dV, 1 = Q^T -> dK.
  // db-major, so that the two per-lane addresses a block needs (tr_base[s] ^ (db << 6): the block index enters through XOR on swizzled chunk bits) live for
  // four fragments and are made when the block's first fragment is requested; a tile is accumulated into every fourth MFMA.
  constexpr int NFB = 4 * DB;
  int tx0 = 0, tx1 = 0;   // addresses of the block being requested
  auto rd_t = [&](auto Fc, int t0p, int t1p) __attribute__((always_inline)) {
    constexpr int F = decltype(Fc)::value, db = F >> 2, t = (F >> 1) & 1, src = F & 1;
    constexpr int imm = 16 * t * ROW_BYTES + (src == 0 ? QT : 0);
    if constexpr ((F & 3) == 0) { tx0 = t0p ^ (db << 6); tx1 = t1p ^ (db << 6); }
    const s16x4 lo = lds_read_tr16((const char FA_LDS*)(unsigned long)(unsigned)(tx0 + imm));
    const s16x4 hi4 = lds_read_tr16((const char FA_LDS*)(unsigned long)(unsigned)(tx1 + imm));
    ring[F % NT] = __builtin_bit_cast(u32x4, combine_tr<V8>(lo, hi4));
  };

  auto pack2 = [&](float x0, float x1) __attribute__((always_inline)) {
    using V2 = __attribute__((ext_vector_type(2))) E;
    V2 pr;
    pr[0] = (E)x0;
    pr[1] = (E)x1;
    return __builtin_bit_cast(unsigned, pr);
  };

  // ---- phase B (+ SM): dV / dK of the previous tile (P / dS in pr / dr, transposed operands from ring slot `slot_b`), with the softmax arithmetic of the
  // current tile (-> pw / dw) in its gaps: element e = 16*kb + r (row acc_row(r, hi) of the tile, column = this lane's key of block kb) goes through
  // stage 1 (multiply by scale*log2e, mask, exp2) in gap e and stage 2 (dS = P * (dP - delta), packing of a finished pair) one gap later -- a v_exp's
  // consumer in the same gap costs a wait state.  DO_B / DO_SM switch the halves off (pipeline fill / drain, a wave outside its visible range).
  auto phase_b = [&](auto dobc, auto dosmc, auto maskc, int slot_b, int q0, const u32x4 (&pr)[KB][2], const u32x4 (&dr)[KB][2],
                     u32x4 (&pw)[KB][2], u32x4 (&dw)[KB][2], int slot_sm = 0) __attribute__((always_inline)) {
    constexpr bool DO_B = decltype(dobc)::value != 0, DO_SM = decltype(dosmc)::value != 0, MASK = decltype(maskc)::value != 0;
    static_assert(!CAP || (D == 128 && DO_B), "the softcap variant: head dim 128 (one element per gap), inside the fused phase");
    constexpr int NG = 2 * NFB;   // MFMA gaps (8 * DB)
    constexpr int EPG = 32 / NG;  // elements per gap: 1 at D = 128, 2 at D = 64
    // MASK: bit o of vis[kb] = the lane's key of block kb is visible to tile row 4*hi + o (the rows of the accumulator elements are acc_row(r, 0) above 4*hi);
    // an element is masked with two instructions (sign-extended bit -> bit-field select between the score and -inf), no compare / lane-mask traffic
    unsigned vis[KB] = {0u, 0u};
    float ninf = -INFINITY;
    if constexpr (DO_SM && MASK) {
      asm volatile("" : "+v"(ninf));
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int key = wk0 + 32 * kb + ki;
        int lo = (p.wr >= 0) ? (key - shift - p.wr - q0 - 4 * hi) : 0;       // visible offsets: lo <= o <= hi_
        int hi_ = (p.wl >= 0) ? (key - shift + p.wl - q0 - 4 * hi) : 31;
        lo = max(lo, 0); hi_ = min(hi_, 31);
        const unsigned ones = (hi_ - lo >= 31) ? 0xffffffffu : ((2u << ((hi_ - lo) & 31)) - 1u);
        vis[kb] = (hi_ >= lo && key < sk) ? (ones << (lo & 31)) : 0u;
      }
    }
    if constexpr (DO_SM && !DO_B) { mfma_drain_v(s[0], s[1]); mfma_drain_v(dp[0], dp[1]); }   // (no MFMA segment between the chains and the vector ALU)
    float c_l = cs;
    asm volatile("" : "+s"(c_l));   // (a copy of its own per instantiation: hipcc otherwise hoists all 32 multiplies by scale*log2e -- common to the masked and the
                                    // plain instantiation -- in front of the branch between them: 32 registers)
    float lb[KB] = {0.f, 0.f};   // (ALIBI) slope*log2e * (key - q0 - shift) of this lane's key in block kb
    if constexpr (ALIBI && DO_SM) {
      const float f0 = (float)(wk0 + ki - shift - q0);
      lb[0] = slope_v * f0;
      lb[1] = slope_v * (f0 + 32.f);
    }
    float pv[32], dv[32];   // P and dS of the tile's elements (each lives for a gap or two)
    // (CAP) three stages: A = scale, mask, the mask carrier + the row's c - LSE*log2e, 2^y + 1; B = r = 1/(2^y + 1), the exponent, r - r^2; C = P, dS, packing.
    // The rows' values come four at a time (elements 4g .. 4g+3 of either key block = rows 8g + 4*hi .. + 3), requested one group ahead.
    float cap_a1[32], cap_d[32], cap_arg[32], cap_w[32];
    f32x4 nl4[2];
    const int nl_addr = opaque(aux_lane) + slot_sm * AUX_SLOT;
    auto nl_read = [&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value;
      nl4[g & 1] = *(const f32x4 FA_LDS*)(unsigned long)(unsigned)(nl_addr + 32 * (g & 3));
    };
    auto capA = [&](auto ec) __attribute__((always_inline)) {
      constexpr int e = decltype(ec)::value, kb = e >> 4, r = e & 15;
      float xv = s[kb][r] * c_l;
      if constexpr (MASK) {
        unsigned t;
        asm volatile("v_bfe_i32 %1, %2, %c3, 1\n\tv_bfi_b32 %0, %1, %0, %4" : "+v"(xv), "=&v"(t) : "v"(vis[kb]), "i"(acc_row(r, 0)), "v"(ninf));
      }
      cap_a1[e] = __builtin_fmaf(xv, 7.888609052210118e-31f, nl4[(e >> 2) & 1][e & 3]);   // (2^-100: nothing for a finite score, -inf for a masked one)
      cap_d[e] = fast_exp2(xv) + 1.f;
    };
    auto capB = [&](auto ec) __attribute__((always_inline)) {
      constexpr int e = decltype(ec)::value;
      const float rr = __builtin_amdgcn_rcpf(cap_d[e]);
      cap_arg[e] = __builtin_fmaf(rr, cap_m2c, cap_a1[e]);
      cap_w[e] = __builtin_fmaf(-rr, rr, rr);
    };
    auto capC = [&](auto ec) __attribute__((always_inline)) {
      constexpr int e = decltype(ec)::value, kb = e >> 4, r = e & 15;
      pv[e] = fast_exp2(cap_arg[e]);
      dv[e] = pv[e] * dp[kb][r] * cap_w[e];
      if constexpr ((r & 1) == 1) {
        unsigned pwv = pack2(pv[e - 1], pv[e]), dwv = pack2(dv[e - 1], dv[e]);
        asm volatile("" : "+v"(pwv), "+v"(dwv));   // pinned to this gap
        pw[kb][r >> 3][(r & 7) >> 1] = pwv;
        dw[kb][r >> 3][(r & 7) >> 1] = dwv;
      }
    };
    auto sm1 = [&](auto ec) __attribute__((always_inline)) {
      constexpr int e = decltype(ec)::value, kb = e >> 4, r = e & 15;
      float xv = ALIBI ? __builtin_fmaf(s[kb][r], c_l, lb[kb]) : s[kb][r] * c_l;
      if constexpr (MASK) {
        unsigned t;
        asm volatile("v_bfe_i32 %1, %2, %c3, 1\n\tv_bfi_b32 %0, %1, %0, %4" : "+v"(xv), "=&v"(t) : "v"(vis[kb]), "i"(acc_row(r, 0)), "v"(ninf));
      }
      pv[e] = fast_exp2(xv);
    };
    auto sm2 = [&](auto ec) __attribute__((always_inline)) {
      constexpr int e = decltype(ec)::value, kb = e >> 4, r = e & 15;
      dv[e] = pv[e] * dp[kb][r];
      if constexpr ((r & 1) == 1) {
        unsigned pwv = pack2(pv[e - 1], pv[e]), dwv = pack2(dv[e - 1], dv[e]);
        asm volatile("" : "+v"(pwv), "+v"(dwv));   // pinned to this gap
        pw[kb][r >> 3][(r & 7) >> 1] = pwv;
        dw[kb][r >> 3][(r & 7) >> 1] = dwv;
      }
    };
    if constexpr (DO_B) {
      const int t0p = opaque(tr_base[0]) + slot_b * SLOT, t1p = opaque(tr_base[1]) + slot_b * SLOT;
      if constexpr (CAP && DO_SM) nl_read(ICw<0>{});   // (ahead of the transposed reads: older than every read the counted waits below leave in flight)
      static_for<AHT>([&](auto Fc) __attribute__((always_inline)) { rd_t(Fc, t0p, t1p); });
      __builtin_amdgcn_sched_barrier(0);
      static_for<NG>([&](auto xc) __attribute__((always_inline)) {
        constexpr int x = decltype(xc)::value, F = x >> 1, kb = x & 1;
        constexpr int db = F >> 2, t = (F >> 1) & 1, src = F & 1;
        if constexpr (kb == 0 && F + AHT < NFB) rd_t(ICw<F + AHT>{}, t0p, t1p);
        if constexpr (kb == 0 && (F & 1) == 0 && AHT >= 2) {   // fragments F and F + 1 have landed; the ones behind them (two transpose reads each) may stay in flight
          constexpr int last = F + AHT < NFB ? F + AHT : NFB - 1;
          constexpr int out = last - (F + 1) > 0 ? 2 * (last - (F + 1)) : 0;
          __builtin_amdgcn_s_waitcnt(0xC07F | (out << 8));
        }
        if constexpr (src == 0) kv_mfma_tile<E, kb * DB + db>(ring[F % NT], pr[kb][t]);
        else kv_mfma_tile<E, 2 * DB + kb * DB + db>(ring[F % NT], dr[kb][t]);
        if constexpr (DO_SM && CAP) {
          // (the next group's rows: requested behind this gap's counted wait and its MFMA -- at the next counted wait, four gaps on, every read issued since is newer)
          if constexpr ((x & 3) == 0 && x + 4 < 32) nl_read(ICw<((x + 4) / 4)>{});
          if constexpr (x >= 2) capC(ICw<(x >= 2 ? x - 2 : 0)>{});
          if constexpr (x >= 1) capB(ICw<(x >= 1 ? x - 1 : 0)>{});
          capA(ICw<x>{});
        } else if constexpr (DO_SM) {
          static_for<EPG>([&](auto ic) __attribute__((always_inline)) {
            constexpr int e = x * EPG + decltype(ic)::value;
            sm1(ICw<e>{});
            if constexpr (e >= EPG) sm2(ICw<e - EPG>{});
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (DO_SM && CAP) { capB(ICw<31>{}); capC(ICw<30>{}); capC(ICw<31>{}); }
      else if constexpr (DO_SM) static_for<EPG>([&](auto ic) __attribute__((always_inline)) { sm2(ICw<32 - EPG + decltype(ic)::value>{}); });
    } else if constexpr (DO_SM) {
      static_for<16>([&](auto hc) __attribute__((always_inline)) {   // pair by pair: few values alive at a time
        constexpr int e = 2 * decltype(hc)::value;
        sm1(ICw<e>{}); sm1(ICw<e + 1>{}); sm2(ICw<e>{}); sm2(ICw<e + 1>{});
        __builtin_amdgcn_sched_barrier(0);
      });
    }
  };

  // ---- the step loop ----------------------------------------------------------------------------------------------------------------
  // step st: phase A of tile st, phase B of tile st - 1 with the vector work of tile st; tile st + 2 is requested; one barrier.
  // ---- the step loop ----------------------------------------------------------------------------------------------------------------
  // Control block of a step: everything scalar it needs, made DURING the step before it (inside phase A's gaps, where the scalar ALU idles): the first version
  // made it at the step's head -- ~90 instructions and a dozen branches per 64 MFMAs with nothing to hide behind (profiles/r05_bwd_dkdv_w64.txt).
  struct Ctl { bool act, msk; int slot_a, slot_b, slot2, q0, hq; unsigned ds_off; Strm z; };
  int c_mt = mt_first, c_left = __builtin_amdgcn_readfirstlane(nm), c_h = 0;   // tile st of the walk: tile index, tiles left in its head, head of the group
  bool act_prev = false;
  auto make_ctl = [&](int st) __attribute__((always_inline)) {   // for step st, from the counters as they stand (compute tile st, stream tile st + 2)
    Ctl c;
    c.act = (st < n_steps) && ((unsigned)(c_mt - a_lo) <= a_span);
    c.msk = !((unsigned)(c_mt - f_lo) <= f_span);
    c.slot_a = st & 3; c.slot2 = (st + 2) & 3;
    // (the fill's MFMAs multiply zeros by whatever the "previous tile's" slot holds: at step 0 that slot was never written -- NaN bit patterns times zero --,
    // so the fill reads the current tile's slot instead)
    c.slot_b = act_prev ? ((st - 1) & 3) : c.slot_a;
    c.q0 = c_mt * TQ;
    c.hq = c_h;
    c.ds_off = 0u;
    if constexpr (DS) c.ds_off = (ds_unit0 + (unsigned)c_h * (unsigned)p.ds_head_tiles + (unsigned)ds_row_start(c_mt, p.ds_c1, p.ds_jb, p.ds_nk32)) << 11;
    c.z = stream_prep(c.slot2);
    return c;
  };
  Ctl nxt = make_ctl(0);
  auto advance = [&](const Ctl& c, int st) __attribute__((always_inline)) {   // counters -> step st + 1, then its control block
    act_prev = c.act;
    const bool wrap = c_left == 1;
    c_mt = wrap ? mt_first : c_mt + mt_dir;
    c_left = wrap ? nm : c_left - 1;
    if constexpr (ALIBI || DS) c_h = wrap ? c_h + 1 : c_h;
    stream_advance();
    nxt = make_ctl(st + 1);
  };
  preload_a(0, 0);
  auto ds_store_piece = [&](auto ic, const u32x4& srd, unsigned soff) __attribute__((always_inline)) {   // (DS) fragment (key block i >> 1, query half i & 1) of the previous tile's dS
    constexpr int i = decltype(ic)::value;
    const u32x4 x = dn[i >> 1][i & 1];
    const unsigned vo = ds_voff;   // (copies: clang does not capture a variable a generic lambda names only as an asm operand)
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen offset:%c4" : : "v"(x), "v"(vo), "s"(srd), "s"(soff), "i"((i >> 1) * 2048 + (i & 1) * 1024) : "memory");
  };
  auto step = [&](int st) __attribute__((always_inline)) {
    const Ctl c = nxt;
    // (DS) the dS of the PREVIOUS tile -- still in dn: this step's phase B rewrites it -- leaves in four stores; a step without a previous tile stores through a
    // descriptor of range zero (dropped by the hardware)
    u32x4 st_srd = ds_srd;
    const unsigned st_off = ds_prev_off;
    if constexpr (DS) { if (c.slot_b == c.slot_a) st_srd[2] = 0u; ds_prev_off = c.ds_off; }
    auto ds_store = [&](auto ic) __attribute__((always_inline)) { ds_store_piece(ic, st_srd, st_off); };
    stream_aux(c.z);
    if (__builtin_expect(c.act, 1)) {
      phase_a(c.slot_a, [&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g == 1) stream_dma_q(c.z);
        if constexpr (g == 3) stream_dma_d(c.z);
        if constexpr (DS && g >= 5 && g <= 11 && (g & 1) == 1) ds_store(ICw<(g - 5) / 2>{});
      });
    } else {
      stream_dma_q(c.z); stream_dma_d(c.z);
      if constexpr (DS) static_for<4>(ds_store);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int t = 0; t < 2; ++t) { pc[kb][t] = pn[kb][t]; dc[kb][t] = dn[kb][t]; }
    }
    // ONE body for every step that has work: the fused phase B + SM.  Pipeline fill (no previous tile): its P / dS are zeroed first, the MFMAs add zeros;
    // drain (no current tile): the vector half chews on stale S / dP and its output is never read.  (Separate fill / drain variants were four more copies of
    // the step whose register live ranges joined the hot ones: ~90 spilled registers.)
    using Y = ICw<1>; using N = ICw<0>;
    const bool had_prev = c.slot_b != c.slot_a;   // (= act_prev at the time the block was made)
    if (__builtin_expect(c.act || had_prev, 1)) {
      if (__builtin_expect(!had_prev, 0)) {   // (asm: as plain assignments hipcc turns the branch into 32 selects on the hot path)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              unsigned z0, z1;
              asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0" : "=v"(z0), "=v"(z1));
              pc[kb][t][i] = z0; dc[kb][t][i] = z1;
            }
      }
      if (__builtin_expect(c.act && c.msk, 0)) phase_b(Y{}, Y{}, Y{}, c.slot_b, c.q0, pc, dc, pn, dn, c.slot_a);
      else phase_b(Y{}, Y{}, N{}, c.slot_b, c.q0, pc, dc, pn, dn, c.slot_a);
    }
    // The step's tail, in THIS order (round 6; everything here is exposed -- one wave per SIMD, no MFMA to hide behind): the DMA wait and the aux store, then the next
    // step's operand requests, and only then the ~30 scalar instructions that make the next step's control block -- they run while the requests are in flight; the
    // barrier's own LDS wait (hipcc drains lgkmcnt in front of every s_barrier) then finds them landed.  (Round 5 had the control block first and the requests right in
    // front of the barrier: their whole LDS round trip was waited out there, 17 reads per step.)
    lds_dma_wait_all();        // tile st + 2 has landed (requested in this step's first gaps)
    stream_store_aux(c.slot2);
    __builtin_amdgcn_sched_barrier(0);
    const int hq_n = (ALIBI || DS) ? (c_left == 1 ? c_h + 1 : c_h) : 0;   // (= nxt.hq once advance() has run)
    preload_a((st + 1) & 3, hq_n);   // (tile st + 1 was published by the previous barrier.  Unconditional: a conditional definition keeps the 96 registers of
                               // S / dP / the fragment rings alive across phase B in hipcc's eyes; past the last tile it reads a stale slot nobody uses)
    __builtin_amdgcn_sched_barrier(0);
    advance(c, st);            // (not inside phase A's gaps: there the ~70 scalar instructions cost more -- 1994 -> 2329 us, profiles/r05_bwd_dkdv_w64.txt)
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  };
#pragma unroll 1
  for (int st = 0; st <= n_steps; ++st) step(st);

  // ---- epilogue: dK = scale * acc, dV = acc, through the freed ring (whole-row stores); every key row of the block that exists is written, zeros included
  // (empty-sequence contract of the CK tests)
  char FA_LDS* stage = lds + wave * 32 * (ROW_BYTES + 16);
  mfma_drain_acc();
  static_for<KB>([&](auto kbc) __attribute__((always_inline)) {
    constexpr int kb = decltype(kbc)::value;
    const int kb0 = wk0 + 32 * kb;
    if (kb0 < sk) {
      E* dktile = (E*)p.dk + dk_boff + (k_row0 + kb0) * p.dk_rs + (int64_t)hk * p.dk_hs;
      E* dvtile = (E*)p.dv + dv_boff + (k_row0 + kb0) * p.dv_rs + (int64_t)hk * p.dv_hs;
      f32x16 tt[DB];
      static_for<DB>([&](auto dbc) __attribute__((always_inline)) { acc_read_tuple<16 * (2 * DB + kb * DB + decltype(dbc)::value)>(tt[decltype(dbc)::value]); });
      store_tile_via_lds<E, D>(stage, tt, CAP ? 4.f * p.scale : p.scale, dktile, p.dk_rs, sk - kb0, lane);
      static_for<DB>([&](auto dbc) __attribute__((always_inline)) { acc_read_tuple<16 * (kb * DB + decltype(dbc)::value)>(tt[decltype(dbc)::value]); });
      store_tile_via_lds<E, D>(stage, tt, 1.f, dvtile, p.dv_rs, sk - kb0, lane);
    }
  });
}

#if FA_DKDV64_PART != 2
template <typename E, int D, int FEAT>
__global__ void __launch_bounds__(256, 1) fa_bwd_dkdv_w64_kernel(const BwdK p) {
  dkdv_w64_body<E, D, FEAT, false>(p, blockIdx.x);
}
#endif

#if FA_DKDV64_PART != 1
// ------------------------------------------------------------------------------------------------------------------------------------------------
// dQ = dS . K from the dS the dK/dV items handed over: the fifth contraction of the 5-contraction backward (reference: the dQ accumulation of
// compute_dq_dk_dv_1colblock, csrc/flash_attn/src/flash_bwd_kernel.h:682-724, which reads dS back transposed from shared memory the same way).
// ------------------------------------------------------------------------------------------------------------------------------------------------
// One item = one 256-row query block of one head: 4 waves x 64 rows (two 32-row blocks per wave, so that every K^T fragment feeds two MFMAs),
//   dQ^T[d][query] += K^T[d][key] . dS^T[key][query]      A = K^T (ds_read_b64_tr_b16 of the K tile), B = dS^T (ds_read_b64_tr_b16 of the writer's image)
// over 64-key tiles in a ring of three LDS slots (K tile + each wave's four sub-tile images, all by LDS-DMA).  No softmax, no mask arithmetic: a sub-tile either
// was written (fa_device.h ds_tile_active) and is multiplied, or is skipped.  Bound: LDS -- 48 transposed reads per 32 MFMAs and wave, 96 KB read + 48 KB
// written per tile and workgroup.
template <typename E, int D>
static __device__ __forceinline__ void dq_ds_w64_body(const BwdK p, const int cbid) {
  using T = ElemTraits<E>;
  using V8 = typename T::v8;
  constexpr int NW = 4, QB = 2, BM = NW * 64, BN = 64, CPR = D / 8;
  constexpr int ROW_BYTES = D * 2, TILE_K = BN * ROW_BYTES, DB = D / 32;
  constexpr int DS_WAVE = QB * 2 * 2048;            // a wave's four sub-tile images of a tile: [query block][key sub-tile][query half][1 KiB]
  constexpr int SLOT = TILE_K + NW * DS_WAVE, RING = 3;
  constexpr int RPD = 1024 / ROW_BYTES, DPW = TILE_K / 1024 / NW;   // K: rows per 1-KiB DMA piece, pieces per wave
  static_assert(DPW == 4 || DPW == 2, "K pieces per wave");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char FA_LDS* lds = (char FA_LDS*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;

  // item -> (unit, head of the group, query block): the rounds of eight units the dK/dV items were dealt in, heaviest query block first under a right-bounded mask
  const int xcd = cbid & 7, slot_i = cbid >> 3;
  const int per_unit = p.hk_ratio * p.nmb;
  const int jr = slot_i / per_unit, within = slot_i - jr * per_unit;
  const int hg = within / p.nmb, mbr = within - hg * p.nmb;
  const int j = p.c5_cj0 + jr;
  int unit;
  if (p.k_unit_hpx > 0) { const int bb = j / p.k_unit_hpx; unit = bb * (8 * p.k_unit_hpx) + xcd * p.k_unit_hpx + (j - bb * p.k_unit_hpx); }
  else unit = xcd + 8 * j;
  if (unit >= p.k_units) return;
  const int b = unit / p.h_k, hk = unit - b * p.h_k;
  const int h = hk * p.hk_ratio + hg;
  const int m_block = (p.wr >= 0) ? (p.nmb - 1 - mbr) : mbr;
  const int sq = p.sq, sk = p.sk, shift = sk - sq;
  const int m0 = m_block * BM;
  if (m0 >= sq) return;
  const int blk_last = min(m0 + BM, sq) - 1;
  const int kmax = (p.wr >= 0) ? min(sk - 1, blk_last + shift + p.wr) : sk - 1;
  const int n_max = kmax >= 0 ? kmax / BN + 1 : 0;   // tiles 0 .. n_max - 1 (no left window on this path)
  const int w_row0 = m0 + wave * 64;

  // ---- sources: K rows through a descriptor of the (batch, kv head)'s rows (rows past the last key read as zeros), the dS images through one of the slot
  const E* __restrict__ kp = (const E*)p.k + (int64_t)b * p.k_bs + (int64_t)hk * p.k_hs;
  auto make_srd = [&](const void* base, unsigned long long bytes) __attribute__((always_inline)) {
    const unsigned long long a = (unsigned long long)base;
    u32x4 sd = {(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a), (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu,
                (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(bytes > 0xffffffffull ? 0xffffffffull : bytes)), 0x00020000u};
    return sd;
  };
  const u32x4 k_srd = make_srd(kp, ((unsigned long long)(sk - 1) * (unsigned long long)p.k_rs + D) * 2ull);
  const u32x4 ds_srd = make_srd(p.ds_rd, p.c5_slot_bytes);
  unsigned koff_l[DPW];
  {
    const int d_row = lane / CPR, d_pc = lane % CPR;
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      const int row = (wave * DPW + i) * RPD + d_row;
      const int c = d_pc ^ swz16<D>(row);
      koff_l[i] = (unsigned)(row * (int)p.k_rs + c * 8) * 2u - (unsigned)(i * 1024);
    }
  }
  // this wave's two query row blocks: byte offset of their packed rows in the slot (+ the lane's 16 bytes of a 1-KiB piece); a row block past the last row has no image
  unsigned ds_voff[QB];
  {
    const unsigned head0 = (unsigned)((jr * 8 + xcd) * p.hk_ratio + hg) * (unsigned)p.ds_head_tiles;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const int q32 = min((w_row0 >> 5) + qb, p.ds_nq32 - 1);
      ds_voff[qb] = ((head0 + (unsigned)ds_row_start(q32, p.ds_c1, p.ds_jb, p.ds_nk32)) << 11) + (unsigned)lane * 16u;
    }
  }
  auto load_tile = [&](int n, int st) __attribute__((always_inline)) {
    const unsigned kdst = __builtin_amdgcn_readfirstlane((unsigned)(st * SLOT + wave * DPW * 1024));
    const unsigned ktoff = __builtin_amdgcn_readfirstlane((unsigned)n * (unsigned)(BN * 2) * (unsigned)p.k_rs);
    unsigned keep;
    if constexpr (DPW == 4)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
                   "buffer_load_dwordx4 %1, %6, %7 offen lds\n\t"
                   "buffer_load_dwordx4 %2, %6, %7 offen offset:1024 lds\n\t"
                   "buffer_load_dwordx4 %3, %6, %7 offen offset:2048 lds\n\t"
                   "buffer_load_dwordx4 %4, %6, %7 offen offset:3072 lds\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(koff_l[0]), "v"(koff_l[1]), "v"(koff_l[DPW - 2]), "v"(koff_l[DPW - 1]), "s"(kdst), "s"(k_srd), "s"(ktoff) : "memory");
    else
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                   "buffer_load_dwordx4 %1, %4, %5 offen lds\n\t"
                   "buffer_load_dwordx4 %2, %4, %5 offen offset:1024 lds\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(koff_l[0]), "v"(koff_l[DPW - 1]), "s"(kdst), "s"(k_srd), "s"(ktoff) : "memory");
    const unsigned dtoff = __builtin_amdgcn_readfirstlane((unsigned)n * 4096u);   // key sub-tiles 2n, 2n + 1 of the row
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const unsigned ddst = __builtin_amdgcn_readfirstlane((unsigned)(st * SLOT + TILE_K + wave * DS_WAVE + qb * 4096));
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "buffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
                   "buffer_load_dwordx4 %1, %3, %4 offen offset:1024 lds\n\t"
                   "buffer_load_dwordx4 %1, %3, %4 offen offset:2048 lds\n\t"
                   "buffer_load_dwordx4 %1, %3, %4 offen offset:3072 lds\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(ds_voff[qb]), "s"(ddst), "s"(ds_srd), "s"(dtoff) : "memory");
    }
  };
  constexpr int PER_TILE = DPW + 4 * QB;   // DMA instructions per wave and tile: the same for every wave and tile, so "tile n has landed" is a literal vmcnt
  auto wait_tile = [&](bool next_in_flight) __attribute__((always_inline)) {
    if (next_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  // ---- per-lane LDS read offsets: transposed K fragments (fa_bwd_w64.hip), transposed dS fragments (fa_bwd.hip fa_bwd_dq_from_ds: reader half hi takes keys 4*hi .. + 3 and 8 + 4*hi ..)
  const int tr_i = lane & 15, tr_half = (lane >> 4) & 1, tr_rr = tr_i >> 2, tr_cc = tr_i & 3;
  int tr_base[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) tr_base[s2] = tile_off<D>(8 * s2 + 4 * hi + tr_rr, 2 * tr_half + (tr_cc >> 1)) + (tr_cc & 1) * 8;
  const int ds_lane = TILE_K + wave * DS_WAVE + tr_half * 1024 + hi * 128 + (tr_cc & 1) * 64 + tr_rr * 16 + (tr_cc >> 1) * 8;

  // dQ^T accumulators: tile (qb, db) = accumulator registers a[16 * (qb*DB + db) .. + 15], named in the asm like the dK/dV part's (as C++ values hipcc gave the three
  // ring positions three register assignments and copied all 128 registers at every tile: 128 v_accvgpr_mov per 32 MFMAs); zeroed by the matrix pipe
  {
    u32x4 zf = {0u, 0u, 0u, 0u};
    asm volatile("" : "+v"(zf));
    acc_zero_tuples_mfma<0>(zf, std::make_integer_sequence<int, QB * DB>{});
  }
  auto rd_tr = [&](int a0, int a1) __attribute__((always_inline)) {
    const s16x4 lo = lds_read_tr16(lds + a0), hi4 = lds_read_tr16(lds + a1);
    return __builtin_bit_cast(u32x4, combine_tr<V8>(lo, hi4));
  };
  // tiles 0 .. n_all - 1 are complete for this wave (all four sub-tiles written: no mask edge, every row and key exists); the rest is tested sub-tile by sub-tile
  int n_all = n_max;
  if (w_row0 + 64 > sq) n_all = 0;
  n_all = min(n_all, sk / BN);
  if (p.wr >= 0) n_all = min(n_all, (w_row0 + shift + p.wr - 1) / BN + ((w_row0 + shift + p.wr - 1) >= 0 ? 1 : 0));   // 64 n + 32 <= w_row0 + 31 + shift + wr
  n_all = __builtin_amdgcn_readfirstlane(max(n_all, 0));
  if (n_max > 0) load_tile(0, 0);
  if (n_max > 1) load_tile(1, 1);
  auto tile = [&](auto curc, int n) __attribute__((always_inline)) {
    constexpr int cur = decltype(curc)::value;
    constexpr int KOFF = cur * SLOT, DOFF = cur * SLOT;
    wait_tile(n + 1 < n_max);
    __syncthreads();   // tile n is in LDS for every wave, and every wave is done with tile n - 1, whose slot the next request overwrites
    if (n + 2 < n_max) load_tile(n + 2, (cur + 2) % RING);
    if (__builtin_expect(n < n_all, 1)) {
      static_for<4>([&](auto ksc) __attribute__((always_inline)) {   // k-step = 16 keys: key sub-tile kb = ks >> 1, half t = ks & 1
        constexpr int ks = decltype(ksc)::value, kb = ks >> 1, t = ks & 1;
        u32x4 bf[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          const int a = DOFF + ds_lane + qb * 4096 + kb * 2048 + t * 512;
          bf[qb] = rd_tr(a, a + 256);
        }
        static_for<DB>([&](auto dbc) __attribute__((always_inline)) {
          constexpr int db = decltype(dbc)::value;
          constexpr int kbase = KOFF + kb * 32 * ROW_BYTES + 16 * t * ROW_BYTES;
          const u32x4 af = rd_tr(kbase + (tr_base[0] ^ (db << 6)), kbase + (tr_base[1] ^ (db << 6)));
          kv_mfma_tile<E, db>(af, bf[0]);
          kv_mfma_tile<E, DB + db>(af, bf[1]);
        });
      });
    } else {
      static_for<4>([&](auto ksc) __attribute__((always_inline)) {
        constexpr int ks = decltype(ksc)::value, kb = ks >> 1, t = ks & 1;
        const bool a0 = ds_tile_active(w_row0, n * BN + 32 * kb, sq, sk, shift, p.wl, p.wr), a1 = ds_tile_active(w_row0 + 32, n * BN + 32 * kb, sq, sk, shift, p.wl, p.wr);
        if (a0 || a1) {
          u32x4 bf[QB];
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) {
            const int a = DOFF + ds_lane + qb * 4096 + kb * 2048 + t * 512;
            bf[qb] = rd_tr(a, a + 256);
          }
          static_for<DB>([&](auto dbc) __attribute__((always_inline)) {
            constexpr int db = decltype(dbc)::value;
            constexpr int kbase = KOFF + kb * 32 * ROW_BYTES + 16 * t * ROW_BYTES;
            const u32x4 af = rd_tr(kbase + (tr_base[0] ^ (db << 6)), kbase + (tr_base[1] ^ (db << 6)));
            if (a0) kv_mfma_tile<E, db>(af, bf[0]);
            if (a1) kv_mfma_tile<E, DB + db>(af, bf[1]);
          });
        }
      });
    }
  };
  for (int n = 0; n < n_max; n += RING) {
    tile(ICw<0>{}, n);
    if (n + 1 < n_max) tile(ICw<1>{}, n + 1);
    if (n + 2 < n_max) tile(ICw<2>{}, n + 2);
  }
  __syncthreads();   // every wave is done with the ring: the staging below reuses it
  mfma_drain_acc();
  E* dqtile = (E*)p.dq + (int64_t)b * p.dq_bs + (int64_t)w_row0 * p.dq_rs + (int64_t)h * p.dq_hs;
  static_for<QB>([&](auto qbc) __attribute__((always_inline)) {
    constexpr int qb = decltype(qbc)::value;
    const int row0 = w_row0 + 32 * qb;
    f32x16 tt[DB];
    static_for<DB>([&](auto dbc) __attribute__((always_inline)) { acc_read_tuple<16 * (qb * DB + decltype(dbc)::value)>(tt[decltype(dbc)::value]); });
    if (row0 < sq)
      store_tile_via_lds<E, D>(lds + (wave * 64 + qb * 32) * (ROW_BYTES + 16), tt, p.scale, dqtile + (int64_t)(32 * qb) * p.dq_rs, p.dq_rs, sq - row0, lane);
  });
}

// The mixed launch of the 5-contraction backward: workgroups 0 .. c5_np - 1 are dK/dV items (they are dealt first: the heavy ones), the rest dQ items of the chunk
// before -- small, and independent of everything else in the launch: they fill the CUs the dK/dV items leave as they finish.
template <typename E, int D>
__global__ void __launch_bounds__(256, 1) fa_bwd_c5_kernel(const BwdK p) {
  // (dispatch order = grid order: the dK/dV items first.  Interleaving the two lists -- n dK/dV items, one dQ item, ... -- was measured and removed: beside the dK/dV
  // items the dQ items' dS stream slows those more than it hides, profiles/r06_bwd_c5.txt)
  const int bid = blockIdx.x;
  if (bid < p.c5_np) dkdv_w64_body<E, D, 0, true>(p, bid + p.c5_pbid0);
  else dq_ds_w64_body<E, D>(p, bid - p.c5_np);
}

template <typename E, int D>
static int launch_c5_t(const BwdK& p, hipStream_t stream) {
  constexpr int dkdv = 4 * 2 * 32 * D * 2 + 256 * D * 2 + 4 * 2 * 32 * 4 + (D == 128 ? 4 * (D / 16 - 1) * 1024 : 0);
  constexpr int dq = 3 * (64 * D * 2 + 4 * 8192), stage = 256 * (D * 2 + 16);
  constexpr int smem = dkdv > dq ? (dkdv > stage ? dkdv : stage) : (dq > stage ? dq : stage);
  static_assert(smem <= 160 * 1024, "LDS");
  auto kern = fa_bwd_c5_kernel<E, D>;
  static std::atomic<unsigned long long> attr_mask{0};
  if (ensure_dyn_lds(attr_mask, (const void*)kern, 160 * 1024, true) != 0) return -1;
  const long long total = (long long)p.c5_np + p.c5_nc;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), smem, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// One launch of the 5-contraction backward (fa_api.cpp: do_bwd_c5 sizes the chunks and the workspace): plain attention, fixed-length batches, head dim 64 / 128, no left
// window, sk >= sq; -2 = not covered.
int launch_bwd_c5(const BwdK& p, int dtype_bf16, int d, hipStream_t stream) {
  if (p.rng != nullptr || p.alibi != nullptr || p.softcap > 0.f || p.d_chunks > 0 || p.cu_q || p.cu_k || p.seqused_q || p.seqused_k || p.k_list) return -2;
  if (p.wl >= 0 || p.sk < p.sq || (d != 128 && d != 64) || (p.c5_np > 0 && !p.ds_ws) || (p.c5_nc > 0 && !p.ds_rd)) return -2;
  const uint64_t span = ((uint64_t)(p.sq > 0 ? p.sq : 1) + 64) * (uint64_t)(p.q_rs > p.do_rs ? p.q_rs : p.do_rs) * 2u;
  const uint64_t kspan = ((uint64_t)p.sk + 128) * (uint64_t)p.k_rs * 2u;
  if (span >= (1ull << 32) || kspan >= (1ull << 32)) return -2;
  if (d == 128) return dtype_bf16 ? launch_c5_t<__bf16, 128>(p, stream) : launch_c5_t<_Float16, 128>(p, stream);
  return dtype_bf16 ? launch_c5_t<__bf16, 64>(p, stream) : launch_c5_t<_Float16, 64>(p, stream);
}
#endif  // FA_DKDV64_PART != 1

#if FA_DKDV64_PART != 2
template <typename E, int D, int FEAT>
static int launch_dkdv_w64_f(const BwdK& p, hipStream_t stream) {
  constexpr bool ALIBI = FEAT == FEAT_ALIBI;
  constexpr int base = 4 * 2 * 32 * D * 2 + 256 * D * 2 + 4 * 2 * 32 * 4 + (D == 128 ? 4 * (D / 16 - 1) * 1024 : 0);   // tile ring | V block | aux ring | K overflow
  const int smem = base + (ALIBI ? (4 * (p.hk_ratio + 1) + 15) / 16 * 16 : 0);                                          // | (ALIBI) slope table
  if (smem > 160 * 1024) return -2;
  auto kern = fa_bwd_dkdv_w64_kernel<E, D, FEAT>;
  static std::atomic<unsigned long long> attr_mask{0};
  if (ensure_dyn_lds(attr_mask, (const void*)kern, 160 * 1024, true) != 0) return -1;
  const long long total = p.k_list ? (long long)p.k_bound * p.h_k : units_grid(p.k_units, p.k_unit_size);
  if (total <= 0) return 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), smem, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
template <typename E, int D>
static int launch_dkdv_w64_t(const BwdK& p, hipStream_t stream) {
  if constexpr (D == 128) { if (p.softcap > 0.f) return launch_dkdv_w64_f<E, D, FEAT_CAP>(p, stream); }
  if (p.softcap > 0.f) return -2;   // (the softcap variant: head dim 128)
  return p.alibi ? launch_dkdv_w64_f<E, D, FEAT_ALIBI>(p, stream) : launch_dkdv_w64_f<E, D, 0>(p, stream);
}

// 4 waves x 64 keys per workgroup (the same 256-key blocks as fa_bwd_dkdv_kernel: grid and work list unchanged).  Plain attention, softcap (head dim 128) or ALiBi under a causal right
// bound, head dim 64 / 128; -2 = not covered, the caller runs fa_bwd_dkdv_kernel.
int launch_bwd_dkdv_w64(const BwdK& p, int dtype_bf16, int d, hipStream_t stream) {
  if (p.rng != nullptr || p.ds_ws != nullptr || p.d_chunks > 0) return -2;
  if (p.softcap > 0.f && p.alibi != nullptr) return -2;
  if (p.alibi != nullptr && p.wr != 0) return -2;   // the bias is linear in the key only where no visible key lies right of the diagonal
  if (d != 128 && d != 64) return -2;
  // buffer addressing of the streamed tiles: 32-bit byte offsets from the head's first row
  const uint64_t span = ((uint64_t)(p.sq > 0 ? p.sq : 1) + 64) * (uint64_t)(p.q_rs > p.do_rs ? p.q_rs : p.do_rs) * 2u;
  if (span >= (1ull << 32)) return -2;
  if (d == 128) return dtype_bf16 ? launch_dkdv_w64_t<__bf16, 128>(p, stream) : launch_dkdv_w64_t<_Float16, 128>(p, stream);
  return dtype_bf16 ? launch_dkdv_w64_t<__bf16, 64>(p, stream) : launch_dkdv_w64_t<_Float16, 64>(p, stream);
}
#endif  // FA_DKDV64_PART != 2

}  // namespace fa
