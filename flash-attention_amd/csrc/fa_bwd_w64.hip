// dQ kernel of the backward pass for gfx950, "64 query rows per wave, one wave per SIMD" schedule (the forward's
// fa_fwd_w64.hip shape applied to fa_bwd.hip's dQ kernel; reference: the dQ half of compute_dq_dk_dv_1colblock,
// csrc/flash_attn/src/flash_bwd_kernel.h:457-724).
//
// A wave owns two 32-row query blocks.  Their Q and dO fragments (B operands, 64 + 64 registers) and the dQ^T accumulators
// (128 registers) fill the accumulator half of the 512-entry register file for the whole block and are named literally in
// the inline asm; scores come out of the matrix pipe in arch VGPRs.  K/V tiles of 64 keys stream through LDS by DMA, every
// fragment read from LDS feeds TWO MFMAs (one per query block):
//   S^T[key][query]  = K . Q^T        A = K row fragment (LDS),     B = Q fragment (AGPR)
//   dP^T[key][query] = V . dO^T       A = V row fragment (LDS),     B = dO fragment (AGPR)
//   dQ^T[d][query]  += K^T . dS^T     A = K^T (LDS transpose read), B = dS^T = P*(dP^T - delta) packed from the VGPR tuples
// column = query = lane, so LSE and delta are lane-local scalars.  Unlike the forward there is no running maximum and no
// rescale: the steady-state step is branch-free.  Step i (32 keys) = 16 MFMAs S_{i+1}, 16 MFMAs dP_{i+1}, 16 MFMAs
// dQ += K_{i-1}^T.dS_{i-1}, with the VALU work of dS_i (fma, exp2, sub, mul, pack: ~5 per element) hand-assigned to the 48
// MFMA gaps.  K is read twice per tile -- row fragments when its scores are computed, transposed one tile later -- so K
// (and, slot for slot, V) sit in a ring of three tiles: tile u+1 arrives while tiles u and u-1 are read.
// Arithmetic as in fa_bwd.hip: P = exp2(S*scale*log2e - LSE*log2e), dS = P*(dP - delta), dS rounded to the input dtype,
// softmax_scale applied once in the epilogue.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "fa_device.h"
#include "fa_kernel_params.h"
#include "fa_launch.h"
#include "fa_fwd_w64_regs.h"
#define FA_W64_CLOB FA_W64_ACC_CLOBBERS_256
#include "fa_w64_asm.h"

// LDS operand reads run this many fragment slots (2 gaps each) ahead of their MFMAs, one explicit wait per two slots (3 costs five spilled registers; the forward
// measured no difference between 1 and 4).  The timing ablations and the -delta-in-the-C-operand variant live in experiments/ablations/fa_bwd_w64.patch (tools/ablate_bw64.sh).
#define FA_BW64_AH 2

namespace fa {
namespace {

constexpr int BW_Q_BASE = 128;    // Q fragments: fragment F at a[128+4F : 131+4F], F = qb*KS + ks
constexpr int BW_DO_BASE = 192;   // dO fragments, same indexing
constexpr float kLog2eW = 1.4426950408889634f;

// d(VGPR) = a(VGPR) . frag(AGPR a[BASE:BASE+3]) + c(VGPR)     first k-step of a chain: c = the lane's -LSE*log2e (score chain) or -delta (dP chain)
// in all sixteen registers, so that the subtraction every element needs is done by the matrix pipe (as the forward's -m, fa_fwd_w64.hip)
template <typename E, int BASE> FA_DEVINL void mfma_v_first(f32x16& d, u32x4 a, const f32x16& c) {
  if constexpr (std::is_same<E, __bf16>::value)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c3:%c4], %2" : "=&v"(d) : "v"(a), "v"(c), "i"(BASE), "i"(BASE + 3) : FA_W64_CLOB);
  else
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%c3:%c4], %2" : "=&v"(d) : "v"(a), "v"(c), "i"(BASE), "i"(BASE + 3) : FA_W64_CLOB);
}
// the same with C = inline constant 0
template <typename E, int BASE> FA_DEVINL void mfma_v_first0(f32x16& d, u32x4 a) {
  if constexpr (std::is_same<E, __bf16>::value)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], 0" : "=&v"(d) : "v"(a), "i"(BASE), "i"(BASE + 3) : FA_W64_CLOB);
  else
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%c2:%c3], 0" : "=&v"(d) : "v"(a), "i"(BASE), "i"(BASE + 3) : FA_W64_CLOB);
}
// d(VGPR) += a(VGPR) . frag(AGPR)
template <typename E, int BASE> FA_DEVINL void mfma_v_acc(f32x16& d, u32x4 a) {
  if constexpr (std::is_same<E, __bf16>::value)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[%c2:%c3], %0" : "+v"(d) : "v"(a), "i"(BASE), "i"(BASE + 3) : FA_W64_CLOB);
  else
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%c2:%c3], %0" : "+v"(d) : "v"(a), "i"(BASE), "i"(BASE + 3) : FA_W64_CLOB);
}
// dQ tuple T (AGPR a[16T:16T+15]) += a(VGPR) . b(VGPR)
template <typename E, int T> FA_DEVINL void mfma_q_acc(u32x4 a, u32x4 b) {
  if constexpr (std::is_same<E, __bf16>::value)
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" : : "v"(a), "v"(b), "i"(16 * T), "i"(16 * T + 15) : FA_W64_CLOB);
  else
    asm volatile("v_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" : : "v"(a), "v"(b), "i"(16 * T), "i"(16 * T + 15) : FA_W64_CLOB);
}

}  // namespace

// ALIBI (under a causal right bound, where the bias -slope * |key - row - shift| is linear in the key): the exponent of P gets slope*log2e * (key - row - shift).
// Element r of a step scores key k0 + 4*hi + acc_row(r, 0): slope*log2e * acc_row(r, 0) rides in the score chains' C operand next to -LSE*log2e (per block),
// slope*log2e * (k0 + 4*hi - row - shift) -- one value per lane, query block and step, from an integer difference -- is added to every score of the step
// in front of its exp2: one more vector instruction per element (32 per 48 MFMAs).
//
// FEAT_CAP (softcap, round 5; reference flash_bwd_kernel.h:588 + utils.h:395-409): Q carries scale/softcap * 2*log2e, the score chains start from C = 0 and deliver
// y = 2*log2e * z (z = score*scale/softcap); with r = 1/(2^y + 1) and c = softcap*log2e: P = 2^(c - LSE*log2e - 2c*r), 1 - tanh^2(z) = 4 * (r - r^2).  Ten vector
// instructions per element instead of three, in three stages a gap apart (none waits for its predecessor); the factor 4 meets softmax_scale in the epilogue.
// FEAT_DROP (dropout; reference flash_bwd_kernel.h + dropout.h): the random stream of fa_device.h drop_bytes -- four consecutive keys of a row share a Philox2x32-7
// call and are four accumulator rows of one lane: eight calls per step and lane, their rounds spread over the gaps ahead of their elements (scalar round keys);
// dS = P * (keep ? dP / (1-p) : 0 - delta).
template <typename E, int D, bool FUSE_DELTA, int FEAT>
__global__ void __launch_bounds__(256, 1) fa_bwd_dq_w64_kernel(const BwdK p) {
  constexpr bool ALIBI = FEAT == FEAT_ALIBI, CAP = FEAT == FEAT_CAP, DROP = FEAT == FEAT_DROP;
  static_assert(FEAT == 0 || ALIBI || CAP || DROP, "feature variants of this schedule: none, causal ALiBi, softcap, dropout");
  using T = ElemTraits<E>;
  using V8 = typename T::v8;
  constexpr int NW = 4, QB = 2, BM = NW * 64, BN = 64, CPR = D / 8;
  constexpr int ROW_BYTES = D * 2, TILE_BYTES = BN * ROW_BYTES;
  constexpr int KS = D / 16, DB = D / 32;
  constexpr int AHK = FA_BW64_AH;
  constexpr int RING = 3;                         // K tiles u-1 (transposed), u (rows), u+1 (arriving); V shares the slot index
  constexpr int V_RING = RING * TILE_BYTES;       // V ring behind the K ring
  constexpr int DO_OFF = BM * ROW_BYTES;          // prologue staging: Q rows at 0, dO rows behind (both under the rings)
  static_assert(D == 128 || D == 64, "head dims of this schedule: 64, 128");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char FA_LDS* lds = (char FA_LDS*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, qi = lane & 31;
  const int d_row = lane / CPR, d_pc = lane % CPR;   // row inside a 1-KiB DMA piece, physical 16-byte chunk
  constexpr int RPD = 1024 / ROW_BYTES;

  int b, h, m_block;
  if (p.q_list) {
    if (!work_list_item(p.q_list, blockIdx.x, p.h, p.h_k, b, h, m_block)) return;
  } else {
    const int w = xcd_interleave(blockIdx.x, p.q_units, p.q_unit_size, p.q_unit_hpx);
    if (w < 0) return;
    const int bh = w / p.nmb;
    const int mbr = w - bh * p.nmb;
    m_block = (p.wr >= 0) ? (p.nmb - 1 - mbr) : mbr;
    b = bh / p.h;
    h = bh - b * p.h;
  }
  const int hk = h / p.hk_ratio;
  int sq = p.sq, sk = p.sk;
  int64_t q_row0 = 0, k_row0 = 0;
  int64_t q_boff = (int64_t)b * p.q_bs, do_boff = (int64_t)b * p.do_bs, dq_boff = (int64_t)b * p.dq_bs;
  int64_t k_boff = (int64_t)b * p.k_bs, v_boff = (int64_t)b * p.v_bs;
  if (p.cu_q) { const int c0 = p.cu_q[b]; sq = p.cu_q[b + 1] - c0; q_row0 = c0; q_boff = 0; do_boff = 0; dq_boff = 0; }
  if (p.cu_k) { const int c0 = p.cu_k[b]; sk = p.cu_k[b + 1] - c0; k_row0 = c0; k_boff = 0; v_boff = 0; }
  if (p.seqused_q) sq = min(sq, p.seqused_q[b]);
  if (p.seqused_k) sk = min(p.seqused_k[b], sk);   // (include/fa_gfx950.h: in the backward seqused_k can only SHORTEN a sequence -- the key-block work list and its bound are sized from cu_seqlens_k)
  const int m0 = m_block * BM;
  if (m0 >= sq) return;

  const E* __restrict__ kp = (const E*)p.k + k_boff + k_row0 * p.k_rs + (int64_t)hk * p.k_hs;
  const E* __restrict__ vp = (const E*)p.v + v_boff + k_row0 * p.v_rs + (int64_t)hk * p.v_hs;

  const int shift = sk - sq;
  const int blk_last = min(m0 + BM, sq) - 1;
  int kmax = sk - 1, kmin = 0;
  if (p.wr >= 0) kmax = min(kmax, blk_last + shift + p.wr);
  if (p.wl >= 0) kmin = max(0, m0 + shift - p.wl);
  const int n_min = kmin / BN;
  const int n_max = (kmax >= kmin) ? (kmax / BN + 1) : n_min;
  const int n_tiles = n_max - n_min;
  const int n_steps = 2 * n_tiles;
  const int key_base = n_min * BN;

  const int w_row0 = m0 + wave * 64;
  const int w_row1 = min(w_row0 + 63, sq - 1);
  const bool wave_valid = w_row0 < sq;
  const int w_kmax = (p.wr >= 0) ? min(sk - 1, w_row1 + shift + p.wr) : sk - 1;
  const int w_kmin = (p.wl >= 0) ? max(0, w_row0 + shift - p.wl) : 0;
  const int w_full_hi = (p.wr >= 0) ? min(sk - 1, w_row0 + shift + p.wr) : sk - 1;
  const int w_full_lo = (p.wl >= 0) ? (w_row1 + shift - p.wl) : 0;
  const float cs = CAP ? p.scale * 2.885390081777927f / p.softcap : p.scale_log2;   // (softcap: the chains deliver 2*log2e * score*scale/softcap)
  float slope2 = 0.f;   // ALiBi slope of this head in log2 units
  if constexpr (ALIBI) slope2 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.alibi[(int64_t)b * p.alibi_bs + h] * kLog2eW)));   // (uniform; said so: one scalar register)
  int lim_hi[QB], lim_lo[QB];
  float lse_l[QB], delta_l[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int my_row = w_row0 + 32 * qb + qi;
    lim_hi[qb] = (p.wr >= 0) ? min(sk - 1, my_row + shift + p.wr) : sk - 1;
    lim_lo[qb] = (p.wl >= 0) ? (my_row + shift - p.wl) : 0;
    lse_l[qb] = INFINITY;   // rows past the end: P = 0
    delta_l[qb] = 0.f;
    if (my_row < sq) {
      const int64_t base = p.cu_q ? ((int64_t)h * p.total_q + q_row0) : (((int64_t)b * p.h + h) * p.sq);
      lse_l[qb] = p.lse[base + my_row] * kLog2eW;
      if constexpr (!FUSE_DELTA) delta_l[qb] = p.delta[base + my_row];
    }
  }

  auto step_active = [&](int i) __attribute__((always_inline)) {
    const int k0 = key_base + 32 * i;
    return wave_valid && (i >= 0) && (i < n_steps) && (k0 <= w_kmax) && (k0 + 31 >= w_kmin);
  };
  auto step_needs_mask = [&](int i) __attribute__((always_inline)) {
    const int k0 = key_base + 32 * i;
    return (k0 + 31 > w_full_hi) || (k0 < w_full_lo);
  };

  // ---- prologue: this wave's 64 rows of Q and dO -> LDS (coalesced DMA, swizzled source chunk) -> B-operand fragments in
  // accumulator registers.  Only the wave's own rows are touched, so its own vmcnt wait publishes them.
  {
    const E* qsrc = (const E*)p.q + q_boff + q_row0 * p.q_rs + (int64_t)h * p.q_hs;
    const E* dosrc = (const E*)p.dout + do_boff + q_row0 * p.do_rs + (int64_t)h * p.do_hs;
    constexpr int QDMA = (64 * ROW_BYTES) / 1024;   // pieces per wave and operand
    // Round 4, fuse_delta: softmax_d = rowsum(dO * O) of this wave's rows, computed HERE instead of by a pre-pass over the whole tensor (fa_bwd_delta_kernel: 55 us
    // of a 2.0 ms backward at config 3, and a launch).  O is read straight from memory in the dO fragments' layout -- lane (qi, hi) holds the eight channels
    // 16*ks + 8*hi .. of its row: sixteen 16-byte loads per lane and block, issued before the Q / dO DMA wait -- and meets the dO fragments on their way into the
    // accumulator registers (v_dot2_f32); the two lane halves of a row are added with one swap.  The dK/dV kernel reads the result from softmax_d: it is launched
    // behind this kernel (fa_api.cpp: do_bwd).
    u32x4 ofr[QB * KS];
    if constexpr (FUSE_DELTA) {
      const E* osrc = (const E*)p.o + (p.cu_q ? 0 : (int64_t)b * p.o_bs) + q_row0 * p.o_rs + (int64_t)h * p.o_hs;
      static_for<QB * KS>([&](auto fc) __attribute__((always_inline)) {
        constexpr int f = decltype(fc)::value, qb = f / KS, ks = f % KS;
        const int grow = min(w_row0 + 32 * qb + qi, sq - 1);
        ofr[f] = ld_global_16B(osrc + (int64_t)grow * p.o_rs + 16 * ks + 8 * hi, true);
      });
    }
    float dpart[QB] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < QDMA; ++i) {
      const int row = wave * 64 + i * RPD + d_row;
      const int grow = min(m0 + row, sq - 1);
      const int c = d_pc ^ swz16<D>(row);
      lds_dma_16B(qsrc + (int64_t)grow * p.q_rs + c * 8, lds + (wave * QDMA + i) * 1024);
      lds_dma_16B(dosrc + (int64_t)grow * p.do_rs + c * 8, lds + DO_OFF + (wave * QDMA + i) * 1024);
    }
    lds_dma_wait_all();
    const int fbase = (wave * 64 + qi) * ROW_BYTES + ((hi ^ swz16<D>(qi)) << 4);
    static_for<QB * KS>([&](auto fc) __attribute__((always_inline)) {
      constexpr int f = decltype(fc)::value, qb = f / KS, ks = f % KS;
      const int a = (fbase + qb * 32 * ROW_BYTES) ^ (ks << 5);
      const u32x4 xq = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)a;
      const u32x4 xd = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)(a + DO_OFF);
      // Q is multiplied by softmax_scale*log2(e) here, once, and rounded to the input dtype (the forward's 64-rows-per-wave kernel
      // does the same to its Q): scores leave the pipe ready for exp2
      const V8 raw = bitcast_u32x4<V8>(xq);
      V8 sc;
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        f32x2 pr = {(float)raw[j], (float)raw[j + 1]};
        pr *= f32x2{cs, cs};
        sc[j] = (E)pr[0]; sc[j + 1] = (E)pr[1];
      }
      acc_write_frag<BW_Q_BASE + 4 * f>(__builtin_bit_cast(u32x4, sc));
      acc_write_frag<BW_DO_BASE + 4 * f>(xd);
      if constexpr (FUSE_DELTA) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dpart[qb] = dot2_acc<E>(xd[j], ofr[f][j], dpart[qb]);
      }
    });
    if constexpr (FUSE_DELTA) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        const int my_row = w_row0 + 32 * qb + qi;
        const float dl = half_sum(dpart[qb]);
        if (my_row < sq) {
          delta_l[qb] = dl;
          const int64_t base = p.cu_q ? ((int64_t)h * p.total_q + q_row0) : (((int64_t)b * p.h + h) * p.sq);
          if (hi == 0) p.delta[base + my_row] = dl;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();   // the K/V rings reuse this LDS
  }

  // ---- K/V tiles: global -> LDS by DMA through a buffer descriptor (see fa_fwd_w64.hip), swz16 on the source chunk -----
  constexpr int NDMA = TILE_BYTES / 1024, DPW = NDMA / NW;
  unsigned koff_l[DPW], voff_l[DPW];
#pragma unroll
  for (int i = 0; i < DPW; ++i) {
    const int row = (wave * DPW + i) * RPD + d_row;
    const int c = d_pc ^ swz16<D>(row);
    koff_l[i] = (unsigned)(row * (int)p.k_rs + c * 8) * 2u - (unsigned)(i * 1024);
    voff_l[i] = (unsigned)(row * (int)p.v_rs + c * 8) * 2u - (unsigned)(i * 1024);
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
  auto dma_tile = [&](auto isvc, int slot, int t) __attribute__((always_inline)) {  // t relative to n_min
    constexpr bool ISV = decltype(isvc)::value != 0;
    const int n = n_min + t;
    const int64_t rs = ISV ? p.v_rs : p.k_rs;
    const unsigned lds_dst = (unsigned)((ISV ? V_RING : 0) + slot * TILE_BYTES + wave * DPW * 1024);
    unsigned vo[DPW];
    if (n * BN + BN <= sk) {
      const unsigned toff = (unsigned)n * (unsigned)(BN * 2) * (unsigned)rs;
#pragma unroll
      for (int i = 0; i < DPW; ++i) vo[i] = (ISV ? voff_l[i] : koff_l[i]) + toff;
    } else {  // last, partial tile: rows past the last key are clamped to it (finite data, masked to P = 0)
#pragma unroll
      for (int i = 0; i < DPW; ++i) {
        const int row = (wave * DPW + i) * RPD + d_row;
        const int grow = min(n * BN + row, sk - 1);
        const int c = d_pc ^ swz16<D>(row);
        vo[i] = ((unsigned)grow * (unsigned)rs + (unsigned)(c * 8)) * 2u - (unsigned)(i * 1024);
      }
    }
    dma_pieces(ISV ? v_srd : k_srd, vo, lds_dst);
  };

  if (n_tiles > 0) {
    dma_tile(ICw<0>{}, 0, 0);
    dma_tile(ICw<1>{}, 0, 0);
  }

  // per-lane LDS read addresses, ring slot included: ka = row fragments of the tile being scored (V at + V_RING),
  // ta = transposed fragments of the tile before it.  Advanced by one slot per iteration.
  int ka[KS], ta0[DB], ta1[DB];
  {
    const int kbase = qi * ROW_BYTES + ((hi ^ swz16<D>(qi)) << 4);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) ka[ks] = kbase ^ (ks << 5);
    const int tr_i = lane & 15, tr_half = (lane >> 4) & 1, tr_rr = tr_i >> 2, tr_cc = tr_i & 3;
    const int tb0 = tile_off<D>(4 * hi + tr_rr, 2 * tr_half + (tr_cc >> 1)) + (tr_cc & 1) * 8;
    const int tb1 = tile_off<D>(8 + 4 * hi + tr_rr, 2 * tr_half + (tr_cc >> 1)) + (tr_cc & 1) * 8;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      ta0[db] = (tb0 ^ (db << 6)) + (RING - 1) * TILE_BYTES;   // "tile -1" sits in the last slot
      ta1[db] = (tb1 ^ (db << 6)) + (RING - 1) * TILE_BYTES;
    }
  }
  int slot_k = 0;   // ring slot of the tile being scored (wave-uniform)

  acc_zero_range<0>(std::make_integer_sequence<int, 32 * DB>{});   // dQ^T: tuple qb*DB + db
  f32x16 sA[QB], sB[QB], dpA[QB], dpB[QB];
  u32x4 fA[QB][2], fB[QB][2];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { sA[qb][r] = 0.f; sB[qb][r] = 0.f; dpA[qb][r] = 0.f; dpB[qb][r] = 0.f; }
#pragma unroll
    for (int t = 0; t < 2; ++t) { fA[qb][t] = u32x4{0u, 0u, 0u, 0u}; fB[qb][t] = u32x4{0u, 0u, 0u, 0u}; }
  }
  // The score chains' C operand: -LSE*log2e broadcast (+inf LSE of a row past the end: -inf, P = 0).  (The dP chains could take -delta the same
  // way -- built and measured as FA_BW64_CDELTA, experiments/ablations/fa_bwd_w64.patch -- but two more 32-register broadcasts do not fit: 352 bytes of scratch in the tile loop.)
  f32x16 nlse[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      nlse[qb][r] = ALIBI ? __builtin_fmaf(slope2, (float)acc_row(r, 0), -lse_l[qb]) : -lse_l[qb];
    }
  }
  // (CAP) c = softcap*log2e, -2c, and per row c - LSE*log2e (-inf for a row past the end: P = 0)
  float cap_m2c = 0.f, cap_off[QB] = {0.f, 0.f};
  if constexpr (CAP) {
    const float capc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.softcap * kLog2eW)));
    cap_m2c = -2.f * capc;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) cap_off[qb] = capc - lse_l[qb];
  }
  constexpr float kCapTiny = 7.888609052210118e-31f;   // 2^-100: y * tiny vanishes for a finite y and keeps a masked score's -inf
  // (DROP) the seven round keys of this (batch, head), the keep threshold, 1 / (1 - p), each lane's row
  unsigned drop_kr[7] = {0u, 0u, 0u, 0u, 0u, 0u, 0u}, drop_thr = 255u, drop_row[QB] = {0u, 0u};
  float drop_rp = 1.f;
  if constexpr (DROP) {
    const unsigned k0_ = (unsigned)__builtin_amdgcn_readfirstlane((int)drop_bh_key(p.rng, b * p.h + h));
#pragma unroll
    for (int r = 0; r < 7; ++r) drop_kr[r] = k0_ + (unsigned)r * 0x9E3779B9u;
    drop_thr = (unsigned)__builtin_amdgcn_readfirstlane((int)p.drop_thr8);
    drop_rp = p.rp_keep;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) drop_row[qb] = (unsigned)(w_row0 + 32 * qb + qi);
  }
  bool have_cur = false, have_prev = false;

  lds_dma_wait_all();
  __syncthreads();

  auto rd_tr = [&](int a0, int a1) __attribute__((always_inline)) {
    const s16x4 lo = lds_read_tr16((const char FA_LDS*)(unsigned long)(unsigned)a0);
    const s16x4 hi4 = lds_read_tr16((const char FA_LDS*)(unsigned long)(unsigned)a1);
    return __builtin_bit_cast(u32x4, combine_tr<V8>(lo, hi4));
  };
  // dS_i of one element
  auto ds_elem = [&](float sv, float dpv, int qb, float sb) __attribute__((always_inline)) {
    if constexpr (CAP) {   // (one element in one go: the head / tail steps; the steady-state step stages this over three gaps)
      const float rr = __builtin_amdgcn_rcpf(fast_exp2(sv) + 1.f);
      const float pv = fast_exp2(__builtin_fmaf(rr, cap_m2c, __builtin_fmaf(sv, kCapTiny, cap_off[qb])));
      return pv * (dpv - delta_l[qb]) * __builtin_fmaf(-rr, rr, rr);
    }
    if constexpr (ALIBI) sv += sb;
    // (DROP: the chains start from C = 0 and LSE is subtracted here -- one more instruction per element in a variant that has ten, for the 32 registers of the
    // broadcast: with them the D = 128 instantiation spilled five)
    if constexpr (DROP) sv -= lse_l[qb];
    const float pv = fast_exp2(sv);   // sv = s*scale*log2e - LSE*log2e (the subtraction rode in the chain's C operand)
    return pv * (dpv - delta_l[qb]);
  };
  // (DROP) dP of a kept pair is scaled by 1 / (1 - p), a dropped pair contributes -delta alone; w = the Philox word of the element's group, byte r & 3 its random byte
  auto drop_dp = [&](float dpv, unsigned w, int r) __attribute__((always_inline)) {
    return (((w >> (8 * (r & 3))) & 0xffu) > drop_thr) ? 0.f : dpv * drop_rp;
  };
  // (ALIBI) the step's per-lane part of the bias, slope*log2e * (first key of the step + 4*hi - row - shift), from the same integer the masked steps compare
  // against: rel_hi = (row + shift) - k0 - 4*hi is the lane's right bound under the causal mask this variant requires (a row past the end, where the bound is
  // clamped, has LSE = +inf and P = 0 whatever the bias) -- no register of its own across the tile loop
  auto bias_of = [&](int rel_hi) __attribute__((always_inline)) { return -slope2 * (float)rel_hi; };
  // (ALIBI) ... and the left bound relative to the step follows from the right one (window_right = 0: lim_lo = lim_hi - window_left): two registers fewer across the loop
  const int lo_gap = __builtin_amdgcn_readfirstlane(p.wl >= 0 ? p.wl : (1 << 30));
  auto rel_lo_of = [&](int qb, int rel_hi, int k04) __attribute__((always_inline)) { return ALIBI ? rel_hi - lo_gap : lim_lo[qb] - k04; };
  // (ALIBI) ... and query block 1's right bound is block 0's + 32 (where that is past the last key the row is past the end: P = 0 through its LSE): one more
  auto lim_hi_of = [&](int qb) __attribute__((always_inline)) { return ALIBI ? lim_hi[0] + 32 * qb : lim_hi[qb]; };
  auto pack2 = [&](float x0, float x1) __attribute__((always_inline)) {
    using V2 = __attribute__((ext_vector_type(2))) E;
    V2 pr;
    pr[0] = (E)x0;
    pr[1] = (E)x1;
    return __builtin_bit_cast(unsigned, pr);
  };

  // ---- generic (head / tail) step: compiler-ordered, drains the matrix pipe before the VALU touches MFMA results ------
  auto generic_step = [&](auto halfc, int i, f32x16 (&s_cur)[QB], f32x16 (&dp_cur)[QB], f32x16 (&s_nxt)[QB], f32x16 (&dp_nxt)[QB],
                          const u32x4 (&f_prev)[QB][2], u32x4 (&f_cur)[QB][2]) __attribute__((always_inline)) {
    constexpr int half = decltype(halfc)::value;
    constexpr int HOFF = half * 32 * ROW_BYTES;
    const bool do_qk = step_active(i + 1);
    const bool do_sm = have_cur;
    const bool do_pv = have_prev;
    if (do_qk) {
      static_for<KS>([&](auto ksc) __attribute__((always_inline)) {
        constexpr int ks = decltype(ksc)::value;
        const u32x4 kf = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)(ka[ks] + HOFF);
        const u32x4 vf = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)(ka[ks] + HOFF + V_RING);
        if constexpr (ks == 0) {
          if constexpr (CAP || DROP) {
            mfma_v_first0<E, BW_Q_BASE>(s_nxt[0], kf);
            mfma_v_first0<E, BW_Q_BASE + 4 * KS>(s_nxt[1], kf);
          } else {
            mfma_v_first<E, BW_Q_BASE>(s_nxt[0], kf, nlse[0]);
            mfma_v_first<E, BW_Q_BASE + 4 * KS>(s_nxt[1], kf, nlse[1]);
          }
          mfma_v_first0<E, BW_DO_BASE>(dp_nxt[0], vf);
          mfma_v_first0<E, BW_DO_BASE + 4 * KS>(dp_nxt[1], vf);
        } else {
          mfma_v_acc<E, BW_Q_BASE + 4 * ks>(s_nxt[0], kf);
          mfma_v_acc<E, BW_Q_BASE + 4 * (KS + ks)>(s_nxt[1], kf);
          mfma_v_acc<E, BW_DO_BASE + 4 * ks>(dp_nxt[0], vf);
          mfma_v_acc<E, BW_DO_BASE + 4 * (KS + ks)>(dp_nxt[1], vf);
        }
      });
    }
    if (do_sm) {
      mfma_drain_v(s_cur[0], s_cur[1]);
      mfma_drain_v(dp_cur[0], dp_cur[1]);
      const bool mask = step_needs_mask(i);
      const int k0 = key_base + 32 * i;
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        const int rel_hi = lim_hi_of(qb) - k0 - 4 * hi, rel_lo = rel_lo_of(qb, rel_hi, k0 + 4 * hi);
        const float sbq = bias_of(rel_hi);
        float dsv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int off = acc_row(r, 0);
          float sv = s_cur[qb][r];
          if (mask) sv = ((off <= rel_hi) && (off >= rel_lo)) ? sv : -INFINITY;
          float dpv = dp_cur[qb][r];
          if constexpr (DROP) dpv = drop_dp(dpv, drop_bytes(drop_kr[0], (int)drop_row[qb], (k0 + 8 * (r >> 2) + 4 * hi) >> 2), r);
          dsv[r] = ds_elem(sv, dpv, qb, sbq);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) f_cur[qb][c >> 2][c & 3] = pack2(dsv[2 * c], dsv[2 * c + 1]);
      }
    }
    if (do_pv) {
      static_for<2 * DB>([&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, db = g % DB, t = g / DB;
        constexpr int off = HOFF + 16 * t * ROW_BYTES;
        const u32x4 kt = rd_tr(ta0[db] + off, ta1[db] + off);
        mfma_q_acc<E, db>(kt, f_prev[0][t]);
        mfma_q_acc<E, DB + db>(kt, f_prev[1][t]);
      });
    }
    if (do_qk) {
      mfma_drain_v(s_nxt[0], s_nxt[1]);
      mfma_drain_v(dp_nxt[0], dp_nxt[1]);
    }
    have_prev = do_sm;
    have_cur = do_qk;
  };

  // ---- steady-state step: NG MFMA gaps -----------------------------------------------------------------------------------
  //   gaps [0, 2KS)        : S_{i+1}[qb] chain, k-step g/2          (K row fragment read once, used by both query blocks)
  //   gaps [2KS, 4KS)      : dP_{i+1}[qb] chain                     (V row fragment)
  //   gaps [4KS, 4KS+4DB)  : dQ[qb][db] += K_{i-1}^T . dS_{i-1}[qb] (transposed K fragment)
  //   VALU: the 32 elements of dS_i spread over the gaps, each pair packed one gap after it is complete; the step's DMA
  //   pieces in the odd gaps 1, 3, ...  The score chains end 16 MFMAs before the next step reads them.
  auto fast_step = [&](auto halfc, auto maskc, int i_cur, f32x16 (&s_cur)[QB], f32x16 (&dp_cur)[QB], f32x16 (&s_nxt)[QB],
                       f32x16 (&dp_nxt)[QB], const u32x4 (&f_prev)[QB][2], u32x4 (&f_cur)[QB][2], const u32x4& dma_srd,
                       const unsigned (&dma_off)[DPW], unsigned dma_toff, unsigned dma_dst, u32x4 (&fr)[AHK + 1]) __attribute__((always_inline)) {
    constexpr int half = decltype(halfc)::value;
    constexpr bool MASK = decltype(maskc)::value != 0;
    constexpr int HOFF = half * 32 * ROW_BYTES;
    constexpr int QKG = 2 * KS, DQG = 4 * DB, NG = 2 * QKG + DQG;
    constexpr int AH = AHK, RNG = AH + 1;     // operand reads run AH fragment slots (2 gaps each) ahead of their MFMAs
    constexpr int NF = 2 * KS + 2 * DB;     // fragment slots: KS K rows, KS V rows, 2*DB transposed K
    auto rd_frag = [&](int f) __attribute__((always_inline)) {
      if (f < KS) {
        fr[f % RNG] = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)(ka[f] + HOFF);
      } else if (f < 2 * KS) {
        fr[f % RNG] = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)(ka[f - KS] + HOFF + V_RING);
      } else if (f < NF) {
        const int op = f - 2 * KS, db = op % DB, t = op / DB;
        fr[f % RNG] = rd_tr(ta0[db] + HOFF + 16 * t * ROW_BYTES, ta1[db] + HOFF + 16 * t * ROW_BYTES);
      }
    };
    // elements of dS_i whose (first) stage was issued before gap x (e = 16*qb + r); the last stage of all 32 before the last gap, pairs are packed one gap later.
    // CAP: three stages a gap apart (LAT = 2 more gaps); DROP: the first elements wait E0 gaps for their group's Philox word
    constexpr int E0 = DROP ? (D == 128 ? 4 : 2) : 0, LAT = CAP ? 2 : 0, ESPAN = NG - 1 - LAT - E0;
    auto el_end = [](int x) constexpr { return x <= E0 ? 0 : ((32 * (x - E0) + ESPAN - 1) / ESPAN > 32 ? 32 : (32 * (x - E0) + ESPAN - 1) / ESPAN); };
    // (DROP) round r of the Philox call of group G = 4*qb + g (elements 16*qb + 4*g .. + 3) sits in gap drop_gap(G, r): done no later than the gap of its first element
    auto drop_gap = [](int G, int r) constexpr { return ((7 * G + r) * (NG - 2 * E0)) / 56; };
    constexpr bool drop_in_time = [&]() constexpr {
      for (int G = 0; G < 8; ++G) { int x = 0; while (el_end(x + 1) <= 4 * G) ++x; if (drop_gap(G, 6) > x) return false; }
      return true;
    }();
    static_assert(!DROP || drop_in_time, "a Philox word is finished before its first element is processed");
    unsigned ph_c0[8], ph_c1[8];
    float cap_a1[32], cap_d[32], cap_arg[32], cap_w[32];   // (CAP) values in flight between the stages
    float dsv[QB][16];
    float sbv = 0.f;   // (ALIBI) the bias of the query block whose elements are being processed: made when its first element comes up
    int rel_hi[QB] = {0, 0}, rel_lo[QB] = {0, 0};
    const int k0c = key_base + 32 * i_cur;
    if constexpr (MASK) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) { rel_hi[qb] = lim_hi_of(qb) - k0c - 4 * hi; rel_lo[qb] = rel_lo_of(qb, rel_hi[qb], k0c + 4 * hi); }
    }
    if constexpr (half != 1) {   // (second step: requested by the first step's last gaps)
#pragma unroll
      for (int f = 0; f < AH; ++f) rd_frag(f);
    }
    __builtin_amdgcn_sched_barrier(0);
    static_for<NG>([&](auto xc) __attribute__((always_inline)) {
      constexpr int x = decltype(xc)::value;
      constexpr int f = x / 2, qb = x & 1;
      if constexpr (qb == 0) rd_frag(f + AH);
      // one wait per TWO fragment slots (as fa_fwd_w64.hip): before the MFMAs of an even slot f, wait until slot f + 1 has landed too
      if constexpr (qb == 0 && (f & 1) == 0 && f + 1 < NF) {
        constexpr auto ops = [](int g) constexpr { return g < 2 * KS ? 1 : g < NF ? 2 : 0; };
        constexpr int out = [&]() constexpr { int n = 0; for (int g = f + 2; g <= f + AH; ++g) n += ops(g); return n; }();
        __builtin_amdgcn_s_waitcnt(0xC07F | (out << 8));
      }
      if constexpr (x < QKG) {
        if constexpr (f == 0 && (CAP || DROP)) mfma_v_first0<E, BW_Q_BASE + 4 * (qb * KS)>(s_nxt[qb], fr[f % RNG]);
        else if constexpr (f == 0) mfma_v_first<E, BW_Q_BASE + 4 * (qb * KS)>(s_nxt[qb], fr[f % RNG], nlse[qb]);
        else mfma_v_acc<E, BW_Q_BASE + 4 * (qb * KS + f)>(s_nxt[qb], fr[f % RNG]);
      } else if constexpr (x < 2 * QKG) {
        constexpr int ks = f - KS;
        if constexpr (ks == 0) mfma_v_first0<E, BW_DO_BASE + 4 * (qb * KS)>(dp_nxt[qb], fr[f % RNG]);
        else mfma_v_acc<E, BW_DO_BASE + 4 * (qb * KS + ks)>(dp_nxt[qb], fr[f % RNG]);
      } else {
        constexpr int op = f - 2 * KS;
        mfma_q_acc<E, qb * DB + op % DB>(fr[f % RNG], f_prev[qb][op / DB]);
      }
      if constexpr ((x & 1) && (x / 2) < DPW) {
        constexpr int pc = x / 2;
        // (the tile's byte offset rides in the scalar-offset operand, as in fa_fwd_w64.hip: no per-piece address add)
        if constexpr (pc == 0)
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" : : "v"(dma_off[pc]), "s"(dma_dst), "s"(dma_srd), "s"(dma_toff) : "memory");
        else
          asm volatile("buffer_load_dwordx4 %0, %1, %3 offen offset:%c2 lds" : : "v"(dma_off[pc]), "s"(dma_srd), "i"(1024 * pc), "s"(dma_toff) : "memory");
      }
      if constexpr (DROP) {
#pragma unroll
        for (int G = 0; G < 8; ++G)
#pragma unroll
          for (int r = 0; r < 7; ++r)
            if (drop_gap(G, r) == x) {
              if (r == 0) { ph_c0[G] = (unsigned)((k0c >> 2) + 2 * (G & 3)) + (unsigned)hi; ph_c1[G] = drop_row[G >> 2]; }
              const unsigned mh = __umulhi(0xD256D193u, ph_c0[G]), ml = 0xD256D193u * ph_c0[G];
              unsigned x3;   // mh ^ round key ^ c1 in one instruction (the key is a scalar operand)
              asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(x3) : "v"(mh), "s"(drop_kr[r]), "v"(ph_c1[G]));
              ph_c0[G] = x3;
              ph_c1[G] = ml;
            }
      }
      if constexpr (CAP) {   // oldest elements first: C = the probability and dS, B = 1/(2^y + 1), the exponent and 1 - tanh^2, A = the mask carrier, 2^y + 1
#pragma unroll
        for (int e = el_end(x - 2); e < el_end(x - 1); ++e) {
          const int eq = e >> 4, r = e & 15;
          dsv[eq][r] = fast_exp2(cap_arg[e]) * (dp_cur[eq][r] - delta_l[eq]) * cap_w[e];
        }
#pragma unroll
        for (int e = el_end(x - 1); e < el_end(x); ++e) {
          const float rr = __builtin_amdgcn_rcpf(cap_d[e]);
          cap_arg[e] = __builtin_fmaf(rr, cap_m2c, cap_a1[e]);
          cap_w[e] = __builtin_fmaf(-rr, rr, rr);
        }
      }
#pragma unroll
      for (int e = el_end(x); e < el_end(x + 1); ++e) {
        const int eq = e >> 4, r = e & 15;
        float sv = s_cur[eq][r];
        if constexpr (ALIBI) { if (r == 0) sbv = bias_of(MASK ? rel_hi[eq] : lim_hi_of(eq) - k0c - 4 * hi); }
        if constexpr (MASK) {
          const int off = acc_row(r, 0);
          sv = ((off <= rel_hi[eq]) && (off >= rel_lo[eq])) ? sv : -INFINITY;
        }
        if constexpr (CAP) {
          cap_a1[e] = __builtin_fmaf(sv, kCapTiny, cap_off[eq]);
          cap_d[e] = fast_exp2(sv) + 1.f;
        } else {
          float dpv = dp_cur[eq][r];
          if constexpr (DROP) dpv = drop_dp(dpv, ph_c0[4 * eq + (r >> 2)], r);
          dsv[eq][r] = ds_elem(sv, dpv, eq, sbv);
        }
      }
#pragma unroll
      for (int c = el_end(x - 1 - LAT) / 2; c < el_end(x - LAT) / 2; ++c) {
        const int cq = c >> 3, r = 2 * (c & 7);
        unsigned pw = pack2(dsv[cq][r], dsv[cq][r + 1]);
        asm volatile("" : "+v"(pw));   // pinned to this gap
        f_cur[cq][r >> 3][(r & 7) >> 1] = pw;
      }
      // carry (fa_fwd_w64.hip): behind this step's last LDS wait the ring is free -- the first AH K-row fragments of the second step (half 1 of the same tile)
      if constexpr (half == 0 && x >= NG - AH) {
        constexpr int fn = x - (NG - AH);
        fr[fn % RNG] = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)(ka[fn] + 32 * ROW_BYTES);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // iteration u (0..n_tiles): steps 2u-1 and 2u score tile u (K_u, V_u row fragments) and accumulate tile u-1 (K_{u-1}
  // transposed); K_{u+1}, V_{u+1} are DMA'd during the iteration into the third ring slot.
  int uf_lo = 1, uf_hi = 0;
  if (wave_valid && n_tiles > 0) {
    const int a_lo = max(0, (w_kmin - key_base) >> 5);
    const int a_hi = min(n_steps - 1, (w_kmax - key_base) >> 5);
    uf_lo = (a_lo + 3) >> 1;
    uf_hi = (a_hi - 1) >> 1;
    if (sk % BN != 0) uf_hi = min(uf_hi, sk / BN - n_min - 2);   // the steady-state DMA does not clamp rows: full tiles only
  }
  auto iter_head = [&](int u) __attribute__((always_inline)) {
    if (u + 1 < n_tiles) {
      const int nslot = slot_k == RING - 1 ? 0 : slot_k + 1;
      dma_tile(ICw<0>{}, nslot, u + 1);
      dma_tile(ICw<1>{}, nslot, u + 1);
    }
  };
  auto iter_tail = [&]() __attribute__((always_inline)) {
    // advance the read addresses by one ring slot (ka: slot_k -> slot_k + 1; ta: slot_k - 1 -> slot_k)
    const int dk = slot_k == RING - 1 ? -(RING - 1) * TILE_BYTES : TILE_BYTES;
    const int dt = slot_k == 0 ? -(RING - 1) * TILE_BYTES : TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) ka[ks] += dk;
#pragma unroll
    for (int db = 0; db < DB; ++db) { ta0[db] += dt; ta1[db] += dt; }
    slot_k = slot_k == RING - 1 ? 0 : slot_k + 1;
    lds_dma_wait_all();
    __syncthreads();
  };
  auto generic_iter = [&](int u) __attribute__((always_inline)) {
    iter_head(u);
    generic_step(ICw<0>{}, 2 * u - 1, sA, dpA, sB, dpB, fA, fB);
    generic_step(ICw<1>{}, 2 * u, sB, dpB, sA, dpA, fB, fA);
    iter_tail();
  };
  if (n_tiles > 0) {
    int u = 0;
    const int head_end = min(max(uf_lo, 0), n_tiles + 1);
    for (; u < head_end; ++u) generic_iter(u);
    const unsigned wave_dst = (unsigned)(wave * DPW * 1024);
    auto fast_iter = [&](auto maskc, int uu) __attribute__((always_inline)) {
      const int nslot = slot_k == RING - 1 ? 0 : slot_k + 1;
      const unsigned toff_k = (unsigned)(n_min + uu + 1) * (unsigned)(BN * 2) * (unsigned)p.k_rs;
      const unsigned toff_v = (unsigned)(n_min + uu + 1) * (unsigned)(BN * 2) * (unsigned)p.v_rs;
      // (a tile past the last one lands in the slot nobody reads again; rows past the descriptor's range are never a fault)
      u32x4 fring[AHK + 1];   // operand ring of the two steps (the second step's first fragments are requested by the first)
      fast_step(ICw<0>{}, maskc, 2 * uu - 1, sA, dpA, sB, dpB, fA, fB, k_srd, koff_l, toff_k,
                __builtin_amdgcn_readfirstlane((unsigned)(nslot * TILE_BYTES) + wave_dst), fring);
      fast_step(ICw<1>{}, maskc, 2 * uu, sB, dpB, sA, dpA, fB, fA, v_srd, voff_l, toff_v,
                __builtin_amdgcn_readfirstlane((unsigned)(V_RING + nslot * TILE_BYTES) + wave_dst), fring);
      have_prev = step_active(2 * uu);
      have_cur = step_active(2 * uu + 1);
      iter_tail();
    };
    int um_lo = uf_lo, um_hi = uf_hi;
    {
      const int f_lo = (w_full_lo - key_base + 31) >> 5;   // first step with no left-masked key
      const int f_hi = (w_full_hi - 31 - key_base) >> 5;   // last step with no right-masked key
      um_lo = max(uf_lo, (f_lo + 2) >> 1);                 // steps 2u-1 and 2u both unmasked
      um_hi = min(uf_hi, f_hi >> 1);
    }
    auto fast_range = [&](auto maskc, int hi_incl) __attribute__((always_inline)) {
      for (; u <= hi_incl; ++u) fast_iter(maskc, u);
    };
    fast_range(ICw<1>{}, min(uf_hi, um_lo - 1));
    fast_range(ICw<0>{}, um_hi);
    fast_range(ICw<1>{}, uf_hi);
    for (; u <= n_tiles; ++u) generic_iter(u);
  }

  if (!wave_valid) return;
  mfma_drain_acc();
  // dQ tile through LDS (the rings are free after the last tile barrier): whole-row stores, softmax_scale applied here
  E* dqtile = (E*)p.dq + dq_boff + (q_row0 + w_row0) * p.dq_rs + (int64_t)h * p.dq_hs;
  static_for<QB>([&](auto qbc) __attribute__((always_inline)) {
    constexpr int qb = decltype(qbc)::value;
    f32x16 o_v[DB];
    static_for<DB>([&](auto dbc) __attribute__((always_inline)) {
      constexpr int db = decltype(dbc)::value;
      acc_read_tuple<16 * (qb * DB + db)>(o_v[db]);
    });
    const int row0 = w_row0 + 32 * qb;
    // (the lane index made again from the hardware counter: as `lane` it is one more register alive across the tile loop -- the ALiBi variant at D = 128 spilled it)
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (row0 < sq)
      store_tile_via_lds<E, D>(lds + (wave * 64 + qb * 32) * (ROW_BYTES + 16), o_v, CAP ? 4.f * p.scale : p.scale, dqtile + (int64_t)(32 * qb) * p.dq_rs, p.dq_rs,
                               sq - row0, lane_e);
  });
}

template <typename E, int D, bool FUSE_DELTA, int FEAT>
static int launch_bwd_dq_w64_f(const BwdK& p, hipStream_t stream) {
  constexpr int TILE = 64 * D * 2, RINGS = 2 * 3 * TILE, STAGE_IN = 2 * 256 * D * 2, STAGE_OUT = 256 * (D * 2 + 16);
  constexpr int smem = (RINGS > STAGE_IN ? (RINGS > STAGE_OUT ? RINGS : STAGE_OUT) : (STAGE_IN > STAGE_OUT ? STAGE_IN : STAGE_OUT));
  auto kern = fa_bwd_dq_w64_kernel<E, D, FUSE_DELTA, FEAT>;
  static std::atomic<unsigned long long> attr_mask{0};
  if (ensure_dyn_lds(attr_mask, (const void*)kern, smem, true) != 0) return -1;
  const long long total = p.q_list ? (long long)p.q_bound * p.h : units_grid(p.q_units, p.q_unit_size);
  if (total <= 0) return 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), smem, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <typename E, int D>
static int launch_bwd_dq_w64_t(const BwdK& p, hipStream_t stream) {
  // (softcap / dropout: built for the fused-delta order only -- the order the dispatch uses; FA_BWD_FUSE_DELTA=0 falls back to fa_bwd.hip's feature kernels -- and for
  // head dim 128 only: at head dim 64, twice the vector work per MFMA, they measured 2-4 % behind the 4-wave feature kernel, profiles/r05_bwd_features_w64.txt)
  if constexpr (D == 128) {
    if (p.softcap > 0.f) return p.fuse_delta ? launch_bwd_dq_w64_f<E, D, true, FEAT_CAP>(p, stream) : -2;
    if (p.rng) return p.fuse_delta ? launch_bwd_dq_w64_f<E, D, true, FEAT_DROP>(p, stream) : -2;
  } else if (p.softcap > 0.f || p.rng) {
    return -2;
  }
  if (p.alibi) return p.fuse_delta ? launch_bwd_dq_w64_f<E, D, true, FEAT_ALIBI>(p, stream) : launch_bwd_dq_w64_f<E, D, false, FEAT_ALIBI>(p, stream);
  return p.fuse_delta ? launch_bwd_dq_w64_f<E, D, true, 0>(p, stream) : launch_bwd_dq_w64_f<E, D, false, 0>(p, stream);
}

// 4 waves x 64 query rows per workgroup (the caller sized nmb / the work list for 256-row blocks).  Plain attention, softcap, dropout, or ALiBi under a causal
// right bound -- one feature at a time; -2 = not covered, the caller falls back to fa_bwd_dq_kernel<.., 8, ..> on the same blocks.
int launch_bwd_dq_w64(const BwdK& p, int dtype_bf16, int d, hipStream_t stream) {
  if ((p.softcap > 0.f) + (p.rng != nullptr) + (p.alibi != nullptr) > 1) return -2;
  if (p.alibi != nullptr && p.wr != 0) return -2;   // the bias is linear in the key only where no visible key lies right of the diagonal
  const uint64_t span = ((uint64_t)(p.sk > 0 ? p.sk : 1) + 128) * (uint64_t)(p.k_rs > p.v_rs ? p.k_rs : p.v_rs) * 2u;
  if (span >= (1ull << 32)) return -2;   // buffer addressing: 32-bit byte offsets from the (batch, kv-head) base
  if (d == 128) return dtype_bf16 ? launch_bwd_dq_w64_t<__bf16, 128>(p, stream) : launch_bwd_dq_w64_t<_Float16, 128>(p, stream);
  if (d == 64) return dtype_bf16 ? launch_bwd_dq_w64_t<__bf16, 64>(p, stream) : launch_bwd_dq_w64_t<_Float16, 64>(p, stream);
  return -2;
}

}  // namespace fa
