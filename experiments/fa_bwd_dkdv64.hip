// dK/dV kernel of the backward pass for gfx950, "64 keys per wave, one wave per SIMD" schedule (reference: the dK/dV half of
// compute_dq_dk_dv_1colblock, csrc/flash_attn/src/flash_bwd_kernel.h:80-795; same contractions and arithmetic as
// fa_bwd_dkdv_kernel in fa_bwd.hip, which stays the kernel for the feature variants and the other head dims).
//
// Why a second schedule.  fa_bwd_dkdv_kernel runs eight waves of 32 keys, two per SIMD, and each wave walks a sub-block
// through three strictly sequential phases -- 16 MFMAs (S, dP), the softmax arithmetic, 16 MFMAs (dV, dK).  The two waves of
// a SIMD sit in the same phase all the time, so matrix and vector work never overlap: its time is MFMA time + everything
// else (profiles/r02_bwd_schedules.txt; rotating one wave's phases did not change that).  Here a workgroup is four waves, one
// per SIMD with the 512-register budget, and a wave owns TWO 32-key blocks; that leaves registers for two S / dP tuple sets,
// and the wave software-pipelines its own stream:
//
//   unit u = (32-query sub-block, key block); per unit:  P1 = S, dP (16 MFMAs),  SM = P, dS (VALU),  P2 = dV, dK (16 MFMAs)
//   steady state:   [ P1(u+1) with SM(u) hand-placed into its 16 MFMA gaps, one element per gap ]   [ P2(u) ]
//
// A streamed tile is 64 queries = 4 units per wave.  The last unit of a tile pipelines against the first unit of the NEXT
// tile, hence two barriers per tile: "next tile landed" before that step, "this tile is free" after it.
// Layouts (lane = key, column = key; P and dS are directly B operands) as in fa_bwd.hip.
#include <cstdlib>
#include <type_traits>

#include "fa_device.h"   // (-I flash-attention_amd/csrc: experiments/build_experiments.py)
#include "fa_kernel_params.h"
#include "fa_launch.h"
#include "fa_fwd_w64_regs.h"
#define FA_W64_CLOB FA_W64_ACC_CLOBBERS_256
#include "fa_w64_asm.h"

#ifndef FA_DKDV64_PF
#define FA_DKDV64_PF 3    // row-major LDS operands of S / dP are read this many MFMA slots minus one ahead
#endif
#ifndef FA_DKDV64_PFT
#define FA_DKDV64_PFT 3   // same for the transposed operands of dV / dK
#endif
#ifndef FA_DKDV64_ABL
#define FA_DKDV64_ABL 0  // timing ablations (results become wrong): 1 = no interleave (P1 then SM, sequentially), 2 = no exp2,
                         // 4 = LDS operands read once per segment, 8 = no DMA wait / barriers, 16 = no P / dS arithmetic,
                         // 32 = no S / dP MFMAs, 64 = no dV / dK MFMAs, 128 = no Q / dO DMA after the first tiles
#endif

namespace fa {

template <int N> using IC64 = std::integral_constant<int, N>;

namespace {
// The sixteen 32 x 32 accumulator tiles of a wave (dV^T and dK^T: 2 key blocks x D/32 blocks each = 256 registers at D = 128) are
// the accumulator half of the register file, owned by the asm and named literally (fa_w64_asm.h): left to the register allocator
// the kernel spilled ~190 registers.  Tile T at a[16T : 16T+15], T = (2 * kb + (dK ? 1 : 0)) * DB + db.
template <typename E, int T> FA_DEVINL void mfma_tile_acc(u32x4 a, u32x4 b) {
  if constexpr (std::is_same<E, __bf16>::value)
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" : : "v"(a), "v"(b), "i"(16 * T), "i"(16 * T + 15) : FA_W64_CLOB);
  else
    asm volatile("v_mfma_f32_32x32x16_f16 a[%c2:%c3], %0, %1, a[%c2:%c3]" : : "v"(a), "v"(b), "i"(16 * T), "i"(16 * T + 15) : FA_W64_CLOB);
}
// S / dP chains: d (arch VGPR tuple) = a . b (+ d).  Asm as well: a builtin MFMA lets hipcc place its result in accumulator
// registers -- the ones the tiles above live in (the clobber lists only protect them across an asm statement, not between two).
// hipcc neither sees these MFMAs' latency nor pads their hazards (fa_w64_asm.h): a result is read by the vector ALU only after
// another 16-MFMA segment, or behind mfma_drain_v.
template <typename E> FA_DEVINL void mfma_v_first(f32x16& d, u32x4 a, u32x4 b) {
  if constexpr (std::is_same<E, __bf16>::value)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b) : FA_W64_CLOB);
  else
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b) : FA_W64_CLOB);
}
template <typename E> FA_DEVINL void mfma_v_acc(f32x16& d, u32x4 a, u32x4 b) {
  if constexpr (std::is_same<E, __bf16>::value)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b) : FA_W64_CLOB);
  else
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b) : FA_W64_CLOB);
}
}  // namespace

template <typename E, int D>
__global__ void __launch_bounds__(256, 1) fa_bwd_dkdv_w64_kernel(const BwdK p) {
  using T = ElemTraits<E>;
  using V8 = typename T::v8;
  constexpr float kLog2eL = 1.4426950408889634f;
  constexpr int NW = 4, NT = NW * 64, KB = 2;
  constexpr int BNK = NW * 64;   // keys per workgroup
  constexpr int BMQ = 64;        // queries per streamed tile (two 32-row sub-blocks)
  constexpr int CPR = D / 8, ROW_BYTES = D * 2;
  constexpr int KS = D / 16, DB = D / 32;
  constexpr int VBLK_BYTES = BNK * ROW_BYTES;
  constexpr int QT_BYTES = BMQ * ROW_BYTES;
  // LDS: Q0 | Q1 | dO0 | dO1 | V block | aux0 aux1   (aux = BMQ x LSE*log2e followed by BMQ x delta)
  constexpr int OFF_Q = 0, OFF_DO = 2 * QT_BYTES, OFF_V = 4 * QT_BYTES, OFF_AUX = OFF_V + VBLK_BYTES;
  static_assert(D == 128, "head dim of this schedule: 128 (its 16 accumulator tiles are exactly the 256 accumulator registers)");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char FA_LDS* lds = (char FA_LDS*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, ki = lane & 31;

  int b, hk, n_block;
  if (p.k_list) {  // varlen: non-empty key blocks only, heaviest first
    if (!work_list_item(p.k_list, blockIdx.x, p.h_k, p.h_k, b, hk, n_block)) return;
  } else {
    const int w = xcd_interleave(blockIdx.x, p.k_units, p.k_unit_size, p.k_unit_hpx);
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
  if (p.seqused_k) sk = min(p.seqused_k[b], p.sk);   // overrides the cu_seqlens_k length, as in the forward (block_info.h:17-36), clamped to max_seqlen_k
  const int n0 = n_block * BNK;
  if (n0 >= sk) return;
  const int n1 = min(n0 + BNK, sk);
  const int shift = sk - sq;

  const E* __restrict__ kp = (const E*)p.k + k_boff + k_row0 * p.k_rs + (int64_t)hk * p.k_hs;
  const E* __restrict__ vp = (const E*)p.v + v_boff + k_row0 * p.v_rs + (int64_t)hk * p.v_hs;

  // query range that can see this key block
  int q_lo = 0, q_hi = sq - 1;
  if (p.wr >= 0) q_lo = max(0, n0 - shift - p.wr);
  if (p.wl >= 0) q_hi = min(sq - 1, n1 - 1 - shift + p.wl);
  const int m_lo = q_lo / BMQ;
  const int nm = (q_hi >= q_lo) ? (q_hi / BMQ + 1 - m_lo) : 0;  // tiles per query head
  const int n_items = nm * p.hk_ratio;

  // this wave's keys: key block kb = keys wk0 + 32 kb .. + 31, lane = key
  const int wk0 = n0 + wave * 64;

  // K fragments (B operand of S = Q.K^T): lane = key, 8 consecutive d per k-step
  V8 kf[KB][KS];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    const int key = wk0 + 32 * kb + ki;
    const E* krow = kp + (int64_t)key * p.k_rs + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kf[kb][ks] = bitcast_u32x4<V8>(ld_global_16B(krow + 16 * ks, key < sk));
  }
  // V block -> LDS (B operand of dP = dO.V^T, re-read per k-step)
  {
    constexpr int LDV = (BNK * CPR) / NT;
#pragma unroll
    for (int i = 0; i < LDV; ++i) {
      const int idx = tid + i * NT;
      const int row = idx / CPR, ch = idx % CPR;
      const u32x4 x = ld_global_16B(vp + (int64_t)(n0 + row) * p.v_rs + ch * 8, n0 + row < sk);
      *(u32x4 FA_LDS*)(lds + OFF_V + tile_off<D>(row, ch)) = x;
    }
  }

  // streamed tiles: Q / dO by LDS-DMA with the swizzle on the source side, LSE / delta through a register (see fa_bwd.hip)
  constexpr int RPD = 1024 / ROW_BYTES;
  constexpr int NDMA = (BMQ * ROW_BYTES) / 1024;
  constexpr int DPW = NDMA / NW;
  static_assert(NDMA % NW == 0 && DPW >= 1, "tile does not divide over the waves");
  float aux_reg = 0.f;  // threads [0,BMQ): LSE*log2e of row tid; [BMQ,2BMQ): delta of row tid-BMQ
  // Item it = (query head ih of the group, query tile im): walked with two counters -- a division per use (item -> head, tile)
  // costs ~100 cycles of mixed scalar / vector code, and nothing hides it with one wave per SIMD (the loop skeleton alone
  // measured 984 us of a 2.8 ms kernel with divisions in it: profiles/r02_bwd_schedules.txt).
  auto load_item = [&](int h, int m0, int buf) {
    const E* qp = (const E*)p.q + q_boff + q_row0 * p.q_rs + (int64_t)h * p.q_hs;
    const E* dop = (const E*)p.dout + do_boff + q_row0 * p.do_rs + (int64_t)h * p.do_hs;
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      const int idx = wave * DPW + i;
      const int row = idx * RPD + lane / CPR;
      const int pc = lane % CPR;
      const int grow = min(m0 + row, sq - 1);
      const int c = pc ^ swz16<D>(row);
      lds_dma_16B(qp + (int64_t)grow * p.q_rs + c * 8, lds + OFF_Q + buf * QT_BYTES + idx * 1024);
      lds_dma_16B(dop + (int64_t)grow * p.do_rs + c * 8, lds + OFF_DO + buf * QT_BYTES + idx * 1024);
    }
    if (tid < 2 * BMQ) {
      const int r = tid & (BMQ - 1);
      const bool ok = (m0 + r) < sq;
      const int64_t base = p.cu_q ? ((int64_t)h * p.total_q + q_row0) : (((int64_t)b * p.h + h) * p.sq);
      const float* src = (tid < BMQ ? p.lse : p.delta) + base + m0 + r;
      const float x = ok ? *src : 0.f;
      aux_reg = (tid < BMQ) ? (ok ? x * kLog2eL : INFINITY) : x;  // rows past the end: LSE = +inf => P = 0
    }
  };
  auto store_item = [&](int buf) {
    if (tid < 2 * BMQ) *(float FA_LDS*)(lds + OFF_AUX + (buf * 2 * BMQ + tid) * 4) = aux_reg;
  };

  // per-lane LDS read offsets (fa_bwd.hip): row-major fragment of k-step ks of row ki = k0 ^ (ks << 5); transposed operand
  // block db = tr_base[s] ^ (db << 6)
  const int rswz = swz16<D>(ki);
  const int k0 = ki * ROW_BYTES + ((hi ^ rswz) << 4);
  const int kv0 = k0 + OFF_V + wave * 64 * ROW_BYTES;
  const int tr_i = lane & 15, tr_half = (lane >> 4) & 1, tr_rr = tr_i >> 2, tr_cc = tr_i & 3;
  int tr_base[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int row = 8 * s + 4 * hi + tr_rr;
    tr_base[s] = tile_off<D>(row, 2 * tr_half + (tr_cc >> 1)) + (tr_cc & 1) * 8;
  }
  const int aux_lane = OFF_AUX + 4 * hi * 4;

  acc_zero_range<0>(std::make_integer_sequence<int, 2 * KB * DB * 16>{});   // dV^T / dK^T tiles (mfma_tile_acc)

  const float cs = p.scale_log2;

  if (n_items > 0) {
    load_item(hk * p.hk_ratio, m_lo * BMQ, 0);
    store_item(0);
  }
  lds_dma_wait_all();
  __syncthreads();

  auto unit_active = [&](bool item_exists, int m0, int qb, int kb) __attribute__((always_inline)) {
    return item_exists && ds_tile_active(m0 + 32 * qb, wk0 + 32 * kb, sq, sk, shift, p.wl, p.wr);
  };

  f32x16 sA, dpA, sB, dpB;   // S / dP of the unit whose P and dS are being formed, and of the unit being contracted
  V8 pfrag[2], dsfrag[2];    // P / dS of the current unit between its SM and P2
#pragma unroll
  for (int r = 0; r < 16; ++r) { sA[r] = 0.f; dpA[r] = 0.f; sB[r] = 0.f; dpB[r] = 0.f; }

  // causal / window mask of a unit on accumulator coordinates (rows = queries acc_row(r, hi), column = this lane's key)
  auto apply_mask = [&](f32x16& s, int q0, int kb) __attribute__((always_inline)) {
    const int my_key = wk0 + 32 * kb + ki;
    const int rel_lo = (p.wr >= 0) ? (my_key - shift - p.wr - q0 - 4 * hi) : -(1 << 30);
    const int rel_hi = (p.wl >= 0) ? (my_key - shift + p.wl - q0 - 4 * hi) : (1 << 30);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int off = acc_row(r, 0);
      s[r] = ((off >= rel_lo) && (off <= rel_hi)) ? s[r] : -INFINITY;
    }
  };
  auto unit_needs_mask = [&](int q0, int kb) __attribute__((always_inline)) {
    const int k_lo = wk0 + 32 * kb, k_hi = min(k_lo + 31, sk - 1);
    bool m = false;
    if (p.wr >= 0) m = m || (k_hi > q0 + shift + p.wr);
    if (p.wl >= 0) m = m || (k_lo < q0 + 31 + shift - p.wl);
    return m;
  };

  // One pipeline step: P1 of the NEXT unit (buffer bufN, sub-block qbN, key block kbN) into (sN, dpN), with the SM arithmetic of
  // the CURRENT unit (aux buffer bufC, sub-block qbC; scores sC / dpC, already masked) placed one accumulator element per MFMA
  // gap.  DO_P1 / DO_SM switch the halves off (an inactive neighbour under a causal / window mask, first and last step).
  // Slot j: LDS operand reads two slots ahead, MFMA j (S k-step j/2 for even j, dP k-step j/2 for odd j), element j of P / dS.
  auto step = [&](auto doP1c, auto doSMc, auto bufNc, auto qbNc, auto kbNc, f32x16& sN, f32x16& dpN, auto bufCc, auto qbCc,
                  const f32x16& sC, const f32x16& dpC) __attribute__((always_inline)) {
    constexpr bool DO_P1 = decltype(doP1c)::value != 0, DO_SM = decltype(doSMc)::value != 0;
    constexpr int bufN = decltype(bufNc)::value, qbN = decltype(qbNc)::value, kbN = decltype(kbNc)::value;
    constexpr int bufC = decltype(bufCc)::value, qbC = decltype(qbCc)::value;
    constexpr int QN = OFF_Q + bufN * QT_BYTES + qbN * 32 * ROW_BYTES, DON = OFF_DO + bufN * QT_BYTES + qbN * 32 * ROW_BYTES;
    constexpr int VN = kbN * 32 * ROW_BYTES;
    constexpr int AUXC = (bufC * 2 * BMQ + qbC * 32) * 4;
    constexpr int NOPS = 2 * KS, PF = FA_DKDV64_PF, RB = (PF + 1) / 2 + 1;
    u32x4 ra[PF], rb[RB];
    f32x4 l4[2], d4[2];
    auto rd = [&](int j) __attribute__((always_inline)) {
      const int ks = j >> 1;
      if ((FA_DKDV64_ABL & 4) && j >= PF) return;
      if ((j & 1) == 0) {
        ra[j % PF] = *(const u32x4 FA_LDS*)(lds + QN + (k0 ^ (ks << 5)));
      } else {
        ra[j % PF] = *(const u32x4 FA_LDS*)(lds + DON + (k0 ^ (ks << 5)));
        rb[ks % RB] = *(const u32x4 FA_LDS*)(lds + VN + (kv0 ^ (ks << 5)));
      }
    };
    auto rd_aux = [&](int g) __attribute__((always_inline)) {
      l4[g & 1] = *(const f32x4 FA_LDS*)(lds + aux_lane + AUXC + 8 * g * 4);
      d4[g & 1] = *(const f32x4 FA_LDS*)(lds + aux_lane + AUXC + 8 * g * 4 + BMQ * 4);
    };
    auto elem = [&](int r) __attribute__((always_inline)) {
      const int g = r >> 2, j = r & 3;
      if (FA_DKDV64_ABL & 16) return;
      const float x = __builtin_fmaf(sC[r], cs, -l4[g & 1][j]);
      const float pv = (FA_DKDV64_ABL & 2) ? x : fast_exp2(x);
      const float dsv = pv * (dpC[r] - d4[g & 1][j]);
      pfrag[r >> 3][r & 7] = (E)pv;
      dsfrag[r >> 3][r & 7] = (E)dsv;
    };
    if constexpr (DO_P1) {
#pragma unroll
      for (int j = 0; j < PF - 1; ++j) rd(j);
    }
    if constexpr (DO_SM) rd_aux(0);
    constexpr int NSLOT = DO_P1 ? NOPS : 16;
    constexpr int EPS = 16 / NSLOT;   // elements per slot (1 at D = 128, 2 at D = 64)
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      if constexpr (DO_P1) {
        if (j + PF - 1 < NOPS) rd(j + PF - 1);
      }
      if constexpr (DO_SM) {
        const int r0 = j * EPS;
        if ((r0 & 3) == 0 && (r0 >> 2) + 1 < 4) rd_aux((r0 >> 2) + 1);   // next group's LSE / delta, four elements ahead
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DO_P1) {
        const int ks = j >> 1;
        if ((FA_DKDV64_ABL & 32) && j >= 2) { }
        else if (j == 0) mfma_v_first<E>(sN, ra[j % PF], __builtin_bit_cast(u32x4, kf[kbN][ks]));
        else if (j == 1) mfma_v_first<E>(dpN, ra[j % PF], rb[ks % RB]);
        else if ((j & 1) == 0) mfma_v_acc<E>(sN, ra[j % PF], __builtin_bit_cast(u32x4, kf[kbN][ks]));
        else mfma_v_acc<E>(dpN, ra[j % PF], rb[ks % RB]);
      }
      if constexpr (DO_SM) {
        if constexpr (DO_P1 && (FA_DKDV64_ABL & 1)) {
          // ablation: the vector work after all the MFMAs instead of in their gaps
        } else {
#pragma unroll
          for (int e = 0; e < EPS; ++e) elem(j * EPS + e);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (DO_P1 && !DO_SM) mfma_drain_v(sN, dpN);   // (no P2 segment follows before the vector ALU reads these)
    if constexpr (DO_SM && DO_P1 && (FA_DKDV64_ABL & 1)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if ((r & 3) == 0) rd_aux(r >> 2);
        elem(r);
      }
    }
  };

  // P2 of a unit: dV^T[d][key] += dO^T[d][query] . P[query][key] ;  dK^T[d][key] += Q^T[d][query] . dS[query][key]
  // op i: source = dO (even) / Q (odd), d-block (i >> 1) % DB, query half t = i / (2 * DB); transpose reads two ops ahead
  auto p2 = [&](auto bufc, auto qbc, auto kbc) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value, qb = decltype(qbc)::value, kb = decltype(kbc)::value;
    constexpr int QB_OFF = OFF_Q + buf * QT_BYTES, DOB_OFF = OFF_DO + buf * QT_BYTES, sub = qb * 32 * ROW_BYTES;
    constexpr int NOPS = 4 * DB, PFT = FA_DKDV64_PFT;
    s16x4 tlo[PFT], thi[PFT];
    auto rd = [&](int i) __attribute__((always_inline)) {
      const int db = (i >> 1) % DB, t = i / (2 * DB);
      if ((FA_DKDV64_ABL & 4) && i >= PFT) return;
      const int base = ((i & 1) ? QB_OFF : DOB_OFF) + sub + 16 * t * ROW_BYTES;
      tlo[i % PFT] = lds_read_tr16(lds + base + (tr_base[0] ^ (db << 6)));
      thi[i % PFT] = lds_read_tr16(lds + base + (tr_base[1] ^ (db << 6)));
    };
#pragma unroll
    for (int i = 0; i < PFT - 1; ++i) rd(i);
    static_for<NOPS>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      if (i + PFT - 1 < NOPS) rd(i + PFT - 1);
      __builtin_amdgcn_sched_barrier(0);
      constexpr int db = (i >> 1) % DB, t = i / (2 * DB);
      if constexpr ((FA_DKDV64_ABL & 64) && i >= 2) return;
      const u32x4 a = __builtin_bit_cast(u32x4, combine_tr<V8>(tlo[i % PFT], thi[i % PFT]));
      if constexpr ((i & 1) == 0) mfma_tile_acc<E, (2 * kb) * DB + db>(a, __builtin_bit_cast(u32x4, pfrag[t]));
      else mfma_tile_acc<E, (2 * kb + 1) * DB + db>(a, __builtin_bit_cast(u32x4, dsfrag[t]));
    });
  };

  using Y = IC64<1>;
  using Nn = IC64<0>;
  // current unit (bufC, qbC, kbC; scores in sC / dpC) -> its P / dS and its dV / dK products; next unit's S / dP into sN / dpN
  auto advance = [&](bool actC, bool actN, int q0C, auto bufCc, auto qbCc, auto kbCc, f32x16& sC, f32x16& dpC, auto bufNc, auto qbNc,
                     auto kbNc, f32x16& sN, f32x16& dpN) __attribute__((always_inline)) {
    constexpr int kbC = decltype(kbCc)::value;
    if (actC && unit_needs_mask(q0C, kbC)) apply_mask(sC, q0C, kbC);
    if (actC && actN) step(Y{}, Y{}, bufNc, qbNc, kbNc, sN, dpN, bufCc, qbCc, sC, dpC);
    else if (actN) step(Y{}, Nn{}, bufNc, qbNc, kbNc, sN, dpN, bufCc, qbCc, sC, dpC);
    else if (actC) step(Nn{}, Y{}, bufNc, qbNc, kbNc, sN, dpN, bufCc, qbCc, sC, dpC);
    if (actC) p2(bufCc, qbCc, kbCc);
  };

  int im = 0, ih = 0;   // tile / head counters of the current item
  auto item = [&](auto curc, int it) __attribute__((always_inline)) {
    constexpr int cur = decltype(curc)::value;
    using C = IC64<cur>;
    using N = IC64<cur ^ 1>;
    using Z0 = IC64<0>;
    using Z1 = IC64<1>;
    const bool has_next = it + 1 < n_items;
    const int q0 = (m_lo + im) * BMQ;
    if (++im == nm) { im = 0; ++ih; }
    const int q0n = (m_lo + im) * BMQ;      // next item
    if (has_next && !((FA_DKDV64_ABL & 128) && it > 0)) load_item(hk * p.hk_ratio + ih, q0n, cur ^ 1);  // (after the previous tile's closing barrier: every wave is done with that buffer)
    const bool a00 = unit_active(true, q0, 0, 0), a01 = unit_active(true, q0, 0, 1), a10 = unit_active(true, q0, 1, 0), a11 = unit_active(true, q0, 1, 1);
    const bool n00 = unit_active(has_next, q0n, 0, 0);
    advance(a00, a01, q0, C{}, Z0{}, Z0{}, sA, dpA, C{}, Z0{}, Z1{}, sB, dpB);
    advance(a01, a10, q0, C{}, Z0{}, Z1{}, sB, dpB, C{}, Z1{}, Z0{}, sA, dpA);
    advance(a10, a11, q0 + 32, C{}, Z1{}, Z0{}, sA, dpA, C{}, Z1{}, Z1{}, sB, dpB);
    if (has_next) store_item(cur ^ 1);
    if (!(FA_DKDV64_ABL & 8)) {
      lds_dma_wait_all();   // the next tile (issued at the top) has landed ...
      __syncthreads();      // ... for every wave
    }
    advance(a11, n00, q0 + 32, C{}, Z1{}, Z1{}, sB, dpB, N{}, Z0{}, Z0{}, sA, dpA);
    if (!(FA_DKDV64_ABL & 8)) __syncthreads();      // every wave is through with this tile: the next DMA may overwrite it
  };
  if (unit_active(n_items > 0, m_lo * BMQ, 0, 0)) step(Y{}, Nn{}, IC64<0>{}, IC64<0>{}, IC64<0>{}, sA, dpA, IC64<0>{}, IC64<0>{}, sB, dpB);
  for (int it = 0; it < n_items; it += 2) {
    item(IC64<0>{}, it);
    if (it + 1 < n_items) item(IC64<1>{}, it + 1);
  }

  // epilogue: dK = scale * acc, dV = acc, through the freed Q/dO buffers (whole-row stores); every key row of the block that
  // exists is written, zeros included (empty-sequence contract of the CK tests)
  char FA_LDS* stage = lds + wave * 32 * (ROW_BYTES + 16);
  mfma_drain_acc();
  static_for<KB>([&](auto kbc) __attribute__((always_inline)) {
    constexpr int kb = decltype(kbc)::value;
    const int kb0 = wk0 + 32 * kb;
    if (kb0 < sk) {
      E* dktile = (E*)p.dk + dk_boff + (k_row0 + kb0) * p.dk_rs + (int64_t)hk * p.dk_hs;
      E* dvtile = (E*)p.dv + dv_boff + (k_row0 + kb0) * p.dv_rs + (int64_t)hk * p.dv_hs;
      f32x16 t[DB];
      static_for<DB>([&](auto dbc) __attribute__((always_inline)) { acc_read_tuple<16 * ((2 * kb + 1) * DB + decltype(dbc)::value)>(t[decltype(dbc)::value]); });
      store_tile_via_lds<E, D>(stage, t, p.scale, dktile, p.dk_rs, sk - kb0, lane);
      static_for<DB>([&](auto dbc) __attribute__((always_inline)) { acc_read_tuple<16 * ((2 * kb) * DB + decltype(dbc)::value)>(t[decltype(dbc)::value]); });
      store_tile_via_lds<E, D>(stage, t, 1.f, dvtile, p.dv_rs, sk - kb0, lane);
    }
  });
}

template <typename E, int D>
static int launch_dkdv_w64_t(const BwdK& p, hipStream_t stream) {
  constexpr int smem = 256 * D * 2 + 4 * 64 * D * 2 + 4 * 64 * 4;
  auto kern = fa_bwd_dkdv_w64_kernel<E, D>;
  static std::atomic<unsigned long long> attr_mask{0};
  if (ensure_dyn_lds(attr_mask, (const void*)kern, smem) != 0) return -1;
  const long long total = p.k_list ? (long long)p.k_bound * p.h_k : units_grid(p.k_units, p.k_unit_size);
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), smem, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// -2 = not covered by this schedule (features, dS spill, head dims other than 128): the caller runs fa_bwd_dkdv_kernel
int launch_bwd_dkdv_w64(const BwdK& p, int dtype_bf16, int d, hipStream_t stream) {
  if (p.softcap > 0.f || p.alibi != nullptr || p.rng != nullptr || p.ds_ws != nullptr || d != 128) return -2;
  return dtype_bf16 ? launch_dkdv_w64_t<__bf16, 128>(p, stream) : launch_dkdv_w64_t<_Float16, 128>(p, stream);
}

}  // namespace fa
