// Forward attention, software-pipelined ("interleaved") schedule for gfx950 -- same arithmetic, layouts and
// LDS swizzles as fa_fwd.hip, different instruction stream.
//
// Why: on one SIMD the matrix pipe takes a v_mfma_f32_32x32x16 every 32 cycles and, while it is busy, the
// issue port sustains about five plain VALU ops (or three transcendentals) per MFMA slot -- from either of
// the SIMD's two waves, but only if they are offered in that mix (profiles/r01_ubench_mfma_valu_overlap.txt).
// A wave that alternates "16 MFMAs" and "150 VALU ops" offers the wrong mix twice.  Here every wave runs a
// steady stream of 32-key steps, each one basic block containing
//       8 MFMAs   S_{i+1} = K_{i+1}.Q^T          (next step's scores)
//       8 MFMAs   O += V_{i-1}^T.P_{i-1}          (previous step's probabilities)
//    ~65 VALU     P_i = exp2(S_i*c - m*c), row sums, row max of S_{i+1}, bf16 packing of P_i
// i.e. ~4 VALU per MFMA with no dependence between the three strands inside a step.
// Online-softmax bookkeeping is arranged so the only data-dependent branch (rescale when a row maximum grew
// by more than `rescale_thr`) sits at the step boundary, moves m and rescales l there, and rescales O one
// step later (after P_i.V, computed at the old scale, has been added), so it never touches the packed P_i.
//
// One iteration of the tile loop = two steps = one 64-key K tile and the previous V tile, double-buffered
// in LDS exactly as in the lock-step kernel (64 KB), one barrier per iteration.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

// The timing ablations of the steady-state step and the occupancy experiment (extra dynamic LDS) live in experiments/ablations/fa_fwd_il.patch (tools/ablate_fwd.sh).
#define FA_IL_AHEAD 4  // LDS operand reads issued this many MFMA slots ahead (LDS latency ~130-200 cycles, slot ~35)

#include "fa_device.h"
#include "fa_kernel_params.h"
#include "fa_launch.h"

namespace fa {

template <int D> FA_DEVINL constexpr int k_swz_il(int row) { return D == 128 ? (row & 15) : ((row >> 1) & 7); }
template <int D> FA_DEVINL constexpr int v_swz_il(int row) { return D == 128 ? (row & 3) : ((row >> 1) & 1); }

template <int N> using ICi = std::integral_constant<int, N>;

template <typename E, int D, int NW, int SCHED>
__global__ void __launch_bounds__(NW * 64, 2) fa_fwd_il_kernel(const FwdK p) {
  // SCHED = operand-read lead (MFMA slots) of the hand-placed steady-state step; 0 = compiler-ordered step
  constexpr int sched_mode = SCHED;
  constexpr bool QLDS = (NW == 8);  // 8 waves: Q block in LDS (1 workgroup/CU); 4 waves: Q fragments in registers (2 workgroups/CU)
  using T = ElemTraits<E>;
  using V8 = typename T::v8;
  using V4 = typename T::v4;
  constexpr int BM = NW * 32, BN = 64, CPR = D / 8, NT = NW * 64;
  constexpr int ROW_BYTES = D * 2;
  constexpr int TILE_BYTES = BN * ROW_BYTES;
  constexpr int KS = D / 16;
  constexpr int DB = D / 32;
  static_assert(D == 64 || D == 128, "head dims built natively: 64, 128");
  constexpr float kLn2 = 0.6931471805599453f;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char FA_LDS* lds = (char FA_LDS*)smem;  // K0 | K1 | V0 | V1 | Q (BM rows, K-style swizzle)
  constexpr int Q_OFF = 4 * TILE_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, qi = lane & 31;

  int b, h, m_block;
  if (p.work_list) {  // varlen: non-empty blocks only, heaviest first (fa_varlen_schedule_kernel)
    if (!work_list_item(p.work_list, blockIdx.x, p.h, p.h_k, b, h, m_block)) return;
  } else {
    const int w = xcd_interleave(blockIdx.x, p.n_units, p.unit_size, p.unit_hpx);
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
  const int bkv = p.kv_batch_idx ? p.kv_batch_idx[b] : b;  // KV-cache row of this batch entry
  int64_t q_boff = (int64_t)b * p.q_bs, k_boff = (int64_t)bkv * p.k_bs, v_boff = (int64_t)bkv * p.v_bs, o_boff = (int64_t)b * p.o_bs;
  if (p.block_table) { k_boff = 0; v_boff = 0; }  // paged cache: the page index supplies the first-dimension offset
  if (p.cu_q) { const int c0 = p.cu_q[b]; sq = p.cu_q[b + 1] - c0; q_row0 = c0; q_boff = 0; o_boff = 0; }
  if (p.seqused_q) sq = min(sq, p.seqused_q[b]);
  if (p.cu_k) { const int c0 = p.cu_k[b]; sk = p.cu_k[b + 1] - c0; k_row0 = c0; k_boff = 0; v_boff = 0; }
  if (p.block_table) k_row0 = 0;  // paged K/V: the page table supplies the rows, cu_seqlens_k only the lengths
  if (p.seqused_k) sk = min(p.seqused_k[b] + p.seqused_add, (p.cu_k && !p.block_table) ? sk : p.sk);  // keys in use, never beyond the addressable capacity; inside a packed batch never beyond the entry's slot (as the backward: include/fa_gfx950.h)
  if (p.leftpad_k) {  // a left-padded sequence starts at row leftpad_k[b] (reference block_info.h:17-36)
    const int lp = p.leftpad_k[b];
    sk = max(0, sk - lp);
    k_row0 += lp;
  }
  const int m0 = m_block * BM;
  if (m0 >= sq) return;

  const E* __restrict__ qp = (const E*)p.q + q_boff + q_row0 * p.q_rs + (int64_t)h * p.q_hs;
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

  const int w_row0 = m0 + wave * 32;
  const int w_row1 = min(w_row0 + 31, sq - 1);
  const bool wave_valid = w_row0 < sq;
  const int w_kmax = (p.wr >= 0) ? min(sk - 1, w_row1 + shift + p.wr) : sk - 1;
  const int w_kmin = (p.wl >= 0) ? max(0, w_row0 + shift - p.wl) : 0;
  const int w_full_hi = (p.wr >= 0) ? min(sk - 1, w_row0 + shift + p.wr) : sk - 1;
  const int w_full_lo = (p.wl >= 0) ? (w_row1 + shift - p.wl) : 0;
  const int my_row = w_row0 + qi;
  const bool row_valid = my_row < sq;
  const int lim_hi = (p.wr >= 0) ? min(sk - 1, my_row + shift + p.wr) : sk - 1;
  const int lim_lo = (p.wl >= 0) ? (my_row + shift - p.wl) : 0;

  const float cs = p.scale_log2;
  const float thr = p.rescale_thr;

  // step i covers keys key_base + 32 i .. +31
  auto step_active = [&](int i) __attribute__((always_inline)) {
    const int k0 = key_base + 32 * i;
    return wave_valid && (i >= 0) && (i < n_steps) && (k0 <= w_kmax) && (k0 + 31 >= w_kmin);
  };
  auto step_needs_mask = [&](int i) __attribute__((always_inline)) {
    const int k0 = key_base + 32 * i;
    return (k0 + 31 > w_full_hi) || (k0 < w_full_lo);
  };

  // ---- K/V tiles go global -> LDS by DMA (global_load_lds, 1 KiB per wave instruction): no staging registers,
  // no ds_write pass.  The destination is lane-linear, so the XOR swizzles are applied to the per-lane SOURCE
  // chunk.  Rows past the last key are clamped to the last key (finite data; their scores are masked to -inf).
  constexpr int RPD = 1024 / ROW_BYTES;            // tile rows per DMA instruction
  constexpr int NDMA = TILE_BYTES / 1024;          // DMA instructions per tile
  constexpr int DPW = NDMA / NW;                   // per wave
  static_assert(NDMA % NW == 0 && DPW >= 1, "tile does not divide over the waves");
  const int d_row = lane / CPR, d_pc = lane % CPR;  // row inside the DMA piece, physical 16-B chunk
  unsigned koff_l[DPW], voff_l[DPW];                // per-lane source byte offsets inside a full tile
#pragma unroll
  for (int i = 0; i < DPW; ++i) {
    const int row = (wave * DPW + i) * RPD + d_row;
    const int kc = d_pc ^ k_swz_il<D>(row);
    const int vc = ((((d_pc >> 2) ^ v_swz_il<D>(row)) << 2) | (d_pc & 3));
    koff_l[i] = (unsigned)(row * (int)p.k_rs + kc * 8) * 2u;
    voff_l[i] = (unsigned)(row * (int)p.v_rs + vc * 8) * 2u;
  }
  auto dma_tile = [&](auto isvc, int buf, int t) __attribute__((always_inline)) {  // t relative to n_min
    constexpr bool ISV = decltype(isvc)::value != 0;
    const int n = n_min + t;
    const int64_t rs = ISV ? p.v_rs : p.k_rs;
    int64_t row_off = (int64_t)n * BN * rs;
    if (p.block_table) {  // paged KV cache: page block_table[b][n*BN / page]; a 64-key tile never straddles two pages
      const int key0 = n * BN;
      const int page = key0 / p.page_size;
      const int blk = p.block_table[(int64_t)b * p.block_table_bs + page];
      row_off = (int64_t)blk * (ISV ? p.v_bs : p.k_bs) + (int64_t)(key0 - page * p.page_size) * rs;
    }
    const char* base = (const char*)((ISV ? vp : kp) + row_off);
    char FA_LDS* dst = lds + (ISV ? 2 + buf : buf) * TILE_BYTES + wave * DPW * 1024;
    if (n * BN + BN <= sk) {
#pragma unroll
      for (int i = 0; i < DPW; ++i)
        lds_dma_16B(base + (ISV ? voff_l[i] : koff_l[i]), dst + i * 1024);
    } else {  // last, partial tile
#pragma unroll
      for (int i = 0; i < DPW; ++i) {
        const int row = (wave * DPW + i) * RPD + d_row;
        const int grow = min(n * BN + row, sk - 1) - n * BN;
        const int c = ISV ? ((((d_pc >> 2) ^ v_swz_il<D>(row)) << 2) | (d_pc & 3)) : (d_pc ^ k_swz_il<D>(row));
        const char* src = base + ((int64_t)grow * rs + c * 8) * 2;
        lds_dma_16B(src, dst + i * 1024);
      }
    }
  };

  // Q block -> LDS once (its fragments are re-read per step instead of living in 32 registers), or, for
  // 4-wave workgroups, Q fragments in registers so that two workgroups fit one CU's LDS.
  V8 qreg[QLDS ? 1 : KS];
  constexpr int QSTAGE = QLDS ? Q_OFF : 2 * TILE_BYTES;  // 4 waves: staged in the (still unused) V buffers, then to registers
  {
    const int64_t rs = p.q_rs;
    constexpr int QDMA = (BM * ROW_BYTES) / 1024 / NW;  // DMA instructions per wave
    static_assert(QLDS || BM * ROW_BYTES <= 2 * TILE_BYTES, "Q block does not fit the V buffers");
#pragma unroll
    for (int i = 0; i < QDMA; ++i) {
      const int row = (wave * QDMA + i) * RPD + d_row;
      const int grow = min(m0 + row, sq - 1);
      const int c = d_pc ^ k_swz_il<D>(row);
      lds_dma_16B(qp + (int64_t)grow * rs + c * 8, lds + QSTAGE + (wave * QDMA + i) * 1024);
    }
  }
  const int qbase = QSTAGE + (wave * 32 + qi) * ROW_BYTES + ((hi ^ k_swz_il<D>(qi)) << 4);

  // per-lane LDS read bases: K fragment of k-step ks at kbase ^ (ks << 5); V d-block db at vbase ^ (db << 6)
  const int kbase = qi * ROW_BYTES + ((hi ^ k_swz_il<D>(qi)) << 4);
  const int tr_i = lane & 15, tr_half = (lane >> 4) & 1;
  const int tr_rr = tr_i >> 2, tr_cc = tr_i & 3;
  const int vbase = (4 * hi + tr_rr) * ROW_BYTES + (v_swz_il<D>(tr_rr) << 6) + tr_half * 32 + tr_cc * 8;

  f32x16 o_acc[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[db][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  float o_lag = 1.f;   // factor O still has to be multiplied by (applied one step after the decision)
  f32x16 sA, sB;       // scores of the current / next step (roles swap every step)
  V8 pfA[2], pfB[2];   // packed P of the previous / current step
#pragma unroll
  for (int r = 0; r < 16; ++r) { sA[r] = 0.f; sB[r] = 0.f; }
  bool have_cur = false, have_prev = false;

  // S^T of one 32-key half tile: 8 (KS) MFMAs, K fragments PF k-steps ahead
  auto qk_half = [&](f32x16& s, int kb_lane, auto halfc) __attribute__((always_inline)) {
    constexpr int half = decltype(halfc)::value;
    const char FA_LDS* kbuf = lds + half * 32 * ROW_BYTES;
    constexpr int PF = 3;
    u32x4 kfrag[PF], qfrag[PF];
#pragma unroll
    for (int ks = 0; ks < PF - 1 && ks < KS; ++ks) {
      kfrag[ks % PF] = *(const u32x4 FA_LDS*)(kbuf + (kb_lane ^ (ks << 5)));
      if constexpr (QLDS) qfrag[ks % PF] = *(const u32x4 FA_LDS*)(lds + (qbase ^ (ks << 5)));
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int nx = ks + PF - 1;
      if (nx < KS) {
        kfrag[nx % PF] = *(const u32x4 FA_LDS*)(kbuf + (kb_lane ^ (nx << 5)));
        if constexpr (QLDS) qfrag[nx % PF] = *(const u32x4 FA_LDS*)(lds + (qbase ^ (nx << 5)));
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch above this k-step's MFMA
      f32x16 c = s;
      if (ks == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
      }
      s = T::mfma(bitcast_u32x4<V8>(kfrag[ks % PF]), QLDS ? bitcast_u32x4<V8>(qfrag[ks % PF]) : qreg[QLDS ? 0 : ks], c);
    }
  };
  // O^T += V^T(32 keys) . P^T : 2*DB MFMAs, transpose reads PFV MFMAs ahead
  auto pv_half = [&](const V8 (&pf)[2], int vb_lane, auto halfc) __attribute__((always_inline)) {
    constexpr int half = decltype(halfc)::value;
    const char FA_LDS* vbuf = lds + half * 32 * ROW_BYTES;
    constexpr int NOP = 2 * DB, PFV = 3;
    s16x4 vlo[PFV], vhi[PFV];
#pragma unroll
    for (int i = 0; i < PFV - 1 && i < NOP; ++i) {
      vlo[i % PFV] = lds_read_tr16(vbuf + (vb_lane ^ ((i % DB) << 6)) + (16 * (i / DB)) * ROW_BYTES);
      vhi[i % PFV] = lds_read_tr16(vbuf + (vb_lane ^ ((i % DB) << 6)) + (16 * (i / DB) + 8) * ROW_BYTES);
    }
#pragma unroll
    for (int i = 0; i < NOP; ++i) {
      const int nx = i + PFV - 1;
      if (nx < NOP) {
        vlo[nx % PFV] = lds_read_tr16(vbuf + (vb_lane ^ ((nx % DB) << 6)) + (16 * (nx / DB)) * ROW_BYTES);
        vhi[nx % PFV] = lds_read_tr16(vbuf + (vb_lane ^ ((nx % DB) << 6)) + (16 * (nx / DB) + 8) * ROW_BYTES);
      }
      __builtin_amdgcn_sched_barrier(0);
      o_acc[i % DB] = T::mfma(combine_tr<V8>(vlo[i % PFV], vhi[i % PFV]), pf[i / DB], o_acc[i % DB]);
    }
  };
  auto apply_mask = [&](f32x16& s, int i) __attribute__((always_inline)) {
    const int k0 = key_base + 32 * i;
    const int rel_hi = lim_hi - k0 - 4 * hi;
    const int rel_lo = lim_lo - k0 - 4 * hi;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int off = acc_row(r, 0);
      s[r] = ((off <= rel_hi) && (off >= rel_lo)) ? s[r] : -INFINITY;
    }
  };
  // Row max of the NEXT step's scores and the rescale decision.
  auto row_max_grow = [&](const f32x16& s_nxt, float& m_new) __attribute__((always_inline)) {
    float tmax = fmaxf(fmaxf(s_nxt[0], s_nxt[1]), s_nxt[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) tmax = fmaxf(fmaxf(tmax, s_nxt[r]), s_nxt[r + 1]);
    tmax = fmaxf(tmax, s_nxt[15]);
    tmax = half_max(tmax);
    m_new = fmaxf(m_run, tmax);
    return (m_new - m_run) * cs > thr;
  };
  // The decision taken at the end of step i (max of S_{i+1} grew by more than thr) moves m and rescales l
  // at once; P_i was already computed (and packed) at the old scale, so O is multiplied one step later, after
  // the product P_i.V has been accumulated at the old scale.  The branch therefore never touches P.
  auto rescale = [&](bool grow, float m_new) __attribute__((always_inline)) {
    const float m_upd = grow ? m_new : m_run;
    const float m_safe = (m_upd == -INFINITY) ? 0.f : m_upd;
    const float alpha = grow ? fast_exp2((m_run - m_safe) * cs) : 1.f;
    m_run = m_upd;
    l_run *= alpha;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o_acc[db][r] *= o_lag;
    o_lag = alpha;
  };
  auto need_rescale = [&](bool grow) __attribute__((always_inline)) { return grow || o_lag != 1.f; };

  int qa[KS];  // loop-invariant Q fragment addresses
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qa[ks] = qbase ^ (ks << 5);

  // Hand-placed fast step (SCHED >= 2; SCHED = how many slots ahead the operand reads run): the step is written as 16 (KS + 2*DB) slots, each = the LDS reads two
  // slots ahead, one MFMA, and that slot's share of the VALU work; slots are pinned with sched_barrier(0).
  //   slots 0 .. KS-1      : S_{i+1} += K.Q^T (k-step g)   +  exp/sum of 16/KS elements of S_i
  //   slots KS .. KS+2DB-1 : O += V^T.P_{i-1}  (op g)      +  the row-max tree of S_{i+1} (from slot KS+2 on,
  //                          when the last QK^T MFMA has retired)
  auto fast_step = [&](auto halfc, auto maskc, int i_nxt, const int (&ka)[KS], const int (&va)[DB], f32x16& s_cur,
                       f32x16& s_nxt, const V8 (&pf_prev)[2], V8 (&pf_cur)[2]) __attribute__((always_inline)) {
    constexpr int half = decltype(halfc)::value;
    constexpr bool MASK = decltype(maskc)::value != 0;  // step i_nxt straddles a mask boundary
    constexpr int NOP = 2 * DB, EPG = 16 / KS, AHEAD = (sched_mode >= 2 ? sched_mode : 2), RING = AHEAD + 1;
    constexpr int HOFF = half * 32 * ROW_BYTES;
    u32x4 kfr[RING], qfr[RING];
    s16x4 vlo[RING], vhi[RING];
    auto rd_kq = [&](int ks) __attribute__((always_inline)) {
      kfr[ks % RING] = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)(ka[ks] + HOFF);
      if constexpr (QLDS) qfr[ks % RING] = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)(qa[ks]);
    };
    auto rd_v = [&](int op) __attribute__((always_inline)) {
      const char FA_LDS* a0 = (const char FA_LDS*)(unsigned long)(unsigned)(va[op % DB] + HOFF + (16 * (op / DB)) * ROW_BYTES);
      vlo[op % RING] = lds_read_tr16(a0);
      vhi[op % RING] = lds_read_tr16(a0 + 8 * ROW_BYTES);
    };
    // operand reads run AHEAD slots in front of their MFMA over the whole 16-slot sequence
    auto rd_slot = [&](int slot) __attribute__((always_inline)) {
      if (slot < KS) rd_kq(slot);
      else if (slot < KS + NOP) rd_v(slot - KS);
    };
    const float neg_mc = (m_run == -INFINITY) ? 0.f : -m_run * cs;
    float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
    for (int g = 0; g < AHEAD; ++g) rd_slot(g);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < KS; ++g) {
      rd_slot(g + AHEAD);
      f32x16 c = s_nxt;
      if (g == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
      }
      s_nxt = T::mfma(bitcast_u32x4<V8>(kfr[g % RING]), QLDS ? bitcast_u32x4<V8>(qfr[g % RING]) : qreg[QLDS ? 0 : g], c);
#pragma unroll
      for (int e = 0; e < EPG; e += 2) {
        const int r = g * EPG + e;
        // (v_pk_fma_f32 for the scale/subtract pair measured 5 % slower than two v_fma_f32)
        const float p0 = fast_exp2(__builtin_fmaf(s_cur[r], cs, neg_mc));
        const float p1 = fast_exp2(__builtin_fmaf(s_cur[r + 1], cs, neg_mc));
        s_cur[r] = p0;
        s_cur[r + 1] = p1;
        ps0 += p0;
        ps1 += p1;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    l_run += ps0 + ps1;
    float tmax = -INFINITY;
    int rel_hi = 0, rel_lo = 0;
    if constexpr (MASK) {
      const int k0 = key_base + 32 * i_nxt;
      rel_hi = lim_hi - k0 - 4 * hi;
      rel_lo = lim_lo - k0 - 4 * hi;
    }
#pragma unroll
    for (int g = 0; g < NOP; ++g) {
      rd_slot(KS + g + AHEAD);
      o_acc[g % DB] = T::mfma(combine_tr<V8>(vlo[g % RING], vhi[g % RING]), pf_prev[g / DB], o_acc[g % DB]);
      if constexpr (MASK) {  // mask.h:172-203 predicate on the freshly produced scores, spread over the first slots
        if (g >= 1 && g < 1 + 2) {
#pragma unroll
          for (int r = (g - 1) * 8; r < (g - 1) * 8 + 8; ++r) {
            const int off = acc_row(r, 0);
            s_nxt[r] = ((off <= rel_hi) && (off >= rel_lo)) ? s_nxt[r] : -INFINITY;
          }
        }
      }
      if (g < 2) {  // P_i is final once its exponentials are done: pack it under the first PV slots
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) pf_cur[g][jj] = (E)s_cur[8 * g + jj];
      }
      if (g >= (MASK ? 3 : 2)) {  // row-max tree, spread over the remaining slots (2 values per max3)
        constexpr int G0 = MASK ? 3 : 2;
        constexpr int SLOTS = NOP - G0;
        constexpr int PER = (8 + SLOTS - 1) / SLOTS;  // max3 ops per slot
#pragma unroll
        for (int t = 0; t < PER; ++t) {
          const int q = (g - G0) * PER + t;
          if (q < 8) tmax = fmaxf(fmaxf(tmax, s_nxt[2 * q]), s_nxt[2 * q + 1]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    tmax = half_max(tmax);
    const float m_new = fmaxf(m_run, tmax);
    const bool grow = (m_new - m_run) * cs > thr;
    if (__any(need_rescale(grow))) rescale(grow, m_new);
  };

  // One pipeline step.  s_cur: scores of step i (masked, decision already taken) -> becomes P_i in place;
  // s_nxt: receives scores of step i+1; pf_prev: packed P_{i-1}; pf_cur: receives packed P_i.
  // K data of step i+1 at LDS offset KOFF, V data of step i-1 at VOFF.
  auto step = [&](auto fastc, auto halfc, int kb_lane, int vb_lane, int i, f32x16& s_cur, f32x16& s_nxt,
                  const V8 (&pf_prev)[2], V8 (&pf_cur)[2]) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fastc)::value != 0;
    const bool do_qk = FAST || step_active(i + 1);
    const bool do_sm = FAST || have_cur;
    const bool do_pv = FAST || have_prev;
    if (do_qk) qk_half(s_nxt, kb_lane, halfc);
    if (do_sm) {
      const float neg_mc = (m_run == -INFINITY) ? 0.f : -m_run * cs;
      float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = fast_exp2(__builtin_fmaf(s_cur[r], cs, neg_mc));
        const float p1 = fast_exp2(__builtin_fmaf(s_cur[r + 1], cs, neg_mc));
        s_cur[r] = p0;
        s_cur[r + 1] = p1;
        ps0 += p0;
        ps1 += p1;
      }
      l_run += ps0 + ps1;
    }
    if (do_pv) pv_half(pf_prev, vb_lane, halfc);
    {
      float m_new = m_run;
      bool grow = false;
      if (do_qk) {
        if (!FAST && step_needs_mask(i + 1)) apply_mask(s_nxt, i + 1);
        grow = row_max_grow(s_nxt, m_new);
      }
      if (__any(need_rescale(grow))) rescale(grow, m_new);
    }
    if (do_sm) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) pf_cur[t][jj] = (E)s_cur[8 * t + jj];
    }
    have_prev = do_sm;
    have_cur = do_qk;
  };
  // ---- prologue: K_0 into LDS ------------------------------------------------------------------------
  if (n_tiles > 0) dma_tile(ICi<0>{}, 0, 0);
  lds_dma_wait_all();
  __syncthreads();
  if constexpr (!QLDS) {  // Q fragments LDS -> registers (coalesced DMA instead of 16-byte loads at row stride), then free the V buffers
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qreg[ks] = bitcast_u32x4<V8>(*(const u32x4 FA_LDS*)(unsigned long)(unsigned)(qbase ^ (ks << 5)));
    __syncthreads();
  }

  // iteration u (0..n_tiles): steps 2u-1 and 2u read K_u (kbuf[u&1]) and V_{u-1} (vbuf[(u-1)&1]);
  // K_{u+1} and V_u are DMA'd during the iteration into the buffers it does not read.
  // Buffer selection is folded into the per-lane LDS bases by XOR (tile offsets do not overlap the lane bits).
  //
  // A wave's iterations split into generic head / steady-state middle / generic tail.  An iteration is "fast"
  // when steps 2u-2 .. 2u+1 are all active for this wave (pipeline full); activity is an interval of the step
  // index, so the fast iterations are one contiguous range [uf_lo, uf_hi] (different per wave -- every
  // iteration still has exactly one barrier).  Fast iterations whose steps straddle a mask boundary use the
  // same pipelined step with the mask predicate added.
  int uf_lo = 1, uf_hi = 0;
  if (wave_valid && n_tiles > 0) {
    const int a_lo = max(0, (w_kmin - key_base) >> 5);
    const int a_hi = min(n_steps - 1, (w_kmax - key_base) >> 5);
    uf_lo = (a_lo + 3) >> 1;
    uf_hi = (a_hi - 1) >> 1;
  }
  auto iter_head = [&](int u) __attribute__((always_inline)) {
    const int par = u & 1;
    if (u + 1 < n_tiles) dma_tile(ICi<0>{}, par ^ 1, u + 1);
    if (u < n_tiles) dma_tile(ICi<1>{}, par, u);
  };
  auto iter_tail = [&]() __attribute__((always_inline)) {
    lds_dma_wait_all();  // this wave's DMA pieces have landed ...
    __syncthreads();     // ... and everybody's are visible before the next iteration reads them
  };
  if (n_tiles > 0) {
    int u = 0;
    const int head_end = min(max(uf_lo, 0), n_tiles + 1);
    for (; u < head_end; ++u) {
      iter_head(u);
      const int kb_lane = kbase ^ ((u & 1) * TILE_BYTES);
      const int vb_lane = vbase ^ ((2 + ((u & 1) ^ 1)) * TILE_BYTES);
      step(ICi<0>{}, ICi<0>{}, kb_lane, vb_lane, 2 * u - 1, sA, sB, pfA, pfB);
      step(ICi<0>{}, ICi<1>{}, kb_lane, vb_lane, 2 * u, sB, sA, pfB, pfA);
      iter_tail();
    }
    // steady state, split so that every loop body is a single straight-line variant:
    //   [uf_lo, um_lo) masked (window's left edge) | [um_lo, um_hi] unmasked | (um_hi, uf_hi] masked (causal diagonal)
    auto fast_iter = [&](auto maskc, int uu) __attribute__((always_inline)) {
      iter_head(uu);
      const int kb_lane = kbase ^ ((uu & 1) * TILE_BYTES);
      const int vb_lane = vbase ^ ((2 + ((uu & 1) ^ 1)) * TILE_BYTES);
      if constexpr (sched_mode >= 2) {
        int ka[KS], va[DB];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) ka[ks] = kb_lane ^ (ks << 5);
#pragma unroll
        for (int db = 0; db < DB; ++db) va[db] = vb_lane ^ (db << 6);
        // step 2u-1: S_{2u} from the first half of K_u, PV of step 2u-2 from the first half of V_{u-1}
        fast_step(ICi<0>{}, maskc, 2 * uu, ka, va, sA, sB, pfA, pfB);
        // step 2u: S_{2u+1} from the second half of K_u, PV of step 2u-1 from the second half of V_{u-1}
        fast_step(ICi<1>{}, maskc, 2 * uu + 1, ka, va, sB, sA, pfB, pfA);
      } else {
        constexpr int F = decltype(maskc)::value ? 0 : 1;  // compiler-ordered variant: masked iterations use the generic step
        step(ICi<F>{}, ICi<0>{}, kb_lane, vb_lane, 2 * uu - 1, sA, sB, pfA, pfB);
        step(ICi<F>{}, ICi<1>{}, kb_lane, vb_lane, 2 * uu, sB, sA, pfB, pfA);
      }
      iter_tail();
    };
    int um_lo = uf_lo, um_hi = uf_hi;
    {
      const int f_lo = (w_full_lo - key_base + 31) >> 5;   // first step with no left-masked key
      const int f_hi = (w_full_hi - 31 - key_base) >> 5;   // last step with no right-masked key
      um_lo = max(uf_lo, (f_lo + 1) >> 1);
      um_hi = min(uf_hi, (f_hi - 1) >> 1);
    }
    for (; u <= uf_hi && u < um_lo; ++u) fast_iter(ICi<1>{}, u);
    for (; u <= um_hi; ++u) fast_iter(ICi<0>{}, u);
    for (; u <= uf_hi; ++u) fast_iter(ICi<1>{}, u);
    for (; u <= n_tiles; ++u) {
      iter_head(u);
      const int kb_lane = kbase ^ ((u & 1) * TILE_BYTES);
      const int vb_lane = vbase ^ ((2 + ((u & 1) ^ 1)) * TILE_BYTES);
      step(ICi<0>{}, ICi<0>{}, kb_lane, vb_lane, 2 * u - 1, sA, sB, pfA, pfB);
      step(ICi<0>{}, ICi<1>{}, kb_lane, vb_lane, 2 * u, sB, sA, pfB, pfA);
      iter_tail();
    }
  }

  if (!wave_valid) return;
#pragma unroll
  for (int db = 0; db < DB; ++db)  // a factor decided but not yet applied (1 unless the last scored step moved the maximum)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[db][r] *= o_lag;
  const float l_tot = half_sum(l_run);
  const bool dead = (l_tot == 0.f) || (l_tot != l_tot);
  const float inv = dead ? 1.f : 1.f / l_tot;
  // O tile through LDS (the K/V buffers are free after the last barrier; fa_device.h store_tile_via_lds).  Measured on
  // config 3: the direct epilogue cost ~60 us of the 660 us kernel (tools/overhead_fit.py).
  {
    store_tile_via_lds<E, D>(lds + wave * 32 * (ROW_BYTES + 16), o_acc, inv, op + (int64_t)w_row0 * p.o_rs, p.o_rs, sq - w_row0, lane);
    if (row_valid && hi == 0) lsep[my_row] = dead ? INFINITY : (m_run * cs * kLn2 + __logf(l_tot));
  }
}

template <typename E, int D, int NW, int SCHED>
static int launch_fwd_il_t(const FwdK& p, hipStream_t stream) {
  constexpr int smem = 4 * 64 * D * 2 + (NW == 8 ? NW * 32 * D * 2 : 0);
  auto kern = fa_fwd_il_kernel<E, D, NW, SCHED>;
  static std::atomic<unsigned long long> attr_mask{0};  // the kernel addresses LDS by byte offset: the dynamic segment must start at 0
  if (ensure_dyn_lds(attr_mask, (const void*)kern, smem, true) != 0) return -1;
  const long long total = p.work_list ? (long long)p.work_bound * p.h : units_grid(p.n_units, p.unit_size);
  if (total <= 0) return 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(NW * 64), smem, stream, p);
  if (hipGetLastError() != hipSuccess) return -1;
  LastSchedule& ls = last_schedule();
  ls.fwd_kernel = 2; ls.fwd_nw = NW; ls.fwd_feat = 0; ls.fwd_splits = 1; ls.fwd_list = p.work_list != nullptr; ls.d = D;
  ls.bf16 = std::is_same<E, __bf16>::value;
  snprintf(ls.name, sizeof(ls.name), "fa::fa_fwd_il_kernel<%s,%d,%d,%d>", ls.bf16 ? "bf16" : "f16", D, NW, SCHED);
  return 0;
}

// nw = 4 or 8 waves per workgroup (query block = 32*nw rows)
int launch_fwd_il(const FwdK& p, int dtype_bf16, int d, int nw, hipStream_t stream) {
  // FA_IL_SCHED=0: compiler-ordered steady-state step; default: hand-placed slots, operand reads 3 slots ahead
  const int sched = knobs().il_sched;
  if ((uint64_t)64 * (uint64_t)(p.k_rs > p.v_rs ? p.k_rs : p.v_rs) * 2u >= (1ull << 31)) return -3;
  if (p.softcap > 0.f || p.alibi != nullptr || p.rng != nullptr || p.n_splits > 1) return -2;
#define FA_IL_CASE(E_, D_, NW_)                                                                       \
  if (d == D_ && nw == NW_)                                                                           \
    return sched == 0 ? launch_fwd_il_t<E_, D_, NW_, 0>(p, stream) : launch_fwd_il_t<E_, D_, NW_, 3>(p, stream);
  if (dtype_bf16) {
    FA_IL_CASE(__bf16, 128, 8) FA_IL_CASE(__bf16, 128, 4) FA_IL_CASE(__bf16, 64, 8) FA_IL_CASE(__bf16, 64, 4)
  } else {
    FA_IL_CASE(_Float16, 128, 8) FA_IL_CASE(_Float16, 128, 4) FA_IL_CASE(_Float16, 64, 8) FA_IL_CASE(_Float16, 64, 4)
  }
#undef FA_IL_CASE
  return -2;
}

}  // namespace fa
