// Fused attention backward for gfx950 (MI355X): delta pre-pass, dK/dV kernel, dQ kernel.
//
// Replaces the reference's three-launch backward (compute_dot_do_o, compute_dq_dk_dv_1colblock with
// fp32 atomicAdd into dq_accum, convert_dQ: csrc/flash_attn/src/flash_bwd_preprocess_kernel.h:57-268,
// flash_bwd_kernel.h:80-795) with a deterministic two-pass structure native to the MFMA layouts:
//
//   dK/dV kernel : one workgroup owns a block of keys (each wave 32 keys, K fragments in registers,
//                  the V block resident in LDS) and streams query tiles (Q, dO, LSE, delta) through
//                  LDS; S = Q.K^T and dP = dO.V^T land with column = key = lane, so P and dS are
//                  directly the B operands of dV^T += dO^T.P and dK^T += Q^T.dS (dO^T / Q^T come from
//                  LDS transpose reads).  The query heads of a GQA group are looped INSIDE the
//                  workgroup, so dK/dV are written once -- no dk_expanded + sum_out
//                  (reference flash_api.cpp:935-942,1002-1005) and no atomics.
//   dQ kernel    : forward-shaped: one wave owns 32 query rows (Q, dO fragments in registers, LSE and
//                  delta lane-local), streams K/V tiles; S^T = K.Q^T, dP^T = V.dO^T,
//                  dQ^T += K^T.dS^T (K^T by transpose reads).  No fp32 dq_accum buffer, no convert pass.
//
// 7 MFMA contractions per tile instead of the reference's 5, traded for: zero atomics, bitwise
// determinism for free, no 268 MB accumulator round trip, and every softmax quantity lane-local.
// Arithmetic follows the reference: P = exp2(S*scale*log2e - LSE*log2e) (flash_bwd_kernel.h:536),
// dS = P*(dP - delta) (:584-595), P and dS rounded to the input dtype before their contractions,
// softmax_scale applied once at the end (:733, flash_bwd_preprocess_kernel.h:250).
#include <cstdlib>
#include <type_traits>

#include "fa_device.h"
#include "fa_kernel_params.h"
#include "fa_launch.h"

// dK/dV kernel: LDS operands are read this many MFMA slots minus one ahead (row-major / transposed), one LDS wait per two MFMA ops.  The timing ablations,
// the cycle statistics of the fused backward and the A/B switches measured in rounds 2-4 live in experiments/ablations/fa_bwd.patch (tools/ablate_dkdv.sh,
// tools/ablate_fused.sh, tools/bwd_fused_check.py --stats).
#define FA_BWD_PF 4
#define FA_BWD_PFT 4
// s_waitcnt lgkmcnt(n) with n known only after unrolling (the builtin wants a literal)
#define FA_WAIT_LGKM_CASE(n) case n: __builtin_amdgcn_s_waitcnt(0xC07F | ((n) << 8)); break;
static __device__ __forceinline__ void wait_lgkm_le(int n) {
  switch (n) {
    FA_WAIT_LGKM_CASE(0) FA_WAIT_LGKM_CASE(1) FA_WAIT_LGKM_CASE(2) FA_WAIT_LGKM_CASE(3) FA_WAIT_LGKM_CASE(4) FA_WAIT_LGKM_CASE(5)
    FA_WAIT_LGKM_CASE(6) FA_WAIT_LGKM_CASE(7) FA_WAIT_LGKM_CASE(8) FA_WAIT_LGKM_CASE(9) FA_WAIT_LGKM_CASE(10)
    default: __builtin_amdgcn_s_waitcnt(0xC07F | (0 << 8)); break;
  }
}
// Four schedule experiments of the dK/dV kernel were built, measured and removed again (code: git history up to 3be077c; numbers:
// profiles/r02_bwd_schedules.txt): 4-wave workgroups of 128 keys at D <= 128 (FA_DKDV_SPLIT), the upper four waves one phase ahead of their SIMD
// partners (FA_DKDV_ROT: 1494 | 2394 us against 1431 | 2278 in lock step), the dV / dK segment's first operands read before the vector phase
// (FA_DKDV_CARRY: identical), static priority for the second-dispatched waves (FA_DKDV_PRIO: slower).
#ifndef FA_BWD_PART
#define FA_BWD_PART 0    // build.py compiles this file three times side by side: 1 = delta + dK/dV, 2 = dQ, 3 = the fused backward; 0 = everything
#endif

namespace fa {

constexpr float kLog2e = 1.4426950408889634f;
// 16 bytes of zeros in global memory: where a load / DMA lane fetches from when its chunk lies behind the head dim (BwdK::d_chunks)
static __device__ const uint4 fa_zero_chunk_bwd = {0u, 0u, 0u, 0u};

// ------------------------------------------------------------------------------------------------
// delta[b,h,i] = sum_d dO[i,d] * O[i,d]   (reference flash_bwd_preprocess_kernel.h:24-51, dropout off)
// ------------------------------------------------------------------------------------------------
// (D, DV): tile pitch and head dimension present in memory, as in fa_fwd_kernel (fa_fwd.hip); DV < D = trimmed head dims 32 / 96 / 192
template <typename E, int D, int DV>
__global__ void __launch_bounds__(256) fa_bwd_delta_kernel(const BwdK p) {
  using V8 = typename ElemTraits<E>::v8;
  constexpr int LPR = D / 8;        // lanes per row (a power of two; lanes past DV / 8 add nothing)
  constexpr int ROWS = 256 / LPR;   // rows per block
  const int b = blockIdx.z, h = blockIdx.y;
  int sq = p.sq;
  int64_t row0 = 0, do_boff = (int64_t)b * p.do_bs, o_boff = (int64_t)b * p.o_bs;
  if (p.cu_q) {
    const int c0 = p.cu_q[b];
    sq = p.cu_q[b + 1] - c0;
    row0 = c0;
    do_boff = 0;
    o_boff = 0;
  }
  if (p.seqused_q) sq = min(sq, p.seqused_q[b]);
  const int row = blockIdx.x * ROWS + threadIdx.x / LPR;
  const int c = threadIdx.x % LPR;
  float acc = 0.f;
  if (row < sq && c < (p.d_chunks > 0 ? p.d_chunks : DV / 8)) {
    const E* dop = (const E*)p.dout + do_boff + (row0 + row) * p.do_rs + (int64_t)h * p.do_hs + c * 8;
    const E* op = (const E*)p.o + o_boff + (row0 + row) * p.o_rs + (int64_t)h * p.o_hs + c * 8;
    const V8 a = bitcast_u32x4<V8>(*reinterpret_cast<const u32x4*>(dop));
    const V8 o = bitcast_u32x4<V8>(*reinterpret_cast<const u32x4*>(op));
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += (float)a[j] * (float)o[j];
  }
#pragma unroll
  for (int off = LPR / 2; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (row < sq && c == 0) {
    float* dst = p.cu_q ? (p.delta + (int64_t)h * p.total_q + row0 + row) : (p.delta + ((int64_t)b * p.h + h) * p.sq + row);
    *dst = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// GQA group split (fa_api.cpp bwd_gsplit_plan): dst[b, row, g, :] = sum over s < gs of src[b, row, g * gs + s, :]  (fp32 sum of partials in the input dtype -- the
// reference's own form: per-query-head dK / dV in the input dtype summed by at::sum_out, flash_api.cpp:1000-1004).  src is contiguous [b][sk][h_k * gs][d].
// ------------------------------------------------------------------------------------------------
template <typename E>
__global__ void __launch_bounds__(256) fa_bwd_gsum_kernel(const E* __restrict__ src, E* __restrict__ dst, long long total, int sk, int h_k, int gs, int d8,
                                                         int64_t dst_bs, int64_t dst_rs, int64_t dst_hs) {
  using V8 = typename ElemTraits<E>::v8;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // (b, row, g, 16-byte chunk)
  if (i >= total) return;
  const int c = (int)(i % d8);
  const long long t = i / d8;
  const int g = (int)(t % h_k);
  const long long br = t / h_k;
  const int row = (int)(br % sk);
  const int b = (int)(br / sk);
  const E* sp = src + ((br * h_k + g) * gs) * (int64_t)(d8 * 8) + c * 8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int s_ = 0; s_ < gs; ++s_) {
    const V8 x = bitcast_u32x4<V8>(*reinterpret_cast<const u32x4*>(sp + (int64_t)s_ * d8 * 8));
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += (float)x[j];
  }
  V8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (E)acc[j];
  *reinterpret_cast<u32x4*>(dst + (int64_t)b * dst_bs + (int64_t)row * dst_rs + (int64_t)g * dst_hs + c * 8) = __builtin_bit_cast(u32x4, o);
}

// ------------------------------------------------------------------------------------------------
// dK / dV
// ------------------------------------------------------------------------------------------------
// The kernel's text is a device function so that the fused backward (fa_bwd_fused_kernel below, FUSED = true) can run the dQ contractions of finished
// query blocks behind it in the same workgroup; `bid` = the workgroup's position in the 1-D grid.
template <typename E, int D, int DV, int FEAT, bool FUSED>
static __device__ __forceinline__ void fa_bwd_dkdv_body(const BwdK& p, const int bid) {
  constexpr bool XFORM = (FEAT & (FEAT_CAP | FEAT_ALIBI)) != 0;  // scores pass through the scaled domain
  constexpr bool F_CAP = (FEAT & FEAT_CAP) != 0, F_ALIBI = (FEAT & FEAT_ALIBI) != 0, F_DROP = (FEAT & FEAT_DROP) != 0;
  using T = ElemTraits<E>;
  using V8 = typename T::v8;
  using V4 = typename T::v4;
  // D <= 128: 8 waves (two per SIMD, 256 registers each); D = 256: 4 waves, one per SIMD, with the 512-register budget
  // the two 32 x 256 accumulators need, and 32-query tiles so that V block + Q/dO double buffers fit 160 KB of LDS
  constexpr int NW = D > 128 ? 4 : 8, NT = NW * 64;
  constexpr int BNK = NW * 32;   // keys per workgroup
  constexpr int BMQ = D > 128 ? 32 : 64;  // queries per streamed tile (32-row sub-blocks)
  constexpr int CPR = D / 8, ROW_BYTES = D * 2;
  constexpr int KS = DV / 16, DB = DV / 32, CV = DV / 8;   // k-steps / output blocks / 16-B chunks that exist
  static_assert(DV % 32 == 0 && DV <= D && 2 * DV >= D, "DV: a multiple of 32 in [D/2, D]");
  constexpr int VBLK_BYTES = BNK * ROW_BYTES;
  constexpr int QT_BYTES = BMQ * ROW_BYTES;
  // PRE (plain variant): the matrix pipe does the two per-element subtractions.  This wave's K fragments are multiplied by
  // softmax_scale*log2(e) once, when they are loaded (rounded once to the input dtype, like Q in fa_fwd_w64.hip), and the score chain's
  // first MFMA takes C = -LSE*log2(e) of its rows instead of 0, so the scores leave the pipe as the exponent of P; the dP chain
  // starts from C = -delta.  Per element that leaves exp2, one multiply and the packing (was: fma, exp2, sub, mul, packing) -- in a kernel
  // whose vector phase does not overlap its matrix phases (profiles/r02_bwd_schedules.txt).  FA_STRICT=1 runs the run-time-checked
  // variant instead, which scales every score in fp32.
  // Round 4: that variant (FEAT_NONE, "fast") is opt-in (FA_DKDV_PRESCALE=1).  The default plain variant is FEAT_EXACT: every score is scaled in fp32
  // (one fma per element: S*c - LSE*log2e), only the dP chain keeps its C = -delta (exact: nothing is rounded there).  K * scale * log2e rounded to the
  // input dtype puts an error of |score| * 2^-9 (bf16) / 2^-12 (fp16) log2 units into the exponent of P that the forward's LSE does not share; the
  // reference's own acceptance tests see it (tests/test_flash_attn.py::test_flash_attn_causal with one visible key: P must be exactly 1 and dV exactly
  // dO, we returned dO * (1 + 2^-8); test_flash_attn_bwd_overflow in fp16: dV error 7x PyTorch's; profiles/r04_reference_suite.txt).
  constexpr bool PRE = (FEAT == FEAT_NONE);                    // K pre-scaled, score chain starts from C = -LSE*log2e
  constexpr bool PRE_D = ((FEAT & FEAT_ALL) == FEAT_NONE);     // dP chain starts from C = -delta (FEAT_NONE and FEAT_EXACT)
  // LDS: Q0 | Q1 | dO0 | dO1 | V block | aux0 aux1   (aux = BMQ x LSE*log2e followed by BMQ x delta).  The streamed tiles
  // sit below 64 KB so that (buffer, sub-block) offsets fit the 16-bit immediate of ds_read.
  constexpr int OFF_Q = 0, OFF_DO = 2 * QT_BYTES, OFF_V = 4 * QT_BYTES, OFF_AUX = OFF_V + VBLK_BYTES;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char FA_LDS* lds = (char FA_LDS*)smem;

  int tid_ = threadIdx.x;
  if constexpr (FUSED) asm volatile("" : "+v"(tid_));   // (inside the persistent loop of fa_bwd_fused_kernel: keeps everything derived from the thread index -- both
                                                        // parts' address arithmetic -- from being hoisted out of that loop and held in registers across the other part)
  const int tid = tid_, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, ki = lane & 31;

  // 1-D grid, XCD-aware mapping (xcd_interleave): (batch, kv head) units are dealt round-robin to the XCDs and
  // each XCD walks all key blocks of its units, so it sees every key-block index equally often.  (With a plain (n, h, b) grid XCD x
  // would only ever get key blocks n = x mod 8 -- under a causal mask a 2.4x work imbalance between XCDs.)
  // Within a head the low key blocks come first: they see the most queries under a causal mask.
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
  if (p.seqused_k) sk = min(p.seqused_k[b], sk);   // (include/fa_gfx950.h: in the backward seqused_k can only SHORTEN a sequence -- the key-block work list and its bound are sized from cu_seqlens_k)
  const int n0 = n_block * BNK;
  if (n0 >= sk) return;
  const int n1 = min(n0 + BNK, sk);
  const int shift = sk - sq;

  const E* __restrict__ kp = (const E*)p.k + k_boff + k_row0 * p.k_rs + (int64_t)(hk >> p.kv_in_shift) * p.k_hs;   // (kv_in_shift: a GQA group split into virtual kv heads, fa_kernel_params.h)
  const E* __restrict__ vp = (const E*)p.v + v_boff + k_row0 * p.v_rs + (int64_t)(hk >> p.kv_in_shift) * p.v_hs;

  // ---- query range that can see this key block -------------------------------------------------
  int q_lo = 0, q_hi = sq - 1;
  if (p.wr >= 0) q_lo = max(0, n0 - shift - p.wr);
  if (p.wl >= 0) q_hi = min(sq - 1, n1 - 1 - shift + p.wl);
  const int m_lo = q_lo / BMQ;
  const int nm = (q_hi >= q_lo) ? (q_hi / BMQ + 1 - m_lo) : 0;  // tiles per query head
  const int n_items = nm * p.hk_ratio;

  // ---- this wave's keys ----------------------------------------------------------------------------
  const int wk0 = n0 + wave * 32;
  const int wk1 = min(wk0 + 31, sk - 1);
  const bool wave_valid = wk0 < sk;
  const int my_key = wk0 + ki;
  const bool key_valid = my_key < sk;

  // head dims between the built sizes (BwdK::d_chunks): the chunks behind the head dim read as zeros (K / V by predicate, the streamed
  // Q / dO tiles from a page of zeros) and are not stored
  const int cvr = p.d_chunks > 0 ? p.d_chunks : CV;
  const bool bounded = cvr < CV;
  const E* zsrc = (const E*)&fa_zero_chunk_bwd;
  // K fragments (B operand of S = Q.K^T): lane = key, 8 consecutive d per k-step
  V8 kf[KS];
  {
    const E* krow = kp + (int64_t)my_key * p.k_rs + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kf[ks] = bitcast_u32x4<V8>(ld_global_16B(krow + 16 * ks, key_valid && 2 * ks + hi < cvr));
    if constexpr (PRE) {  // K <- K * softmax_scale * log2(e), rounded once to the input dtype (see PRE above)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) kf[ks][j] = (E)((float)kf[ks][j] * p.scale_log2);
    }
  }
  // V block -> LDS (B operand of dP = dO.V^T is re-read per step to keep registers for the accumulators)
  {
    constexpr int LDV = (BNK * CPR) / NT;
#pragma unroll
    for (int i = 0; i < LDV; ++i) {
      const int idx = tid + i * NT;
      const int row = idx / CPR, ch = idx % CPR;
      const u32x4 x = ld_global_16B(vp + (int64_t)(n0 + row) * p.v_rs + ch * 8, n0 + row < sk && ch < cvr);
      *(u32x4 FA_LDS*)(lds + OFF_V + tile_off<D>(row, ch)) = x;
    }
  }

  // ---- streamed tile staging: Q / dO tiles go global -> LDS by DMA (global_load_lds, 1 KiB per wave
  // instruction, no staging registers); the XOR swizzle is applied to the per-lane SOURCE address because
  // the DMA destination is lane-linear.  Rows past the sequence end are clamped to the last row: their
  // LSE is +inf, so P = dS = 0 and they contribute nothing.
  constexpr int RPD = 1024 / ROW_BYTES;            // tile rows per DMA instruction
  constexpr int NDMA = (BMQ * ROW_BYTES) / 1024;   // DMA instructions per tile
  constexpr int DPW = NDMA / NW;                   // per wave
  static_assert(NDMA % NW == 0 && DPW >= 1, "tile does not divide over the waves");
  float aux_reg = 0.f;  // threads [0,BMQ): LSE*log2e of row tid; [BMQ,2BMQ): delta of row tid-BMQ
  // item it = (query head it / nm of the group, query tile it % nm), kept as counters for the current and the next item (the
  // only two anybody asks about): a division per use is ~35 instructions of mixed scalar / vector code, several times per item
  int it_cur = 0, c_im = 0, c_ih = 0, n_im = (nm > 1) ? 1 : 0, n_ih = (nm > 1) ? 0 : 1;
  auto item_head = [&](int it) { return hk * p.hk_ratio + (it == it_cur ? c_ih : n_ih); };
  // Walk order of the query tiles.  Under a causal (right-bounded) mask key block n sees the tiles m_lo(n) .. last: walking them
  // DOWN from the last one, all key blocks of a (batch, head) -- co-resident on one XCD -- read the same Q/dO tile at the same
  // time and it is fetched into that L2 once; walking up, every key block starts at its own m_lo and the L2 would have to
  // hold the head's whole Q and dO (the round-1 counters: 51 % L2 hits, 2.8x the algorithmic HBM traffic at config 3).
  const bool walk_down = p.wr >= 0 && p.wl < 0;
  auto item_m0 = [&](int it) {
    const int im = (it == it_cur ? c_im : n_im);
    return (m_lo + (walk_down ? nm - 1 - im : im)) * BMQ;
  };
  auto item_done = [&]() { ++it_cur; c_im = n_im; c_ih = n_ih; if (++n_im == nm) { n_im = 0; ++n_ih; } };
  auto load_item = [&](int it, int buf) {
    const int h = item_head(it);
    const int m0 = item_m0(it);
    const E* qp = (const E*)p.q + q_boff + q_row0 * p.q_rs + (int64_t)h * p.q_hs;
    const E* dop = (const E*)p.dout + do_boff + q_row0 * p.do_rs + (int64_t)h * p.do_hs;
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      const int idx = wave * DPW + i;
      const int row = idx * RPD + lane / CPR;
      const int pc = lane % CPR;
      const int grow = min(m0 + row, sq - 1);
      int c = pc ^ swz16<D>(row);
      if (DV < D) c = c < CV ? c : 0;  // columns past the head dimension are never read from LDS: fetch something that exists
      const E* qsrc = qp + (int64_t)grow * p.q_rs + c * 8;
      const E* dsrc = dop + (int64_t)grow * p.do_rs + c * 8;
      if (bounded && c >= cvr) { qsrc = zsrc; dsrc = zsrc; }
      lds_dma_16B(qsrc, lds + OFF_Q + buf * QT_BYTES + idx * 1024);
      lds_dma_16B(dsrc, lds + OFF_DO + buf * QT_BYTES + idx * 1024);
    }
    if (tid < 2 * BMQ) {
      const int r = tid & (BMQ - 1);
      const bool ok = (m0 + r) < sq;
      const int64_t base = p.cu_q ? ((int64_t)h * p.total_q + q_row0) : (((int64_t)b * p.h + h) * p.sq);
      const float* src = (tid < BMQ ? p.lse : p.delta) + base + m0 + r;
      const float x = ok ? *src : 0.f;
      aux_reg = (tid < BMQ) ? (ok ? x * kLog2e : INFINITY) : x;  // rows past the end: LSE = +inf => P = 0
      if (PRE || (PRE_D && tid >= BMQ)) aux_reg = -aux_reg;            // (the C operands of the score / dP chains)
    }
  };
  auto store_item = [&](int buf) {
    if (tid < 2 * BMQ) *(float FA_LDS*)(lds + OFF_AUX + (buf * 2 * BMQ + tid) * 4) = aux_reg;
  };

  // per-lane LDS read offsets
  const int row_off = tile_off<D>(ki, 0) - (swz16<D>(ki) << 4);  // = ki * ROW_BYTES
  const int rswz = swz16<D>(ki);
  const int tr_i = lane & 15, tr_half = (lane >> 4) & 1, tr_rr = tr_i >> 2, tr_cc = tr_i & 3;
  // offset(db, s) = tr_base[s] ^ (db << 6): the d-block index only enters through XOR on the 64-B chunk bits
  int tr_base[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int row = 8 * s + 4 * hi + tr_rr;
    tr_base[s] = tile_off<D>(row, 2 * tr_half + (tr_cc >> 1)) + (tr_cc & 1) * 8;
  }

  f32x16 dk_acc[DB], dv_acc[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk_acc[db][r] = 0.f; dv_acc[db][r] = 0.f; }

  const float cs = XFORM ? kLog2e : p.scale_log2;

  if (n_items > 0) {
    load_item(0, 0);
    store_item(0);
  }
  lds_dma_wait_all();
  __syncthreads();

  // Per-lane operand offsets inside a 32-row sub-tile.  Row-major fragment of k-step ks: k0 ^ (ks << 5) (the row offset has
  // zero low bits, so the swizzled chunk index (2ks+hi)^swz folds into one XOR); the V block adds a multiple of 8 KB.
  // The 8 + 8 + 8 derived addresses are recomputed where they are used (one v_xor each) behind an opaque copy of the
  // base: hoisted out of the loop they cost 24 registers the accumulators need, and a spill reload waits on vmcnt(0),
  // i.e. on the Q/dO DMA prefetch in flight.
  const int k0 = row_off + ((hi ^ rswz) << 4);
  const int kv0 = k0 + OFF_V + wave * 32 * ROW_BYTES;
  const int aux_lane = OFF_AUX + 4 * hi * 4;
  auto opaque = [](int x) __attribute__((always_inline)) { asm volatile("" : "+v"(x)); return x; };

  // A sub-block (32 queries x this wave's 32 keys) goes through three phases: P1 = the S / dP contractions (16 MFMAs), SM = the
  // softmax arithmetic (VALU: P, dS, packing), P2 = the dV / dK contractions (16 MFMAs).  The two waves that share a SIMD
  // (waves w and w + 4) start every item together and sit in the same phase all the time; timing ablations give
  // time = MFMA time + everything else (profiles/r02_bwd_schedules.txt).  Running the upper four waves one phase ahead was tried and
  // measured slower (1494 | 2394 us against 1431 | 2278 us, causal | full, config 3): the vector phase of one wave does not hide under
  // the matrix phase of its partner.
  f32x16 s, dp;              // S / dP of a sub-block between P1 and SM
  V8 pfrag[2], dsfrag[2];    // P / dS between SM and P2
  constexpr int NQB = BMQ / 32;
  auto sub_active = [&](int it, int qb) __attribute__((always_inline)) {
    return it < n_items && ds_tile_active(item_m0(it) + 32 * qb, wk0, sq, sk, shift, p.wl, p.wr);
  };

  // P1: S[query][key] = Q.K^T ; dP[query][key] = dO.V^T   (column = key = lane).  The two accumulation chains alternate
  // (op j = k-step j/2 of S for even j, of dP for odd j) and the LDS operands are read PF-1 ops ahead.
  auto p1 = [&](auto bufc, auto qbc) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value, qb = decltype(qbc)::value;
    constexpr int QB_OFF = OFF_Q + buf * QT_BYTES, DOB_OFF = OFF_DO + buf * QT_BYTES, sub = qb * 32 * ROW_BYTES;
    constexpr int NOPS = 2 * KS, PF = FA_BWD_PF;
    u32x4 ra[PF], rb[2];
    const int k0p = opaque(k0), kv0p = opaque(kv0);
    f32x16 c_s, c_dp;   // PRE: -LSE*log2e / -delta of the accumulator rows (queries acc_row(r, hi))
    if constexpr (PRE_D) {
      const int auxp = opaque(aux_lane);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if constexpr (PRE) {
          const f32x4 l4 = *(const f32x4 FA_LDS*)(unsigned long)(unsigned)(auxp + (buf * 2 * BMQ + qb * 32 + 8 * g) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) c_s[4 * g + j] = l4[j];
        }
        const f32x4 d4 = *(const f32x4 FA_LDS*)(unsigned long)(unsigned)(auxp + (buf * 2 * BMQ + qb * 32 + 8 * g) * 4 + BMQ * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) c_dp[4 * g + j] = d4[j];
      }
    }
    auto rd = [&](int j) __attribute__((always_inline)) {
      const int ks = j >> 1;
      if ((j & 1) == 0) {
        ra[j % PF] = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)((QB_OFF + sub) + (k0p ^ (ks << 5)));   // (byte offsets, not lds + ..: the segment base
                                                                                                              // is 0 by construction, and as a pointer add it costs a v_add per read)
      } else {
        ra[j % PF] = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)((DOB_OFF + sub) + (k0p ^ (ks << 5)));
        rb[ks & 1] = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)(kv0p ^ (ks << 5));
      }
    };
#pragma unroll
    for (int j = 0; j < PF - 1; ++j) rd(j);
#pragma unroll
    for (int j = 0; j < NOPS; ++j) {
      if (j + PF - 1 < NOPS) rd(j + PF - 1);
      // one LDS wait per TWO ops (as the 64-rows-per-wave kernels): before an even op, wait until the operands of the odd op behind it have
      // landed too -- everything requested later (ops j + 2 .. j + PF - 1: one read for an S op, two for a dP op) may stay in flight.
      // hipcc models the explicit wait and emits none in front of the odd op.
      if ((j & 1) == 0 && j + 1 < NOPS) {
        int out = 0;
        for (int g = j + 2; g <= j + PF - 1 && g < NOPS; ++g) out += (g & 1) ? 2 : 1;
        wait_lgkm_le(out);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch above this op's MFMA
      const int ks = j >> 1;
      f32x16 c = (j & 1) ? dp : s;
      if (j < 2) {
        if ((j & 1) ? PRE_D : PRE) {
          c = (j & 1) ? c_dp : c_s;
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) c[r] = 0.f;
        }
      }
      if ((j & 1) == 0) s = T::mfma(bitcast_u32x4<V8>(ra[j % PF]), kf[ks], c);
      else dp = T::mfma(bitcast_u32x4<V8>(ra[j % PF]), bitcast_u32x4<V8>(rb[ks & 1]), c);
    }
  };

  // SM: score transform, mask, P = exp2(S*c - LSE*log2e), dS = P * (dP - delta); rows are queries acc_row(r,hi)
  auto sm = [&](auto bufc, auto qbc, int it) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value, qb = decltype(qbc)::value;
    const int q0 = item_m0(it) + 32 * qb;
    const bool use_alibi = F_ALIBI && (FEAT != FEAT_ALL || p.alibi != nullptr);
    const bool use_cap = F_CAP && (FEAT != FEAT_ALL || p.softcap > 0.f);
    const float slope = use_alibi ? p.alibi[(int64_t)b * p.alibi_bs + item_head(it)] : 0.f;
    const bool drop = F_DROP && (FEAT != FEAT_ALL || p.rng != nullptr);
    // stream key of this (batch, query head) plus this lane's key group; rows are added per 4-query group below
    const uint32_t drop_key = drop ? drop_bh_key(p.rng, b * p.h + item_head(it)) : 0u;

    f32x16 dcap;  // d(softcap*tanh(x/softcap))/dx = 1 - tanh^2 (reference flash_bwd_kernel.h:588 / utils.h:395-409)
    if constexpr (XFORM) {
      const float rcap = use_cap ? 1.f / p.softcap : 0.f;
      // ALiBi: distance of this lane's key from the query of element r = rel + acc_row(r, 0), formed in fp32 (exact below 2^24) from ONE value
      // made here, behind an opaque copy: written as (float)(q0 + acc_row(r, hi) + shift - my_key) hipcc hoists the sixteen loop-invariant parts
      // (acc_row(r, hi) + shift - my_key) out of the item loop, finds no registers for them next to the accumulators and reloads them from
      // scratch in every sub-block -- behind a vmcnt(0) that waits for the Q / dO prefetch in flight (46 spills, 188 bytes of scratch).
      const float rel = F_ALIBI ? (float)opaque(q0 + 4 * hi + shift - my_key) : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float y = s[r] * p.scale;
        if constexpr (F_CAP) {
          dcap[r] = 1.f;
          if (use_cap) {
            const float t = fast_tanh(y * rcap);
            y = p.softcap * t;
            dcap[r] = 1.f - t * t;
          }
        }
        if constexpr (F_ALIBI) {
          if (use_alibi) y -= slope * fabsf(rel + (float)acc_row(r, 0));
        }
        s[r] = y;
      }
    }
    bool need_mask = false;
    if (p.wr >= 0) need_mask = need_mask || (wk1 > q0 + shift + p.wr);
    if (p.wl >= 0) need_mask = need_mask || (wk0 < q0 + 31 + shift - p.wl);
    if (need_mask) {
      const int rel_lo = (p.wr >= 0) ? (my_key - shift - p.wr - q0 - 4 * hi) : -(1 << 30);
      const int rel_hi = (p.wl >= 0) ? (my_key - shift + p.wl - q0 - 4 * hi) : (1 << 30);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int off = acc_row(r, 0);
        s[r] = ((off >= rel_lo) && (off <= rel_hi)) ? s[r] : -INFINITY;
      }
    }

    const int auxp = opaque(aux_lane);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 l4 = {0.f, 0.f, 0.f, 0.f}, d4 = {0.f, 0.f, 0.f, 0.f};
      if constexpr (!PRE) l4 = *(const f32x4 FA_LDS*)(unsigned long)(unsigned)(auxp + (buf * 2 * BMQ + qb * 32 + 8 * g) * 4);
      if constexpr (!PRE_D) d4 = *(const f32x4 FA_LDS*)(unsigned long)(unsigned)(auxp + (buf * 2 * BMQ + qb * 32 + 8 * g) * 4 + BMQ * 4);
      // Dropout: the 4 lanes of a quad hold the 4 keys of one key group; lane a hashes query row a of this 4-row
      // group (4 bytes = those 4 keys) and the quad exchanges words, so each lane reads its key's byte of every row.
      uint32_t hq = 0u;
      if constexpr (F_DROP) {
        if (drop) hq = drop_bytes(drop_key, q0 + 8 * g + 4 * hi + (ki & 3), my_key >> 2);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * g + j;
        const float ex = PRE ? s[r] : __builtin_fmaf(s[r], cs, -l4[j]);   // PRE: the pipe already delivered S*c - LSE*log2e, and dP - delta
        const float pv = fast_exp2(ex);
        float pkeep = pv, dpe = dp[r];
        if constexpr (F_DROP) {
          if (drop) {  // Z = keep / (1 - p): dV uses P*Z (the 1/(1-p) is applied to dV at the end), dS = P*(dP*Z - delta)
            const uint32_t hj = quad_bcast(hq, j);
            const bool keep = ((hj >> (8 * (ki & 3))) & 0xffu) <= p.drop_thr8;
            pkeep = keep ? pv : 0.f;
            dpe = keep ? dp[r] * p.rp_keep : 0.f;
          }
        }
        float dsv = PRE_D ? pv * dpe : pv * (dpe - d4[j]);
        if constexpr (F_CAP) dsv *= dcap[r];
        pfrag[r >> 3][r & 7] = (E)pkeep;
        dsfrag[r >> 3][r & 7] = (E)dsv;
      }
    }

    // (the fused backward's hand-over of dS, FA_BWD_MODE=3; ds_ws is NULL otherwise and the branch is
    // not taken.  It is part of every variant: without it hipcc's register allocation of the softcap variant at D = 128 comes out two vector spills worse --
    // 12 bytes of scratch, tests/test_kernel_resources_cpu.py)
    if (p.ds_ws) {  // dS spill: this sub-tile's fragments, one 16-byte slot per lane (fa_device.h ds_slot), for the dQ contraction
      if (!key_valid) {  // keys past the end (their K rows are clamped copies over there): no contribution
#pragma unroll
        for (int j = 0; j < 8; ++j) { dsfrag[0][j] = (E)0.f; dsfrag[1][j] = (E)0.f; }
      }
      // (round 6: rows packed as in the chunked path -- fa_device.h ds_row_start: under a causal mask row block i holds only the key pairs it can see, half the workspace)
      E* dst = (E*)p.ds_ws + ((((int64_t)b * p.h + item_head(it)) * p.ds_head_tiles + ds_row_start(q0 >> 5, p.ds_c1, p.ds_jb, p.ds_np64) + (wk0 >> 5)) << 10) + ds_slot(ki, hi) * 8;
      // (every wave of the workgroup computes this query sub-block as soon as ONE of the block's 256 keys is visible to it: a wave whose own 32 keys lie behind the
      // packed row's last pair holds zeros nobody reads -- and no slot: it must not store, the next row starts there)
      const bool in_row = (wk0 >> 5) < 2 * min(((q0 >> 5) + p.ds_c1) >> 1, p.ds_np64);
      if constexpr (FUSED) {   // read by another workgroup of the SAME launch: written through (sc1), see fa_bwd_fused_kernel
        if (in_row) {
          st_global_16B_sc1(dst, __builtin_bit_cast(u32x4, dsfrag[0]));
          st_global_16B_sc1(dst + 512, __builtin_bit_cast(u32x4, dsfrag[1]));
        }
      } else if (in_row) {
        *(u32x4*)dst = __builtin_bit_cast(u32x4, dsfrag[0]);
        *(u32x4*)(dst + 512) = __builtin_bit_cast(u32x4, dsfrag[1]);
      }
    }
  };

  // P2: dV^T[d][key] += dO^T[d][query] . P[query][key] ;  dK^T[d][key] += Q^T[d][query] . dS[query][key]
  // op i: source = dO (even) / Q (odd), d-block (i>>1) % DB, query half t = i / (2*DB); transpose reads PFT-1 ops ahead
  // `mid` = the vector phase of the same sub-block: it runs first, then the segment's first operand reads are issued.
  auto p2 = [&](auto bufc, auto qbc, auto&& mid) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value, qb = decltype(qbc)::value;
    constexpr int QB_OFF = OFF_Q + buf * QT_BYTES, DOB_OFF = OFF_DO + buf * QT_BYTES, sub = qb * 32 * ROW_BYTES;
    constexpr int NOPS = 4 * DB, PFT = FA_BWD_PFT;
    s16x4 tlo[PFT], thi[PFT];
    const int t0p = opaque(tr_base[0]), t1p = opaque(tr_base[1]);
    auto rd = [&](int i) __attribute__((always_inline)) {
      const int db = (i >> 1) % DB, t = i / (2 * DB);
      const int base = ((i & 1) ? QB_OFF : DOB_OFF) + sub + 16 * t * ROW_BYTES;
      tlo[i % PFT] = lds_read_tr16((const char FA_LDS*)(unsigned long)(unsigned)(base + (t0p ^ (db << 6))));
      thi[i % PFT] = lds_read_tr16((const char FA_LDS*)(unsigned long)(unsigned)(base + (t1p ^ (db << 6))));
    };
    mid();
#pragma unroll
    for (int i = 0; i < PFT - 1; ++i) rd(i);
#pragma unroll
    for (int i = 0; i < NOPS; ++i) {
      if (i + PFT - 1 < NOPS) rd(i + PFT - 1);
      if ((i & 1) == 0 && i + 1 < NOPS) {   // (two transpose reads per op)
        int out = 0;
        for (int g = i + 2; g <= i + PFT - 1 && g < NOPS; ++g) out += 2;
        wait_lgkm_le(out);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int db = (i >> 1) % DB, t = i / (2 * DB);
      if ((i & 1) == 0) dv_acc[db] = T::mfma(combine_tr<V8>(tlo[i % PFT], thi[i % PFT]), pfrag[t], dv_acc[db]);
      else dk_acc[db] = T::mfma(combine_tr<V8>(tlo[i % PFT], thi[i % PFT]), dsfrag[t], dk_acc[db]);
    }
  };

  // FUSED: per (batch, head, 256-query block) arrival counters.  The workgroup whose ticket is the last one expected for a block puts the block on
  // the ready queue of dQ items (fa_bwd_fused_kernel).  State of thread 0 only.
  constexpr int FZ_TPB = 256 / BMQ;   // streamed tiles per dQ block
  int fz_item = -1, fz_old = 0, fz_exp = 0;
  auto fz_settle = [&]() __attribute__((always_inline)) {
    if (fz_item >= 0) {
      if (fz_old == fz_exp - 1) {
        // publish on this XCD's queue: slot <- item + 1 (an exchange: its return proves the word is in memory), THEN one more available item
        int32_t* ctrl = p.fuse_sync + FZ_CTRL + (bid & 7) * FZ_CTRL_STRIDE;
        const int slot = __hip_atomic_fetch_add(ctrl + FZ_TAIL, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int was = __hip_atomic_exchange(p.fuse_sync + FZ_COUNTERS + (int64_t)p.fuse_items * p.fuse_line + (int64_t)(bid & 7) * p.fuse_items + slot, fz_item + 1, __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" : "+v"(was));   // (the increment below must not be issued before the exchange has returned)
        __hip_atomic_fetch_add(ctrl + FZ_AVAIL, was == 0x7fffffff ? 2 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      fz_item = -1;
    }
  };
  using Q0 = std::integral_constant<int, 0>;
  using Q1 = std::integral_constant<int, 1>;
  // Every wave in lock step, one barrier per item.
  auto item = [&](auto curc, int it) __attribute__((always_inline)) {
    constexpr int cur = decltype(curc)::value;
    using CUR = std::integral_constant<int, cur>;
    const bool has_next = it + 1 < n_items;
    if (has_next) load_item(it + 1, cur ^ 1);  // DMA lands in the other buffer while this item is computed
    const bool a0 = sub_active(it, 0), a1 = NQB > 1 && sub_active(it, 1);
    if (a0) { p1(CUR{}, Q0{}); p2(CUR{}, Q0{}, [&]() __attribute__((always_inline)) { sm(CUR{}, Q0{}, it); }); }
    if constexpr (NQB > 1) {
      if (a1) { p1(CUR{}, Q1{}); p2(CUR{}, Q1{}, [&]() __attribute__((always_inline)) { sm(CUR{}, Q1{}, it); }); }
    }
    if (has_next) store_item(cur ^ 1);
    lds_dma_wait_all();
    __syncthreads();
    if constexpr (FUSED) {
      // Every wave's dS stores of this item are acknowledged (the vmcnt(0) above, in every wave, then the barrier).  Thread 0 settles the ticket it
      // drew one item ago (its return value has long arrived: no wait) and draws one for the query block this item completed, if any.
      if (tid == 0) {
        fz_settle();
        const int mt = item_m0(it) / BMQ;
        const bool block_done = walk_down ? ((mt % FZ_TPB) == 0 || mt == m_lo) : ((mt % FZ_TPB) == FZ_TPB - 1 || mt == m_lo + nm - 1);
        if (block_done) {
          const int mb = mt / FZ_TPB;
          const int blr = min(256 * mb + 255, sq - 1);                    // last row of the block
          const int nnbv = (sk + BNK - 1) / BNK;                          // key blocks that exist
          fz_exp = (p.wr >= 0) ? min(nnbv, (blr + shift + p.wr) / BNK + 1) : nnbv;   // key blocks that visit it (launch_bwd_fused: no left window, sk >= sq)
          fz_item = ((int)b * p.h + item_head(it)) * p.nmb + mb;
          fz_old = __hip_atomic_fetch_add(p.fuse_sync + FZ_COUNTERS + (int64_t)fz_item * p.fuse_line, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  };
  for (int it = 0; it < n_items; it += 2) {
    item(Q0{}, it);
    item_done();
    if (it + 1 < n_items) { item(Q1{}, it + 1); item_done(); }
  }

  if constexpr (FUSED) {
    if (tid == 0) fz_settle();
  }
  // ---- epilogue: dK = scale * acc, dV = acc; every key row of the block is written (zeros included) --
  if (!wave_valid) return;
  // dK / dV tiles through the freed Q/dO buffers: whole-row stores (fa_device.h store_tile_via_lds); every key row of the
  // block is written (zeros included: empty-sequence contract of the CK tests)
  const float dv_scale = (F_DROP && p.rng) ? p.rp_keep : 1.f;
  E* dktile = (E*)p.dk + dk_boff + (k_row0 + wk0) * p.dk_rs + (int64_t)hk * p.dk_hs;
  E* dvtile = (E*)p.dv + dv_boff + (k_row0 + wk0) * p.dv_rs + (int64_t)hk * p.dv_hs;
  char FA_LDS* stage = lds + wave * 32 * (ROW_BYTES + 16);
  store_tile_via_lds<E, D, DV>(stage, dk_acc, p.scale, dktile, p.dk_rs, sk - wk0, lane, cvr);
  store_tile_via_lds<E, D, DV>(stage, dv_acc, dv_scale, dvtile, p.dv_rs, sk - wk0, lane, cvr);
}

template <typename E, int D, int DV, int FEAT>
__global__ void __launch_bounds__(D > 128 ? 256 : 512, D > 128 ? 1 : 2) fa_bwd_dkdv_kernel(const BwdK p) {
  fa_bwd_dkdv_body<E, D, DV, FEAT, false>(p, blockIdx.x);
}

#if FA_BWD_PART == 0 || FA_BWD_PART == 3
// ------------------------------------------------------------------------------------------------
// Fused backward (FA_BWD_MODE=3): 5 contractions per tile, deterministic, one launch
// ------------------------------------------------------------------------------------------------
// Opt-in (measured: +4 % at the headline backward shape, +14 ... +27 % at causal S = 2k / 1k, -4 % non-causal at S = 4k, worse at head dim 64, and it needs
// B*H*Sq*Sk*2 bytes of workspace -- profiles/r04_bwd_fused.txt; the default stays the scratch-free 7-contraction pair).
// The dK/dV part writes every dS sub-tile it forms (rounded to the input dtype, as it enters the dK contraction) to a workspace and counts, per
// (batch, head, 256-query block), the key blocks that have passed that query block.  The key block that brings a counter to its expected value puts the
// query block on its XCD's ready queue; a workgroup, once its own key block is finished (registers and LDS free), takes query blocks off the queue until
// it is empty and computes dQ = dS . K for them -- ONE contraction, where the recomputing dQ kernels spend three.  Nobody ever waits for a consumer (the
// last arrival is the one that publishes, and whoever publishes drains the queue itself afterwards), so correctness needs no assumption about dispatch
// order or residency, and the order of every accumulation is fixed: results are bitwise reproducible, dK / dV bitwise equal to the default path's.  Under
// a causal mask all key blocks of a head walk the query tiles downwards together, so a query block's dS is consumed shortly after it was written.
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): dS stores are written through (sc1) and acknowledged (vmcnt(0) in every wave + barrier) before
// thread 0 draws the ticket (relaxed agent-scope atomic); the consumer takes the item, runs ONE agent-scope acquire, then a barrier, then plain loads.
//
// dQ^T[d][query] = sum_key K^T[d][key] . dS^T[key][query] for one 256-row block: 8 waves x 32 rows; K tiles (64 keys) shared through LDS, each wave's dS
// sub-tiles DMA'd into its private LDS rows and read back transposed (ds_read_b64_tr_b16 turns the writer's lane = key image into the lane = query B
// operand; fa_device.h ds_slot).
template <typename E, int D>
static __device__ __forceinline__ void fa_bwd_dq_from_ds(const BwdK& p, char FA_LDS* lds, const int b, const int h, const int m_block) {
  using T = ElemTraits<E>;
  using V8 = typename T::v8;
  constexpr int NW = 8, BM = NW * 32, BN = 64, CPR = D / 8;
  constexpr int ROW_BYTES = D * 2, TILE_BYTES = BN * ROW_BYTES, DB = D / 32;
  constexpr int DS_WAVE = 2 * 2048;        // per wave and tile: the two 32-key sub-tiles of its 32 rows (contiguous in the workspace)
  constexpr int DS_BUF = NW * DS_WAVE;
  // LDS: K0 | K1 | K2 | dS0 | dS1 | dS2 -- a ring of three tiles, two of them in flight while one is computed on: the loop moves 48 KB per tile and
  // 16 MFMAs per wave, so it runs at what the memory system returns per CU, i.e. bytes in flight / latency (one tile in flight: 1.4 us per tile measured)
  constexpr int NST = 3;
  constexpr int OFF_DS = NST * TILE_BYTES;
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));   // (not hoisted out of the persistent loop, see fa_bwd_dkdv_body)
  const int tid = tid_, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int hk = h / p.hk_ratio;
  const int sq = p.sq, sk = p.sk;
  const int m0 = m_block * BM;
  const E* __restrict__ kp = (const E*)p.k + (int64_t)b * p.k_bs + (int64_t)hk * p.k_hs;
  const int shift = sk - sq;
  const int blk_last = min(m0 + BM, sq) - 1;
  int kmax = sk - 1, kmin = 0;
  if (p.wr >= 0) kmax = min(kmax, blk_last + shift + p.wr);
  if (p.wl >= 0) kmin = max(0, m0 + shift - p.wl);
  const int n_min = kmin / BN;
  const int n_max = (kmax >= kmin) ? (kmax / BN + 1) : n_min;
  const int w_row0 = m0 + wave * 32;
  const bool wave_valid = w_row0 < sq;
  const int w_q32 = min(w_row0 >> 5, p.ds_nq32 - 1);
  const E* __restrict__ ds_row = (const E*)p.ds_ws + ((((int64_t)b * p.h + h) * p.ds_head_tiles + ds_row_start(w_q32, p.ds_c1, p.ds_jb, p.ds_np64)) << 10) + lane * 8;
  const int row_cnt = 2 * min((w_q32 + p.ds_c1) >> 1, p.ds_np64);   // sub-tiles this packed row holds: nothing is fetched from behind it

  constexpr int RPD = 1024 / ROW_BYTES, NDMA = TILE_BYTES / 1024, DPW = NDMA / NW;
  static_assert(NDMA % NW == 0 && DPW >= 1, "tile does not divide over the waves");
  // every wave issues the same number of DMA instructions per tile (DPW for K, + 4 for its dS when its rows exist), so "tile n has landed" is a
  // literal vmcnt: sub-tiles past the last key block are fetched from the last one that exists (they are inactive and never read from LDS)
  auto load_tile = [&](int n, int st) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      const int idx = wave * DPW + i;
      const int row = idx * RPD + lane / CPR;
      const int c = (lane % CPR) ^ swz16<D>(row);
      const int key = min(n * BN + row, sk - 1);   // rows past the last key: clamped copies, their dS is zero
      lds_dma_16B(kp + (int64_t)key * p.k_rs + c * 8, lds + st * TILE_BYTES + idx * 1024);
    }
    if (wave_valid) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k32 = min(2 * n + (j >> 1), row_cnt - 1);
        lds_dma_16B(ds_row + ((int64_t)k32 << 10) + (j & 1) * 512, lds + OFF_DS + st * DS_BUF + wave * DS_WAVE + j * 1024);
      }
    }
  };
  auto wait_tile = [&](bool next_in_flight) __attribute__((always_inline)) {   // this wave's DMA of the current tile has landed
    if (!next_in_flight) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (wave_valid) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW + 4) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW) : "memory");
  };
  // transposed K fragments (as fa_bwd_dq_kernel) and transposed dS fragments
  const int tr_i = lane & 15, tr_half = (lane >> 4) & 1, tr_rr = tr_i >> 2, tr_cc = tr_i & 3;
  int tr_base[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const int row = 8 * s2 + 4 * hi + tr_rr;
    tr_base[s2] = tile_off<D>(row, 2 * tr_half + (tr_cc >> 1)) + (tr_cc & 1) * 8;
  }
  // dS sub-tile image: half (queries 16*tr_half ..) * 1024 + slot(key = 16t + 8s + 4hi + rr, writer half = cc & 1) * 16 + (cc >> 1) * 8
  const int ds_lane = tr_half * 1024 + hi * 128 + (tr_cc & 1) * 64 + tr_rr * 16 + (tr_cc >> 1) * 8 + OFF_DS + wave * DS_WAVE;

  f32x16 dq_acc[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq_acc[i][r] = 0.f;

  if (n_min < n_max) load_tile(n_min, 0);
  if (n_min + 1 < n_max) load_tile(n_min + 1, 1);
  auto tile = [&](auto curc, int n) __attribute__((always_inline)) {
    constexpr int cur = decltype(curc)::value;
    wait_tile(n + 1 < n_max);
    __syncthreads();   // tile n is in LDS for every wave, and every wave is done with tile n - 1, whose stage the next load overwrites
    if (n + 2 < n_max) load_tile(n + 2, (cur + 2) % NST);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int k0 = n * BN + 32 * kb;
      if (!(wave_valid && ds_tile_active(w_row0, k0, sq, sk, shift, p.wl, p.wr))) continue;
      V8 f[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int a = ds_lane + cur * DS_BUF + kb * 2048 + t * 512;
        const s16x4 lo = lds_read_tr16(lds + a), hi4 = lds_read_tr16(lds + a + 256);
        f[t] = combine_tr<V8>(lo, hi4);
      }
      constexpr int NOPS = 2 * DB, PFT = 3;
      s16x4 tlo[PFT], thi[PFT];
      auto rd = [&](int i) __attribute__((always_inline)) {
        const int db = i % DB, t = i / DB;
        const int base = cur * TILE_BYTES + kb * 32 * ROW_BYTES + 16 * t * ROW_BYTES;
        tlo[i % PFT] = lds_read_tr16(lds + base + (tr_base[0] ^ (db << 6)));
        thi[i % PFT] = lds_read_tr16(lds + base + (tr_base[1] ^ (db << 6)));
      };
#pragma unroll
      for (int i = 0; i < PFT - 1; ++i) rd(i);
#pragma unroll
      for (int i = 0; i < NOPS; ++i) {
        if (i + PFT - 1 < NOPS) rd(i + PFT - 1);
        dq_acc[i % DB] = T::mfma(combine_tr<V8>(tlo[i % PFT], thi[i % PFT]), f[i / DB], dq_acc[i % DB]);
      }
    }
  };
  for (int n = n_min; n < n_max; n += NST) {
    tile(std::integral_constant<int, 0>{}, n);
    if (n + 1 < n_max) tile(std::integral_constant<int, 1>{}, n + 1);
    if (n + 2 < n_max) tile(std::integral_constant<int, 2>{}, n + 2);
  }
  __syncthreads();   // every wave is done with the tiles: the staging below reuses their LDS
  if (!wave_valid) return;
  E* dqtile = (E*)p.dq + (int64_t)b * p.dq_bs + (int64_t)w_row0 * p.dq_rs + (int64_t)h * p.dq_hs;
  store_tile_via_lds<E, D>(lds + wave * 32 * (ROW_BYTES + 16), dq_acc, p.scale, dqtile, p.dq_rs, sq - w_row0, lane);
}

// Take one ready dQ item off queue `ctrl` / `slots` (thread 0): -1 = nothing published at the moment.  A counting semaphore (FZ_AVAIL: published and
// unclaimed items) in front of a ticket (FZ_HEAD), all of it returning atomic adds: a claim costs two of them whatever the contention.  (Measured and
// replaced, profiles/r04_bwd_fused.txt: head < tail checked with loads + compare-exchange -- loads of these words may be served from this XCD's L2 with a
// value another XCD has long replaced, and with up to 256 workgroups finishing together the compare-exchange took 60-140 retries per item.)  A workgroup
// whose decrement finds nothing gives it back; if the word is positive after giving back (a publication slipped in between), it tries again -- so of the
// workgroups that fail around a publication the last one to give back always sees it, and nothing is left behind.  Every loop is bounded: a claimed slot
// whose content does not appear (its publisher is between its two atomics) is polled with read-modify-writes; on a time-out the launch's error flag is set and the
// binders raise after the call (fa_bwd_fused_status, include/fa_gfx950.h): the block of dq it stood for was not written.
static __device__ __forceinline__ int fz_pop(int32_t* z, int32_t* ctrl, int32_t* slots) {
  for (int spin = 0; spin < (1 << 12); ++spin) {
    if (__hip_atomic_fetch_add(ctrl + FZ_AVAIL, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0) {
      const int hd = __hip_atomic_fetch_add(ctrl + FZ_HEAD, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int tries = 0; tries < (1 << 18); ++tries) {
        const int v = __hip_atomic_fetch_add(slots + hd, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != 0) return v - 1;
        __builtin_amdgcn_s_sleep(4);
      }
      __hip_atomic_store(z + FZ_ERR, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return -1;
    }
    if (__hip_atomic_fetch_add(ctrl + FZ_AVAIL, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0) return -1;   // still nothing after giving back
  }
  return -1;
}

template <int D> constexpr int fused_smem_bytes() {
  constexpr int DKDV = 8 * 32 * D * 2 + 4 * 64 * D * 2 + 4 * 64 * 4, DQ = 3 * 64 * D * 2 + 3 * 8 * 4096, STAGE = 256 * (D * 2 + 16);
  return (DKDV > DQ ? (DKDV > STAGE ? DKDV : STAGE) : (DQ > STAGE ? DQ : STAGE));
}

// One workgroup per CU, resident for the whole launch.  Workgroup w serves XCD w % 8 (where it is observed to run; performance only): it takes the next
// key-block item of that XCD (x + 8k, the order the dK/dV kernel's grid is dealt in), runs the dK/dV part, then computes dQ for every block on the XCD's
// ready queue, and repeats; once the key blocks are handed out it stays until the XCD's last key block is finished and the queue is empty.  (As a grid of
// one workgroup per key block -- the first version -- the hardware dispatcher deals workgroups to the XCDs strictly in turn, so an XCD that fell behind
// stalled the hand-out to all eight: 14-17 % of the CUs idle mid-launch, and the CUs of finished workgroups were lost to the tail, profiles/r04_bwd_fused.txt.)
template <typename E, int D>
__global__ void __launch_bounds__(512, 2) fa_bwd_fused_kernel(const BwdK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char FA_LDS* lds = (char FA_LDS*)smem;
  constexpr int OFF_ITEM = fused_smem_bytes<D>();   // one word behind everything either part uses
  const int qx = blockIdx.x & 7;
  int32_t* ctrl = p.fuse_sync + FZ_CTRL + qx * FZ_CTRL_STRIDE;
  int32_t* slots = p.fuse_sync + FZ_COUNTERS + (int64_t)p.fuse_items * p.fuse_line + (int64_t)qx * p.fuse_items;
  const int total_x = (p.fuse_total - qx + 7) >> 3;   // key-block items of this XCD
  for (;;) {
    __syncthreads();   // the part before is done with the LDS
    if (threadIdx.x == 0) *(int FA_LDS*)(lds + OFF_ITEM) = __hip_atomic_fetch_add(ctrl + FZ_NEXT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int kx = __builtin_amdgcn_readfirstlane(*(const int FA_LDS*)(lds + OFF_ITEM));
    const bool have_kv = kx < total_x;
    if (have_kv) {
      fa_bwd_dkdv_body<E, D, D, FEAT_EXACT, true>(p, qx + 8 * kx);
      if (threadIdx.x == 0) {   // this key block is over: everything it publishes is published (its atomics have returned)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(ctrl + FZ_DONE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    for (;;) {   // dQ for whatever is ready
      __syncthreads();
      if (threadIdx.x == 0) {
        int item = fz_pop(p.fuse_sync, ctrl, slots);
        // no key block left to take: stay until the XCD's last key block is finished and the queue is empty (bounded, ~2 ms).  Nobody waits for a waiter --
        // publishers never depend on consumers -- and correctness never depends on the waiting: a publisher drains the queue itself.
        if (item < 0 && !have_kv) {
          for (int w = 0; w < 512; ++w) {
            const int done = __hip_atomic_fetch_add(ctrl + FZ_DONE, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            item = fz_pop(p.fuse_sync, ctrl, slots);
            if (item >= 0 || done >= total_x) break;
            __builtin_amdgcn_s_sleep(96);
          }
        }
        if (item >= 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *(int FA_LDS*)(lds + OFF_ITEM) = item;
      }
      __syncthreads();
      const int item = __builtin_amdgcn_readfirstlane(*(const int FA_LDS*)(lds + OFF_ITEM));
      if (item < 0) break;
      const int bh = item / p.nmb;
      fa_bwd_dq_from_ds<E, D>(p, lds, bh / p.h, bh % p.h, item - bh * p.nmb);
    }
    if (!have_kv) break;
  }
}

template <typename E, int D>
static int launch_fused_t(const BwdK& p, hipStream_t stream) {
  constexpr int smem = fused_smem_bytes<D>() + 16;
  auto kern = fa_bwd_fused_kernel<E, D>;
  static std::atomic<unsigned long long> attr_mask{0};
  if (ensure_dyn_lds(attr_mask, (const void*)kern, smem, true) != 0) return -1;
  static_assert(FZ_ERR == 0, "the error flag is the sync area's first word");
  const int keep = p.fuse_keep_err ? 1 : 0;   // (a later chunk of the same call: the flag accumulates)
  if (hipMemsetAsync(p.fuse_sync + keep, 0, (size_t)(fz_sync_words(p.fuse_items, p.fuse_line) - keep) * 4, stream) != hipSuccess) return -1;
  const long long total = units_grid(p.k_units, p.k_unit_size);
  static const int n_cu = [] { int dev = 0, n = 0; return (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256; }();
  BwdK q = p;
  q.fuse_total = (int)total;
  hipLaunchKernelGGL(kern, dim3((unsigned)(total < n_cu ? total : n_cu)), dim3(512), smem, stream, q);   // one workgroup per CU (the LDS footprint admits no second one)
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// The fused backward applies to plain attention (no softcap / ALiBi / dropout), fixed-length batches, head dim 64 / 128, no left window, sk >= sq
// (fa_api.cpp bwd_fused_bytes checks the same and sizes the workspace); -2 = does not apply, nothing enqueued.
int launch_bwd_fused(const BwdK& p, int dtype_bf16, int d, hipStream_t stream) {
  if (!p.ds_ws || !p.fuse_sync || p.cu_q || p.cu_k || p.seqused_q || p.seqused_k || p.k_list || p.d_chunks > 0) return -2;
  if (p.softcap > 0.f || p.alibi || p.rng || p.wl >= 0 || p.sk < p.sq) return -2;
  if (dtype_bf16) {
    if (d == 128) return launch_fused_t<__bf16, 128>(p, stream);
    if (d == 64) return launch_fused_t<__bf16, 64>(p, stream);
  } else {
    if (d == 128) return launch_fused_t<_Float16, 128>(p, stream);
    if (d == 64) return launch_fused_t<_Float16, 64>(p, stream);
  }
  return -2;
}
#endif  // FA_BWD_PART == 0 || FA_BWD_PART == 3

#if FA_BWD_PART != 3
// ------------------------------------------------------------------------------------------------
// dQ
// ------------------------------------------------------------------------------------------------
template <typename E, int D, int DV, int NW, int FEAT>
__global__ void __launch_bounds__(NW * 64, D > 128 ? 1 : 2) fa_bwd_dq_kernel(const BwdK p) {
  constexpr bool XFORM = (FEAT & (FEAT_CAP | FEAT_ALIBI)) != 0;
  constexpr bool F_CAP = (FEAT & FEAT_CAP) != 0, F_ALIBI = (FEAT & FEAT_ALIBI) != 0, F_DROP = (FEAT & FEAT_DROP) != 0;
  using T = ElemTraits<E>;
  using V8 = typename T::v8;
  using V4 = typename T::v4;
  constexpr int NT = NW * 64;
  constexpr int BM = NW * 32, BN = 64;
  constexpr int CPR = D / 8, ROW_BYTES = D * 2, TILE_BYTES = BN * ROW_BYTES;
  constexpr int KS = DV / 16, DB = DV / 32, CV = DV / 8;
  static_assert(DV % 32 == 0 && DV <= D && 2 * DV >= D, "DV: a multiple of 32 in [D/2, D]");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char FA_LDS* lds = (char FA_LDS*)smem;  // K0 | K1 | V0 | V1

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, qi = lane & 31;

  int b, h, m_block;
  if (p.q_list) {  // varlen: non-empty query blocks only, heaviest first (fa_varlen_schedule_kernel)
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

  const float cs = XFORM ? kLog2e : p.scale_log2;
  const bool use_alibi = F_ALIBI && (FEAT != FEAT_ALL || p.alibi != nullptr);
  const bool use_cap = F_CAP && (FEAT != FEAT_ALL || p.softcap > 0.f);
  const float slope = use_alibi ? p.alibi[(int64_t)b * p.alibi_bs + h] : 0.f;
  const bool drop = F_DROP && (FEAT != FEAT_ALL || p.rng != nullptr);
  const uint32_t drop_key = drop ? drop_bh_key(p.rng, b * p.h + h) : 0u;

  const int cvr = p.d_chunks > 0 ? p.d_chunks : CV;   // head dims between the built sizes: see the dK/dV kernel
  const bool bounded = cvr < CV;
  const E* zsrc = (const E*)&fa_zero_chunk_bwd;
  // Q and dO fragments (B operands), LSE and delta (lane-local scalars).  The fragments are staged through the (still
  // idle) LDS by coalesced DMA -- each wave its own 32 rows -- instead of 16-byte loads at row stride.
  V8 qf[KS], dof[KS];
  {
    constexpr int RPDQ = 1024 / ROW_BYTES, QDPW = 32 * ROW_BYTES / 1024;  // rows per DMA instruction, instructions per wave
    const E* qsrc = (const E*)p.q + q_boff + q_row0 * p.q_rs + (int64_t)h * p.q_hs;
    const E* dosrc = (const E*)p.dout + do_boff + q_row0 * p.do_rs + (int64_t)h * p.do_hs;
    const int fr0 = wave * 32 * ROW_BYTES + qi * ROW_BYTES + ((hi ^ swz16<D>(qi)) << 4);
#pragma unroll
    for (int which = 0; which < 2; ++which) {
#pragma unroll
      for (int i = 0; i < QDPW; ++i) {
        const int row = wave * 32 + i * RPDQ + lane / CPR;
        int c = (lane % CPR) ^ swz16<D>(row);
        if (DV < D) c = c < CV ? c : 0;
        const int grow = min(m0 + row, sq - 1);
        const E* src = (which ? (dosrc + (int64_t)grow * p.do_rs) : (qsrc + (int64_t)grow * p.q_rs)) + c * 8;
        if (bounded && c >= cvr) src = zsrc;
        lds_dma_16B(src, lds + (wave * QDPW + i) * 1024);
      }
      lds_dma_wait_all();
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const u32x4 x = *(const u32x4 FA_LDS*)(lds + (fr0 ^ (ks << 5)));
        if (which) dof[ks] = bitcast_u32x4<V8>(x); else qf[ks] = bitcast_u32x4<V8>(x);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads are done before the next DMA may overwrite the rows
    }
    __syncthreads();  // the K/V tile DMA below reuses this LDS
  }
  float lse_l = INFINITY, delta_l = 0.f;
  if (row_valid) {
    const int64_t base = p.cu_q ? ((int64_t)h * p.total_q + q_row0) : (((int64_t)b * p.h + h) * p.sq);
    lse_l = p.lse[base + my_row] * kLog2e;
    delta_l = p.delta[base + my_row];
  }

  // K/V tiles: global -> LDS by DMA (source-side swizzle, rows past the last key clamped; see fa_fwd_il.hip)
  constexpr int RPD = 1024 / ROW_BYTES, NDMA = TILE_BYTES / 1024, DPW = NDMA / NW;
  static_assert(NDMA % NW == 0 && DPW >= 1, "tile does not divide over the waves");
  auto load_tile = [&](int n, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      const int idx = wave * DPW + i;
      const int row = idx * RPD + lane / CPR;
      int c = (lane % CPR) ^ swz16<D>(row);
      if (DV < D) c = c < CV ? c : 0;
      const int key = min(n * BN + row, sk - 1);
      const E* ks_ = kp + (int64_t)key * p.k_rs + c * 8;
      const E* vs_ = vp + (int64_t)key * p.v_rs + c * 8;
      if (bounded && c >= cvr) { ks_ = zsrc; vs_ = zsrc; }
      lds_dma_16B(ks_, lds + buf * TILE_BYTES + idx * 1024);
      lds_dma_16B(vs_, lds + (2 + buf) * TILE_BYTES + idx * 1024);
    }
  };

  const int row_off = qi * ROW_BYTES;
  const int rswz = swz16<D>(qi);
  const int tr_i = lane & 15, tr_half = (lane >> 4) & 1, tr_rr = tr_i >> 2, tr_cc = tr_i & 3;
  // offset(db, s) = tr_base[s] ^ (db << 6): the d-block index only enters through XOR on the 64-B chunk bits
  int tr_base[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int row = 8 * s + 4 * hi + tr_rr;
    tr_base[s] = tile_off<D>(row, 2 * tr_half + (tr_cc >> 1)) + (tr_cc & 1) * 8;
  }

  f32x16 dq_acc[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq_acc[db][r] = 0.f;

  if (n_min < n_max) {
    load_tile(n_min, 0);
    lds_dma_wait_all();
    __syncthreads();
  }

  // per-lane operand bases (see the dK/dV kernel): k-step ks of a row-major fragment = k0 ^ (ks << 5)
  const int k0 = row_off + ((hi ^ rswz) << 4);
  auto opaque = [](int x) __attribute__((always_inline)) { asm volatile("" : "+v"(x)); return x; };

  auto tile = [&](auto curc, int n) __attribute__((always_inline)) {
    constexpr int cur = decltype(curc)::value;
    constexpr int KB_OFF = cur * TILE_BYTES, VB_OFF = (2 + cur) * TILE_BYTES;
    const int kv0 = n * BN;
    const bool has_next = n + 1 < n_max;
    if (has_next) load_tile(n + 1, cur ^ 1);  // lands in the other buffers while this tile is computed

    const bool active = wave_valid && (kv0 <= w_kmax) && (kv0 + BN - 1 >= w_kmin);
    if (active) {
      const bool need_mask = (kv0 + BN - 1 > w_full_hi) || (kv0 < w_full_lo);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int sub = kb * 32 * ROW_BYTES;
        // S^T[key][query] = K.Q^T ; dP^T[key][query] = V.dO^T   (column = query = lane); the two chains alternate and
        // the LDS operands are read PF-1 ops ahead
        f32x16 s, dp;
        {
          constexpr int NOPS = 2 * KS, PF = FA_BWD_PF;
          u32x4 ra[PF];
          const int k0p = opaque(k0);
          auto rd = [&](int j) __attribute__((always_inline)) {
            const int ks = j >> 1;
            ra[j % PF] = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)((((j & 1) ? VB_OFF : KB_OFF) + sub) + (k0p ^ (ks << 5)));
          };
#pragma unroll
          for (int j = 0; j < PF - 1; ++j) rd(j);
#pragma unroll
          for (int j = 0; j < NOPS; ++j) {
            if (j + PF - 1 < NOPS) rd(j + PF - 1);
            __builtin_amdgcn_sched_barrier(0);
            const int ks = j >> 1;
            f32x16 c = (j & 1) ? dp : s;
            if (j < 2) {
#pragma unroll
              for (int r = 0; r < 16; ++r) c[r] = 0.f;
            }
            if ((j & 1) == 0) s = T::mfma(bitcast_u32x4<V8>(ra[j % PF]), qf[ks], c);
            else dp = T::mfma(bitcast_u32x4<V8>(ra[j % PF]), dof[ks], c);
          }
        }
        f32x16 dcap;
        if constexpr (XFORM) {
          const float rcap = use_cap ? 1.f / p.softcap : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kv0 + 32 * kb + acc_row(r, hi);
            float y = s[r] * p.scale;
            if constexpr (F_CAP) {
              dcap[r] = 1.f;
              if (use_cap) {
                const float t = fast_tanh(y * rcap);
                y = p.softcap * t;
                dcap[r] = 1.f - t * t;
              }
            }
            if constexpr (F_ALIBI) {
              if (use_alibi) y -= slope * fabsf((float)(my_row + shift - key));
            }
            s[r] = y;
          }
        }
        if (need_mask) {
          const int rel_hi = lim_hi - kv0 - 32 * kb - 4 * hi;
          const int rel_lo = lim_lo - kv0 - 32 * kb - 4 * hi;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int off = acc_row(r, 0);
            s[r] = ((off <= rel_hi) && (off >= rel_lo)) ? s[r] : -INFINITY;
          }
        }
        V8 dsfrag[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = fast_exp2(__builtin_fmaf(s[r], cs, -lse_l));
          float dpe = dp[r];
          if constexpr (F_DROP) {
            if (drop) {  // acc rows 4g..4g+3 are keys key0..key0+3: one hash per group, byte r&3
              const int key0 = kv0 + 32 * kb + 8 * (r >> 2) + 4 * hi;
              const uint32_t bytes = drop_bytes(drop_key, my_row, key0 >> 2);
              dpe = (((bytes >> (8 * (r & 3))) & 0xffu) <= p.drop_thr8) ? dp[r] * p.rp_keep : 0.f;
            }
          }
          float dsv = pv * (dpe - delta_l);
          if constexpr (F_CAP) dsv *= dcap[r];
          dsfrag[r >> 3][r & 7] = (E)dsv;
        }
        // dQ^T[d][query] += K^T[d][key] . dS^T[key][query]; transpose reads PFT-1 ops ahead
        {
          constexpr int NOPS = 2 * DB, PFT = FA_BWD_PFT;
          s16x4 tlo[PFT], thi[PFT];
          const int t0p = opaque(tr_base[0]), t1p = opaque(tr_base[1]);
          auto rd = [&](int i) __attribute__((always_inline)) {
            const int db = i % DB, t = i / DB;
            const int base = KB_OFF + sub + 16 * t * ROW_BYTES;
            tlo[i % PFT] = lds_read_tr16((const char FA_LDS*)(unsigned long)(unsigned)(base + (t0p ^ (db << 6))));
            thi[i % PFT] = lds_read_tr16((const char FA_LDS*)(unsigned long)(unsigned)(base + (t1p ^ (db << 6))));
          };
#pragma unroll
          for (int i = 0; i < PFT - 1; ++i) rd(i);
#pragma unroll
          for (int i = 0; i < NOPS; ++i) {
            if (i + PFT - 1 < NOPS) rd(i + PFT - 1);
            __builtin_amdgcn_sched_barrier(0);
            dq_acc[i % DB] = T::mfma(combine_tr<V8>(tlo[i % PFT], thi[i % PFT]), dsfrag[i / DB], dq_acc[i % DB]);
          }
        }
      }
    }
    lds_dma_wait_all();
    __syncthreads();
  };
  for (int n = n_min; n < n_max; n += 2) {
    tile(std::integral_constant<int, 0>{}, n);
    if (n + 1 < n_max) tile(std::integral_constant<int, 1>{}, n + 1);
  }

  if (!wave_valid) return;
  // dQ tile through the freed K/V buffers: whole-row stores (fa_device.h store_tile_via_lds)
  E* dqtile = (E*)p.dq + dq_boff + (q_row0 + w_row0) * p.dq_rs + (int64_t)h * p.dq_hs;
  store_tile_via_lds<E, D, DV>(lds + wave * 32 * (ROW_BYTES + 16), dq_acc, p.scale, dqtile, p.dq_rs, sq - w_row0, lane, cvr);
}

#endif  // FA_BWD_PART != 3
// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// query rows per dQ workgroup for the schedule BwdK::dq_nw: 4 | 8 waves x 32 rows, or 64 = 4 waves x 64 rows (fa_bwd_w64.hip)
// The launchers come in two halves so that build.py can compile this file twice side by side (-DFA_BWD_PART=1: delta + dK/dV,
// =2: dQ; 0 = everything in one object): the dK/dV and dQ instantiations are independent and dominate the library's build time.
#if FA_BWD_PART == 0 || FA_BWD_PART == 2
int bwd_block_m(int nw) { return (nw == 8 || nw == 64) ? 256 : 128; }
#else
int bwd_block_m(int nw);
#endif
#if FA_BWD_PART == 0 || FA_BWD_PART == 1
int bwd_block_n(int d) { return d > 128 ? 128 : 256; }

template <typename E, int D, int DV>
static int launch_delta_t(const BwdK& p, hipStream_t stream) {
  constexpr int ROWS = 256 / (D / 8);
  dim3 grid((p.sq + ROWS - 1) / ROWS, p.h, p.b);
  hipLaunchKernelGGL((fa_bwd_delta_kernel<E, D, DV>), grid, dim3(256), 0, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <typename E, int D, int DV, int FEAT>
static int launch_dkdv_a(const BwdK& p, hipStream_t stream) {
  constexpr int NWK = D > 128 ? 4 : 8, BMQ = D > 128 ? 32 : 64;
  constexpr int smem = NWK * 32 * D * 2 + 4 * BMQ * D * 2 + 4 * BMQ * 4;
  auto kern = fa_bwd_dkdv_kernel<E, D, DV, FEAT>;
  static std::atomic<unsigned long long> attr_mask{0};
  if (ensure_dyn_lds(attr_mask, (const void*)kern, smem, true) != 0) return -1;   // (operand reads address LDS by byte offset: the dynamic segment starts at 0)
  const long long total = p.k_list ? (long long)p.k_bound * p.h_k : units_grid(p.k_units, p.k_unit_size);
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(NWK * 64), smem, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// trimmed head dims (DV < D) are built plain and as the run-time-checked all-features variant only
template <typename E, int D, int DV>
static int launch_dkdv_t(const BwdK& p, hipStream_t stream) {
  int feat = feat_code(p.softcap > 0.f, p.alibi != nullptr, p.rng != nullptr);
  // plain: FEAT_EXACT (fp32 scaling of every score) unless FA_DKDV_PRESCALE=1 opts into the pre-scaled-K variant (and never under FA_STRICT)
  if (feat == FEAT_NONE && (knobs().strict || !knobs().dkdv_prescale)) feat = FEAT_EXACT;
  if constexpr (DV < D) {
    return feat == FEAT_NONE ? launch_dkdv_a<E, D, DV, FEAT_NONE>(p, stream) : feat == FEAT_EXACT ? launch_dkdv_a<E, D, DV, FEAT_EXACT>(p, stream) : launch_dkdv_a<E, D, DV, FEAT_ALL>(p, stream);
  } else {
    switch (feat) {
      case FEAT_NONE: return launch_dkdv_a<E, D, DV, FEAT_NONE>(p, stream);
      case FEAT_EXACT: return launch_dkdv_a<E, D, DV, FEAT_EXACT>(p, stream);
      case FEAT_CAP: return launch_dkdv_a<E, D, DV, FEAT_CAP>(p, stream);
      case FEAT_ALIBI: return launch_dkdv_a<E, D, DV, FEAT_ALIBI>(p, stream);
      case FEAT_DROP: return launch_dkdv_a<E, D, DV, FEAT_DROP>(p, stream);
      case FEAT_CAP | FEAT_DROP: return launch_dkdv_a<E, D, DV, (FEAT_CAP | FEAT_DROP)>(p, stream);
      case FEAT_ALIBI | FEAT_DROP: return launch_dkdv_a<E, D, DV, (FEAT_ALIBI | FEAT_DROP)>(p, stream);
      default: return launch_dkdv_a<E, D, DV, FEAT_ALL>(p, stream);
    }
  }
}

#endif  // FA_BWD_PART == 0 || FA_BWD_PART == 1
#if FA_BWD_PART == 0 || FA_BWD_PART == 2
template <typename E, int D, int DV, int NW, int FEAT>
static int launch_dq_nw(const BwdK& p, hipStream_t stream) {
  constexpr int smem = 4 * 64 * D * 2 + NW * 32 * 16;  // K/V double buffers (+ the row padding of the staged dQ epilogue)
  auto kern = fa_bwd_dq_kernel<E, D, DV, NW, FEAT>;
  static std::atomic<unsigned long long> attr_mask{0};
  if (ensure_dyn_lds(attr_mask, (const void*)kern, smem, true) != 0) return -1;   // (operand reads address LDS by byte offset: the dynamic segment starts at 0)
  const long long total = p.q_list ? (long long)p.q_bound * p.h : units_grid(p.q_units, p.q_unit_size);
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(NW * 64), smem, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
template <typename E, int D, int DV, int FEAT>
static int launch_dq_f(const BwdK& p, hipStream_t stream) {
  if constexpr (D > 128 || DV < D) return launch_dq_nw<E, D, DV, 4, FEAT>(p, stream);
  else return bwd_block_m(p.dq_nw) == 256 ? launch_dq_nw<E, D, DV, 8, FEAT>(p, stream) : launch_dq_nw<E, D, DV, 4, FEAT>(p, stream);
}
template <typename E, int D, int DV>
static int launch_dq_t(const BwdK& p, hipStream_t stream) {
  const int feat = feat_code(p.softcap > 0.f, p.alibi != nullptr, p.rng != nullptr);
  if constexpr (DV < D) {
    return feat == FEAT_NONE ? launch_dq_f<E, D, DV, FEAT_NONE>(p, stream) : launch_dq_f<E, D, DV, FEAT_ALL>(p, stream);
  } else {
    switch (feat) {
      case FEAT_NONE: return launch_dq_f<E, D, DV, FEAT_NONE>(p, stream);
      case FEAT_CAP: return launch_dq_f<E, D, DV, FEAT_CAP>(p, stream);
      case FEAT_ALIBI: return launch_dq_f<E, D, DV, FEAT_ALIBI>(p, stream);
      case FEAT_DROP: return launch_dq_f<E, D, DV, FEAT_DROP>(p, stream);
      case FEAT_CAP | FEAT_DROP: return launch_dq_f<E, D, DV, (FEAT_CAP | FEAT_DROP)>(p, stream);
      case FEAT_ALIBI | FEAT_DROP: return launch_dq_f<E, D, DV, (FEAT_ALIBI | FEAT_DROP)>(p, stream);
      default: return launch_dq_f<E, D, DV, FEAT_ALL>(p, stream);
    }
  }
}

#endif  // FA_BWD_PART == 0 || FA_BWD_PART == 2

#define FA_BWD_DISPATCH_E(fn, E)                                       \
  switch (d) {                                                         \
    case 128: return fn<E, 128, 128>(p, stream);                       \
    case 64: return fn<E, 64, 64>(p, stream);                          \
    case 256: return fn<E, 256, 256>(p, stream);                       \
    case 96: return fn<E, 128, 96>(p, stream);                         \
    case 32: return fn<E, 64, 32>(p, stream);                          \
    case 192: return fn<E, 256, 192>(p, stream);                       \
    default: return -2;                                                \
  }
#define FA_BWD_DISPATCH(fn)                                            \
  if (dtype_bf16) { FA_BWD_DISPATCH_E(fn, __bf16) } else { FA_BWD_DISPATCH_E(fn, _Float16) }

#if FA_BWD_PART == 0 || FA_BWD_PART == 1
int launch_bwd_delta(const BwdK& p, int dtype_bf16, int d, hipStream_t stream) { FA_BWD_DISPATCH(launch_delta_t) }
int launch_bwd_gsum(const void* src, void* dst, int dtype_bf16, int b, int sk, int h_k, int gs, int d, int64_t dst_bs, int64_t dst_rs, int64_t dst_hs, hipStream_t stream) {
  const long long total = (long long)b * sk * h_k * (d / 8);
  if (total <= 0) return 0;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (dtype_bf16) hipLaunchKernelGGL(fa_bwd_gsum_kernel<__bf16>, grid, dim3(256), 0, stream, (const __bf16*)src, (__bf16*)dst, total, sk, h_k, gs, d / 8, dst_bs, dst_rs, dst_hs);
  else hipLaunchKernelGGL(fa_bwd_gsum_kernel<_Float16>, grid, dim3(256), 0, stream, (const _Float16*)src, (_Float16*)dst, total, sk, h_k, gs, d / 8, dst_bs, dst_rs, dst_hs);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int launch_bwd_dkdv(const BwdK& p, int dtype_bf16, int d, hipStream_t stream) { FA_BWD_DISPATCH(launch_dkdv_t) }
#endif
#if FA_BWD_PART == 0 || FA_BWD_PART == 2
int launch_bwd_dq(const BwdK& p, int dtype_bf16, int d, hipStream_t stream) {
  LastSchedule& ls = last_schedule();
  const bool trimmed = (d == 32 || d == 96 || d == 192);
  ls.bwd_dq_nw = (d > 128 || trimmed) ? 4 : bwd_block_m(p.dq_nw) / 32;
  if (p.dq_nw == 64) {   // 64-rows-per-wave schedule where it applies, else the 8-wave kernel on the same 256-row blocks
    const int rc = launch_bwd_dq_w64(p, dtype_bf16, d, stream);
    if (rc != -2) { ls.bwd_dq_nw = 64; return rc; }
  }
  FA_BWD_DISPATCH(launch_dq_t)
}
#endif

}  // namespace fa
