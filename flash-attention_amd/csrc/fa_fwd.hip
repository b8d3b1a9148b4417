// Fused attention forward for gfx950 (MI355X): S^T = K.Q^T -> online softmax -> O^T += V^T.P^T.
//
// Replaces the reference's flash_fwd_kernel / compute_attn_1rowblock
// (csrc/flash_attn/src/flash_fwd_kernel.h:54-501, softmax.h:128-187, mask.h:111-212,
// block_info.h:12-45) with a CDNA4-native structure -- it is not a translation of the CuTe code:
//
//   * one workgroup = NW waves, each wave owns 32 query rows for the whole key loop
//     (BM = 32*NW rows per workgroup); K/V tiles of BN = 64 keys are staged once per
//     workgroup through LDS (register-staged, double buffered);
//   * both contractions run "swapped" on v_mfma_f32_32x32x16: S^T[key][query] and
//     O^T[d][query].  In the accumulator layout column = lane&31, so EVERY per-query quantity
//     (running max m, running sum l, rescale factor, LSE) is lane-local: the row max is an
//     in-register reduction plus one exchange with lane^32 (v_permlane32_swap), and the O rescale
//     is a per-lane scalar multiply;
//   * P^T needs no shuffle to become the next MFMA's B operand: the k-index permutation the
//     accumulator layout induces on the keys is applied to V instead, by choosing which 4-key
//     groups each ds_read_b64_tr_b16 (LDS transpose read) fetches;
//   * K tile rows are XOR-swizzled at 16-B granularity (conflict-free ds_read_b128 A operands),
//     V tile rows at 64-B granularity (conflict-free transpose reads); measured
//     SQ_LDS_BANK_CONFLICT = 0 (profiles/r01_pmc_v1.txt);
//   * the issue port, not the matrix pipe, is the scarce resource (one VALU slot ~ 4 cycles, an
//     MFMA 32): the key loop is unrolled by two so every LDS address is a loop-invariant register
//     plus an immediate, global tile loads use a scalar base + 32-bit lane offset, and the O
//     rescale is skipped unless some row's maximum grew by more than `rescale_thr` (log2 units;
//     0 = rescale on any growth, exactly the reference's update rule);
//   * causal / sliding-window / key-length masks are evaluated from per-lane key limits only on
//     tiles that straddle a boundary; tiles entirely outside a wave's visible range are skipped
//     per wave, and the block's key range is clipped (reference flash_fwd_kernel.h:90-94);
//   * 1-D grid with an XCD-aware remap so all query blocks of one (batch, head) -- and the
//     query heads sharing a KV head -- run on one XCD and share its L2; heavy (long-key-range)
//     blocks are scheduled first when the mask is right-bounded.
//
// Two schedules of the same pieces (template parameter PP):
//   PP = false  lock-step: per tile {prefetch next tile, QK^T, softmax, PV, store, barrier};
//   PP = true   ping-pong (8 waves): the tile is split into a VALU interval SM(j) and a matrix
//               interval MF(j) = {S_{j+1} = K_{j+1}.Q^T, O += V_j^T.P_j}; waves 4-7 run the same
//               stream one interval late, so each SIMD always has one wave in each kind of interval.
#include <cstdio>
#include <algorithm>
#include <type_traits>

#include "fa_device.h"
#include "fa_kernel_params.h"
#include "fa_launch.h"

namespace fa {

template <int D> FA_DEVINL constexpr int k_swz(int row) { return D >= 128 ? (row & 15) : ((row >> 1) & 7); }
template <int D> FA_DEVINL constexpr int v_swz(int row) { return D >= 128 ? (row & 3) : ((row >> 1) & 1); }

template <int N> using IC = std::integral_constant<int, N>;

// 16 bytes of zeros in global memory: where a DMA lane fetches from when its chunk lies behind the head dim (FwdK::d_chunks)
static __device__ const uint4 fa_zero_chunk = {0u, 0u, 0u, 0u};

// build.py compiles this file twice side by side (-DFA_FWD_PART=1: the bf16 instantiations of fa_fwd_kernel + everything else in
// here, =2: the fp16 instantiations only; 0 = one object): its ~130 kernel instantiations are the slowest unit of the build.
#ifndef FA_FWD_PART
#define FA_FWD_PART 0
#endif
int launch_fwd_bf16(const FwdK& p, int d, int nw, hipStream_t stream);
int launch_fwd_f16(const FwdK& p, int d, int nw, hipStream_t stream);

// D = row pitch of the LDS tiles and of the staging layout (64 / 128 / 256); DV = head dimension actually present in memory and
// contracted over (DV <= D, a multiple of 32).  DV < D are the "trimmed" variants for head dims 32 / 96 / 192 (the reference builds
// those sizes too: static_switch.h:92-110): the QK^T loop runs DV/16 k-steps, the output has DV/32 blocks, tile columns >= DV are
// never read -- the DMA lanes that would fetch them re-fetch column chunk 0 instead -- and never stored.
template <typename E, int D, int DV, int NW, int FEAT, bool PP>
__global__ void __launch_bounds__(NW * 64, D > 128 ? 1 : 2) fa_fwd_kernel(const FwdK p) {
  constexpr bool XFORM = (FEAT & (FEAT_CAP | FEAT_ALIBI)) != 0;  // scores pass through the scaled domain
  constexpr bool F_CAP = (FEAT & FEAT_CAP) != 0, F_ALIBI = (FEAT & FEAT_ALIBI) != 0, F_DROP = (FEAT & FEAT_DROP) != 0;
  using T = ElemTraits<E>;
  using V8 = typename T::v8;
  using V4 = typename T::v4;
  constexpr int BM = NW * 32, BN = 64, CPR = D / 8, NT = NW * 64;
  constexpr int ROW_BYTES = D * 2;
  constexpr int TILE_BYTES = BN * ROW_BYTES;
  constexpr int LD = (BN * CPR) / NT;  // 16-B chunks each thread moves per tile (K and V each)
  constexpr int KS = DV / 16;          // k-steps of the QK^T contraction
  constexpr int DB = DV / 32;          // 32-wide d blocks of the output
  constexpr int CV = DV / 8;           // 16-B chunks of a row that exist in memory
  static_assert(DV % 32 == 0 && DV <= D && 2 * DV >= D, "DV: a multiple of 32 in [D/2, D]");
  static_assert(DV == D || !PP, "trimmed head dims run the lock-step schedule");
  static_assert(D == 64 || D == 128 || D == 256, "tile pitches: 64, 128, 256");
  static_assert(D <= 128 || (NW == 4 && !PP), "D = 256: 4 waves (one per SIMD, 512 registers), lock-step schedule");
  static_assert(LD >= 1 && (BN * CPR) % NT == 0, "tile does not divide over the workgroup");
  static_assert(!PP || NW == 8, "ping-pong schedule pairs waves w and w+4");
  constexpr float kLn2 = 0.6931471805599453f, kLog2e = 1.4426950408889634f;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char FA_LDS* lds = (char FA_LDS*)smem;  // K0 | K1 | V0 | V1

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, qi = lane & 31;

  // ---- which (batch, head, query block) -------------------------------------------------------
  int b, h, m_block, split = 0;
  if (p.work_list) {  // varlen: non-empty blocks only, heaviest first
    if (!work_list_item(p.work_list, blockIdx.x, p.h, p.h_k, b, h, m_block)) return;
  } else {
    const int w = xcd_interleave(blockIdx.x, p.n_units, p.unit_size, p.unit_hpx);
    if (w < 0) return;
    const int nmbs = p.nmb * p.n_splits;  // n_splits >= 1: key splits of one query block are adjacent work items
    const int bh = w / nmbs;
    int mbr = w - bh * nmbs;
    split = mbr % p.n_splits;
    mbr /= p.n_splits;
    m_block = (p.wr >= 0) ? (p.nmb - 1 - mbr) : mbr;
    b = bh / p.h;
    h = bh - b * p.h;
  }
  const int hk = h / p.hk_ratio;

  int sq = p.sq, sk = p.sk;
  int64_t q_row0 = 0, k_row0 = 0;  // first row of this sequence in the packed tensors
  const int bkv = p.kv_batch_idx ? p.kv_batch_idx[b] : b;  // KV-cache row of this batch entry
  int64_t q_boff = (int64_t)b * p.q_bs, k_boff = (int64_t)bkv * p.k_bs, v_boff = (int64_t)bkv * p.v_bs, o_boff = (int64_t)b * p.o_bs;
  if (p.block_table) { k_boff = 0; v_boff = 0; }  // paged cache: the page index supplies the first-dimension offset
  if (p.cu_q) {  // varlen: rows cu[b] .. cu[b+1]-1  (reference block_info.h:17-36)
    const int c0 = p.cu_q[b];
    sq = p.cu_q[b + 1] - c0;
    q_row0 = c0;
    q_boff = 0;
    o_boff = 0;
  }
  if (p.seqused_q) sq = min(sq, p.seqused_q[b]);  // padded batch: only the first seqused_q[b] rows of the entry exist
  if (p.cu_k) {
    const int c0 = p.cu_k[b];
    sk = p.cu_k[b + 1] - c0;
    k_row0 = c0;
    k_boff = 0;
    v_boff = 0;
  }
  if (p.block_table) k_row0 = 0;  // paged K/V: the page table supplies the rows, cu_seqlens_k only the lengths
  if (p.seqused_k) sk = min(p.seqused_k[b] + p.seqused_add, (p.cu_k && !p.block_table) ? sk : p.sk);  // keys in use, never beyond the addressable capacity; inside a packed batch never beyond the entry's slot (as the backward: include/fa_gfx950.h)
  if (p.leftpad_k) {  // a left-padded sequence starts at row leftpad_k[b] (reference block_info.h:17-36)
    const int lp = p.leftpad_k[b];
    sk = max(0, sk - lp);
    k_row0 += lp;
  }
  const int m0 = m_block * BM;
  if (m0 >= sq) return;
  // Head packing (FwdK::pack_g = g > 1; decode / short chunks with grouped heads): this "head" is KV head h and its rows are the
  // g * Sq (query, head-in-group) pairs, row r = query r / g of query head h * g + r % g -- K/V stream once per KV head.
  // Masks, Q / O / LSE addresses and ALiBi slopes go through (row / g, row % g); everything else works on packed rows.
  const int g = p.pack_g;
  const bool packed = g > 1;
  const int cvr = p.d_chunks > 0 ? p.d_chunks : CV;   // 16-byte chunks of a row that exist in memory (< CV: the rest reads as zeros)
  const bool bounded = cvr < CV;
  const E* zsrc = (const E*)&fa_zero_chunk;
  auto q_of = [&](int row) __attribute__((always_inline)) { return packed ? row / g : row; };

  const E* __restrict__ qp = (const E*)p.q + q_boff + q_row0 * p.q_rs + (int64_t)h * g * p.q_hs;
  const E* __restrict__ kp = (const E*)p.k + k_boff + k_row0 * p.k_rs + (int64_t)hk * p.k_hs;
  const E* __restrict__ vp = (const E*)p.v + v_boff + k_row0 * p.v_rs + (int64_t)hk * p.v_hs;
  E* __restrict__ op = (E*)p.o + o_boff + q_row0 * p.o_rs + (int64_t)h * g * p.o_hs;
  float* __restrict__ lsep = p.cu_q ? (p.lse + (int64_t)h * p.total_q + q_row0)
                                    : (p.lse + ((int64_t)b * p.h + h) * p.sq);  // packed: (b, h*g + r%g, r/g) == this base + (r%g)*(sq/g) + r/g

  // ---- key range of the block, per-wave and per-lane visibility limits --------------------------
  const int sq_true = packed ? sq / g : sq;
  const int shift = sk - sq_true;  // bottom-right alignment
  const int blk_last = min(m0 + BM, sq) - 1;
  int kmax = sk - 1, kmin = 0;
  if (p.wr >= 0) kmax = min(kmax, q_of(blk_last) + shift + p.wr);
  if (p.wl >= 0) kmin = max(0, q_of(m0) + shift - p.wl);
  int n_min = kmin / BN;
  int n_max = (kmax >= kmin) ? (kmax / BN + 1) : n_min;
  if (p.n_splits > 1) {  // this workgroup's share of the key tiles (may be empty)
    n_min = max(n_min, split * p.split_tiles);
    n_max = max(n_min, min(n_max, (split + 1) * p.split_tiles));
  }
  const int n_tiles = n_max - n_min;

  const int w_row0 = m0 + wave * 32;
  const int w_row1 = min(w_row0 + 31, sq - 1);
  const bool wave_valid = w_row0 < sq;
  const int w_q0 = q_of(w_row0), w_q1 = q_of(w_row1);  // first / last query of the wave's rows
  const int w_kmax = (p.wr >= 0) ? min(sk - 1, w_q1 + shift + p.wr) : sk - 1;   // last key any row sees
  const int w_kmin = (p.wl >= 0) ? max(0, w_q0 + shift - p.wl) : 0;             // first key any row sees
  const int w_full_hi = (p.wr >= 0) ? min(sk - 1, w_q0 + shift + p.wr) : sk - 1;  // keys <= this: visible to all rows
  const int w_full_lo = (p.wl >= 0) ? (w_q1 + shift - p.wl) : 0;                  // keys >= this: visible to all rows

  const int my_row = w_row0 + qi;
  const bool row_valid = my_row < sq;
  const int my_q = q_of(my_row);           // query index of this lane's row
  const int my_hh = my_row - my_q * g;     // head within the group (0 unless packed)
  const int lim_hi = (p.wr >= 0) ? min(sk - 1, my_q + shift + p.wr) : sk - 1;
  const int lim_lo = (p.wl >= 0) ? (my_q + shift - p.wl) : 0;

  // softcap / ALiBi / dropout are compile-time variants (FEAT) so the common kernel carries none of them
  const float cs = XFORM ? kLog2e : p.scale_log2;  // multiplier taking S to the log2 domain
  const bool use_alibi = F_ALIBI && (FEAT != FEAT_ALL || p.alibi != nullptr);
  const bool use_cap = F_CAP && (FEAT != FEAT_ALL || p.softcap > 0.f);
  const float slope = use_alibi ? p.alibi[(int64_t)b * p.alibi_bs + h * g + my_hh] : 0.f;
  // dropout: per-(batch, head) stream key and this lane's row base
  const bool drop = F_DROP && (FEAT != FEAT_ALL || p.rng != nullptr);
  const uint32_t drop_key = drop ? drop_bh_key(p.rng, b * p.h + h) : 0u;
  uint8_t* rv_row = (drop && p.randval) ? (p.randval + (int64_t)b * p.rv_bs + (int64_t)h * p.rv_hs + (q_row0 + my_row) * p.rv_rs) : nullptr;
  const float thr = p.rescale_thr;

  // ---- Q fragments (B operand of S^T = K.Q^T): lane = query row, 8 consecutive d per k-step -----
  V8 qf[KS];
  const bool q_staged = !PP && sq >= 64 && !packed;  // lock-step schedule: the Q block goes through LDS by coalesced DMA (prologue below)
  if (!q_staged) {  // few query rows (decode), packed heads or ping-pong schedule: 16-byte loads at row stride
    const E* qrow = qp + (int64_t)my_q * p.q_rs + (int64_t)my_hh * p.q_hs + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = bitcast_u32x4<V8>(ld_global_16B(qrow + 16 * ks, row_valid && 2 * ks + hi < cvr));
  }

  // ---- staging: global -> registers -> LDS.  Tile n of K starts at the uniform address
  // kp + n*BN*k_rs; each thread adds a fixed 32-bit byte offset (scalar-base + lane-offset loads). ----
  u32x4 kreg[LD], vreg[LD];
  unsigned kvoff_k[LD], kvoff_v[LD];
  int st_k[LD], st_v[LD], ld_row[LD];
#pragma unroll
  for (int i = 0; i < LD; ++i) {
    const int idx = tid + i * NT;
    const int row = idx / CPR, ch = idx % CPR;
    ld_row[i] = row;
    kvoff_k[i] = (unsigned)(row * (int)p.k_rs + ch * 8) * 2u;
    kvoff_v[i] = (unsigned)(row * (int)p.v_rs + ch * 8) * 2u;
    st_k[i] = row * ROW_BYTES + ((ch ^ k_swz<D>(row)) << 4);
    st_v[i] = row * ROW_BYTES + (((((ch >> 2) ^ v_swz<D>(row)) << 2) | (ch & 3)) << 4);
  }
  // first row of key tile n: contiguous cache, or page block_table[b][n*BN / page] of a paged cache
  // (reference flash_fwd_kernel.h:579-590; pages are multiples of 64 keys, so a tile never straddles two pages)
  auto tile_row_off = [&](int n, int64_t bs, int64_t rs) __attribute__((always_inline)) -> int64_t {
    if (!p.block_table) return (int64_t)n * BN * rs;
    const int key0 = n * BN;
    const int page = key0 / p.page_size;
    const int blk = p.block_table[(int64_t)b * p.block_table_bs + page];
    return (int64_t)blk * bs + (int64_t)(key0 - page * p.page_size) * rs;
  };
  auto load_k = [&](int n) __attribute__((always_inline)) {
    const char* base = (const char*)(kp + tile_row_off(n, p.k_bs, p.k_rs));
#pragma unroll
    for (int i = 0; i < LD; ++i) kreg[i] = ld_global_16B(base + kvoff_k[i], n * BN + ld_row[i] < sk);
  };
  auto load_v = [&](int n) __attribute__((always_inline)) {
    const char* base = (const char*)(vp + tile_row_off(n, p.v_bs, p.v_rs));
#pragma unroll
    for (int i = 0; i < LD; ++i) vreg[i] = ld_global_16B(base + kvoff_v[i], n * BN + ld_row[i] < sk);
  };
  auto store_k = [&](auto bufc) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value;
#pragma unroll
    for (int i = 0; i < LD; ++i) *(u32x4 FA_LDS*)(lds + buf * TILE_BYTES + st_k[i]) = kreg[i];
  };
  auto store_v = [&](auto bufc) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value;
#pragma unroll
    for (int i = 0; i < LD; ++i) *(u32x4 FA_LDS*)(lds + (2 + buf) * TILE_BYTES + st_v[i]) = vreg[i];
  };

  // LDS-DMA staging (lock-step schedule): global_load_lds, 1 KiB per wave instruction, no staging registers and no
  // ds_write pass.  The destination is lane-linear, so the XOR swizzles are applied to the per-lane SOURCE chunk; rows
  // past the last key are clamped to the last key (finite data; their scores are masked to -inf).
  constexpr int RPD = 1024 / ROW_BYTES, NDMA = TILE_BYTES / 1024, DPW = NDMA / NW;
  static_assert(NDMA % NW == 0 && DPW >= 1, "tile does not divide over the waves");
  auto dma_tile = [&](auto isvc, int buf, int n) __attribute__((always_inline)) {
    constexpr bool ISV = decltype(isvc)::value != 0;
    const int64_t rs = ISV ? p.v_rs : p.k_rs;
    const E* base = (ISV ? vp : kp) + tile_row_off(n, ISV ? p.v_bs : p.k_bs, rs);
    char FA_LDS* dst = lds + (ISV ? 2 + buf : buf) * TILE_BYTES + wave * DPW * 1024;
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      const int row = (wave * DPW + i) * RPD + lane / CPR, pc = lane % CPR;
      const int grow = min(n * BN + row, sk - 1) - n * BN;
      int c = ISV ? ((((pc >> 2) ^ v_swz<D>(row)) << 2) | (pc & 3)) : (pc ^ k_swz<D>(row));
      if (DV < D) c = c < CV ? c : 0;  // columns past the head dimension: never read from LDS, fetch something that exists
      const E* src = base + (int64_t)grow * rs + c * 8;
      if (bounded) src = c < cvr ? src : zsrc;
      lds_dma_16B(src, dst + i * 1024);
    }
  };

  // ---- per-lane LDS read addresses (loop invariant; buffers and sub-tiles are immediates) ----------
  int kaddr[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) kaddr[ks] = qi * ROW_BYTES + (((2 * ks + hi) ^ k_swz<D>(qi)) << 4);
  const int tr_i = lane & 15, tr_half = (lane >> 4) & 1;
  const int tr_rr = tr_i >> 2, tr_cc = tr_i & 3;
  int vaddr[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
    vaddr[db] = (4 * hi + tr_rr) * ROW_BYTES + ((db ^ v_swz<D>(tr_rr)) << 6) + tr_half * 32 + tr_cc * 8;

  // ---- online-softmax state (per lane = per query row; both half-waves keep identical m) ----------
  f32x16 o_acc[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[db][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  f32x16 s[2];
  V8 pf[4];

  auto tile_active = [&](int j) __attribute__((always_inline)) {  // j relative to n_min
    const int kv0 = (n_min + j) * BN;
    return wave_valid && (j < n_tiles) && (kv0 <= w_kmax) && (kv0 + BN - 1 >= w_kmin);
  };

  // S^T[key][query] of the tile in K buffer `buf`; operand reads run PF k-steps ahead of their MFMAs
  auto qk = [&](auto bufc) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value;
    const char FA_LDS* kbuf = lds + buf * TILE_BYTES;
    constexpr int PF = 3;
    u32x4 kfrag[PF][2];
#pragma unroll
    for (int ks = 0; ks < PF - 1 && ks < KS; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) kfrag[ks % PF][kb] = *(const u32x4 FA_LDS*)(kbuf + kaddr[ks] + kb * 32 * ROW_BYTES);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int nx = ks + PF - 1;
      if (nx < KS) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) kfrag[nx % PF][kb] = *(const u32x4 FA_LDS*)(kbuf + kaddr[nx] + kb * 32 * ROW_BYTES);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch above this step's MFMAs (hipcc otherwise sinks it to the use)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        f32x16 c = s[kb];
        if (ks == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) c[r] = 0.f;
        }
        s[kb] = T::mfma(bitcast_u32x4<V8>(kfrag[ks % PF][kb]), qf[ks], c);
      }
    }
  };

  // mask + online softmax of s -> pf (P^T as B operand), updates m_run / l_run / o_acc scale
  auto softmax_step = [&](int j) __attribute__((always_inline)) {
    const int kv0 = (n_min + j) * BN;
    if constexpr (XFORM) {  // softcap / ALiBi: move to the scaled domain first (reference utils.h:395-409, alibi.h)
      const float rcap = use_cap ? 1.f / p.softcap : 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float y = s[kb][r] * p.scale;
          if constexpr (F_CAP) {
            if (use_cap) y = p.softcap * fast_tanh(y * rcap);
          }
          if constexpr (F_ALIBI) {
            if (use_alibi) {
              const int key = kv0 + 32 * kb + acc_row(r, hi);
              y -= slope * fabsf((float)(my_q + shift - key));
            }
          }
          s[kb][r] = y;
        }
    }
    const bool need_mask = (kv0 + BN - 1 > w_full_hi) || (kv0 < w_full_lo);
    if (need_mask) {  // reference mask.h:172-203 predicate, evaluated on accumulator coordinates
      const int rel_hi = lim_hi - kv0 - 4 * hi;
      const int rel_lo = lim_lo - kv0 - 4 * hi;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int off = 32 * kb + acc_row(r, 0);
          const bool vis = (off <= rel_hi) && (off >= rel_lo);
          s[kb][r] = vis ? s[kb][r] : -INFINITY;
        }
    }
    // row max: in-lane over 32 keys, then the other half-wave's 32 keys
    float tmax = s[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[1][r]);
    tmax = half_max(tmax);

    // Running max moves only when it grew by more than thr (log2 units): P <= 2^thr stays exactly
    // representable relative to the row sum; thr = 0 is the reference's update rule (softmax.h:136-167).
    const float m_new = fmaxf(m_run, tmax);
    const bool grow = (m_new - m_run) * cs > thr;  // first visible key: -inf -> finite is always "grow"
    if (__any(grow)) {
      const float m_upd = grow ? m_new : m_run;
      const float m_safe = (m_upd == -INFINITY) ? 0.f : m_upd;
      const float alpha = grow ? fast_exp2((m_run - m_safe) * cs) : 1.f;
      m_run = m_upd;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[db][r] *= alpha;
    }
    const float neg_mc = (m_run == -INFINITY) ? 0.f : -m_run * cs;  // fully masked so far (softmax.h:76,154-156)
    float psum0 = 0.f, psum1 = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = fast_exp2(__builtin_fmaf(s[kb][r], cs, neg_mc));
        const float p1 = fast_exp2(__builtin_fmaf(s[kb][r + 1], cs, neg_mc));
        s[kb][r] = p0;
        s[kb][r + 1] = p1;
        psum0 += p0;
        psum1 += p1;
      }
    l_run += psum0 + psum1;
    if constexpr (F_DROP) {
      if (drop) {  // the row sum above is that of the un-dropped probabilities (flash_fwd_kernel.h:357-368)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int key0 = kv0 + 32 * kb + 8 * g4 + 4 * hi;  // acc rows 4*g4 .. 4*g4+3 are keys key0 .. key0+3
            const uint32_t bytes = drop_bytes(drop_key, my_row, key0 >> 2);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const uint32_t byte = (bytes >> (8 * c)) & 0xffu;
              if (byte > p.drop_thr8) s[kb][4 * g4 + c] = 0.f;
              if (rv_row && row_valid && key0 + c < sk) rv_row[key0 + c] = (uint8_t)byte;
            }
          }
      }
    }
    // P^T as B operand: k-step (kb,t) <-> accumulator registers 8t..8t+7 of s[kb]
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) pf[kb * 2 + t][jj] = (E)s[kb][8 * t + jj];
  };

  // O^T[d][query] += V^T[d][key] . P^T[key][query]; transpose reads run PFV MFMAs ahead
  auto pv = [&](auto bufc) __attribute__((always_inline)) {
    constexpr int buf = decltype(bufc)::value;
    const char FA_LDS* vbuf = lds + (2 + buf) * TILE_BYTES;
    constexpr int NOP = 4 * DB, PFV = 4;
    s16x4 vlo[PFV], vhi[PFV];
#pragma unroll
    for (int i = 0; i < PFV - 1 && i < NOP; ++i) {
      vlo[i % PFV] = lds_read_tr16(vbuf + vaddr[i % DB] + (16 * (i / DB)) * ROW_BYTES);
      vhi[i % PFV] = lds_read_tr16(vbuf + vaddr[i % DB] + (16 * (i / DB) + 8) * ROW_BYTES);
    }
#pragma unroll
    for (int i = 0; i < NOP; ++i) {
      const int nx = i + PFV - 1;
      if (nx < NOP) {
        vlo[nx % PFV] = lds_read_tr16(vbuf + vaddr[nx % DB] + (16 * (nx / DB)) * ROW_BYTES);
        vhi[nx % PFV] = lds_read_tr16(vbuf + vaddr[nx % DB] + (16 * (nx / DB) + 8) * ROW_BYTES);
      }
      __builtin_amdgcn_sched_barrier(0);
      o_acc[i % DB] = T::mfma(combine_tr<V8>(vlo[i % PFV], vhi[i % PFV]), pf[i / DB], o_acc[i % DB]);
    }
  };

  if constexpr (!PP) {
    // ------------------------------ lock-step schedule ------------------------------
    if (q_staged) {  // Q block: each wave DMAs its own 32 rows into the idle LDS (K-style swizzle) and copies its fragments to registers
      constexpr int QDPW = 32 * ROW_BYTES / 1024;
#pragma unroll
      for (int i = 0; i < QDPW; ++i) {
        const int row = wave * 32 + i * RPD + lane / CPR, pc = lane % CPR;
        const int grow = min(m0 + row, sq - 1);
        int c = pc ^ k_swz<D>(row);
        if (DV < D) c = c < CV ? c : 0;
        const E* src = qp + (int64_t)grow * p.q_rs + c * 8;
        if (bounded) src = c < cvr ? src : zsrc;
        lds_dma_16B(src, lds + (wave * QDPW + i) * 1024);
      }
      lds_dma_wait_all();
      const int qb = (wave * 32 + qi) * ROW_BYTES;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        qf[ks] = bitcast_u32x4<V8>(*(const u32x4 FA_LDS*)(lds + qb + (((2 * ks + hi) ^ k_swz<D>(qi)) << 4)));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();  // the K/V tile DMA below reuses this LDS
    }
    if (n_tiles > 0) {
      dma_tile(IC<0>{}, 0, n_min);
      dma_tile(IC<1>{}, 0, n_min);
      lds_dma_wait_all();
      __syncthreads();
    }
    auto step = [&](auto bufc, int j) __attribute__((always_inline)) {
      constexpr int buf = decltype(bufc)::value;
      if (j + 1 < n_tiles) {  // lands in the other buffers while this tile is being computed
        dma_tile(IC<0>{}, buf ^ 1, n_min + j + 1);
        dma_tile(IC<1>{}, buf ^ 1, n_min + j + 1);
      }
      if (tile_active(j)) {
        qk(bufc);
        softmax_step(j);
        pv(bufc);
      }
      lds_dma_wait_all();  // this wave's pieces have landed ...
      __syncthreads();     // ... and everybody's are visible before the next tile reads them
    };
    for (int j = 0; j < n_tiles; j += 2) {
      step(IC<0>{}, j);
      if (j + 1 < n_tiles) step(IC<1>{}, j + 1);
    }
  } else {
    // ------------------------------ ping-pong schedule ------------------------------
    // Early half (waves 0-3):  B0 [QK0] B1 [SM0] B2 [MF0] B3 [SM1] B4 [MF1] ... ; late half the same, one
    // barrier later.  MF(j) reads K_{j+1} (kbuf[(j+1)&1]) and V_j (vbuf[j&1]) and stores K_{j+2} over K_j and
    // V_{j+1} over V_{j-1}: the early half's stores start two barriers after the late half's last read of the
    // old contents, and the first reader of the new contents (MF(j+1), early half) starts after the late
    // half's stores have been fenced by a barrier.
    const bool late = wave >= 4;
    if (n_tiles > 0) {
      load_k(n_min);
      load_v(n_min);
      store_k(IC<0>{});
      store_v(IC<0>{});
      if (n_tiles > 1) {
        load_k(n_min + 1);
        store_k(IC<1>{});
      }
    }
    __syncthreads();
    if (late) __builtin_amdgcn_s_barrier();  // run one interval behind waves 0-3
    if (tile_active(0)) qk(IC<0>{});
    __syncthreads();
    auto step = [&](auto bufc, int j) __attribute__((always_inline)) {  // buf = j & 1
      constexpr int buf = decltype(bufc)::value;
      // SM(j): VALU interval
      const bool pre_k = (j + 2 < n_tiles), pre_v = (j + 1 < n_tiles);
      if (pre_k) load_k(n_min + j + 2);
      if (pre_v) load_v(n_min + j + 1);
      const bool act = tile_active(j);
      if (act) softmax_step(j);
      __syncthreads();
      // MF(j): matrix interval
      if (tile_active(j + 1)) qk(IC<buf ^ 1>{});
      if (act) pv(bufc);
      if (pre_k) store_k(bufc);           // K_{j+2} replaces K_j
      if (pre_v) store_v(IC<buf ^ 1>{});  // V_{j+1} replaces V_{j-1}
      __syncthreads();
    };
    for (int j = 0; j < n_tiles; j += 2) {
      step(IC<0>{}, j);
      if (j + 1 < n_tiles) step(IC<1>{}, j + 1);
    }
    if (!late) __builtin_amdgcn_s_barrier();  // keep barrier counts equal across the workgroup
  }

  // ---- epilogue: normalise, store O (bf16/fp16) and LSE -------------------------------------------
  if (!wave_valid) return;
  const float l_tot = half_sum(l_run);
  const bool dead = (l_tot == 0.f) || (l_tot != l_tot);  // no visible key (softmax.h:179-180)
  const float inv = (dead ? 1.f : 1.f / l_tot) * (drop ? p.rp_keep : 1.f);
  if (p.n_splits > 1) {  // partial result of this key split, fp32, merged by fa_splitkv_combine_kernel
    if (row_valid) {
      const int64_t prow = (((int64_t)split * p.b + b) * p.h + h) * p.sq + my_row;
      float* orow = p.o_accum + prow * D;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 ov;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) ov[jj] = o_acc[db][4 * g + jj] * inv;
          *reinterpret_cast<f32x4*>(orow + 32 * db + 8 * g + 4 * hi) = ov;
        }
      if (hi == 0) p.lse_accum[prow] = dead ? -INFINITY : (m_run * cs * kLn2 + __logf(l_tot));
    }
    return;
  }
  // O tile through the freed K/V buffers: whole-row stores (fa_device.h store_tile_via_lds)
  if (packed) {
    store_tile_via_lds_packed<E, D, DV>(lds + wave * 32 * (ROW_BYTES + 16), o_acc, inv, op, p.o_rs, p.o_hs, g, w_row0, sq, lane, cvr);
    if (row_valid && hi == 0) lsep[(int64_t)my_hh * sq_true + my_q] = dead ? INFINITY : (m_run * cs * kLn2 + __logf(l_tot));
    return;
  }
  store_tile_via_lds<E, D, DV>(lds + wave * 32 * (ROW_BYTES + 16), o_acc, inv, op + (int64_t)w_row0 * p.o_rs, p.o_rs, sq - w_row0, lane, cvr);
  if (row_valid && hi == 0) lsep[my_row] = dead ? INFINITY : (m_run * cs * kLn2 + __logf(l_tot));
}

// Merge of the split-KV partials (reference combine_attn_seqk_parallel, flash_fwd_kernel.h:1117-1299): one wave per
// (batch, head, query row); lse = log sum_s exp(lse_s), out = sum_s exp(lse_s - lse) * out_s.
template <typename E, int D, int DV>
__global__ void __launch_bounds__(256) fa_splitkv_combine_kernel(const FwdK p) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t rows = (int64_t)p.b * p.h * p.sq;
  if (r >= rows) return;
  const int i = (int)(r % p.sq);
  const int h = (int)((r / p.sq) % p.h);
  const int b = (int)(r / ((int64_t)p.sq * p.h));
  const float lse_s = lane < p.n_splits ? p.lse_accum[(int64_t)lane * rows + r] : -INFINITY;
  float mx = lse_s;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  const bool dead = (mx == -INFINITY);
  float e = dead ? 0.f : __expf(lse_s - mx);
  float sum = e;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
  const float wgt = dead ? 0.f : e / sum;  // exp(lse_s - lse)
  constexpr int EPL = D / 64;              // output elements per lane (partial rows have pitch D; columns >= DV do not exist)
  const bool col_ok = lane * EPL < (p.d_chunks > 0 ? 8 * p.d_chunks : DV);
  float acc[EPL];
#pragma unroll
  for (int t = 0; t < EPL; ++t) acc[t] = 0.f;
  // 8 partials in flight per round (independent loads, no branch: an empty partial has weight 0 and holds zeros)
  for (int s0 = 0; s0 < p.n_splits; s0 += 8) {
    float part[8][EPL], ws[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int s = min(s0 + j, p.n_splits - 1);
      ws[j] = (s0 + j < p.n_splits) ? __shfl(wgt, s) : 0.f;
      const float* src = p.o_accum + ((int64_t)s * rows + r) * D + lane * EPL;
#pragma unroll
      for (int t = 0; t < EPL; ++t) part[j][t] = col_ok ? src[t] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int t = 0; t < EPL; ++t) acc[t] += ws[j] * part[j][t];
  }
  const int g = p.pack_g;                  // packed heads (fa_fwd_kernel): row i of "head" h is query i / g of head h * g + i % g
  const int iq = g > 1 ? i / g : i, hh = i - iq * g;
  E* dst = (E*)p.o + (int64_t)b * p.o_bs + (int64_t)iq * p.o_rs + ((int64_t)h * g + hh) * p.o_hs + lane * EPL;
#pragma unroll
  for (int t = 0; t < EPL; ++t)
    if (col_ok) dst[t] = (E)acc[t];
  if (lane == 0) p.lse[((int64_t)b * p.h + h) * p.sq + (int64_t)hh * (p.sq / g) + iq] = dead ? INFINITY : (mx + __logf(sum));
}

#if FA_FWD_PART != 2
template <typename E>
static int launch_combine_e(const FwdK& p, int d, dim3 grid, dim3 block, hipStream_t stream) {
  switch (d) {
    case 256: hipLaunchKernelGGL((fa_splitkv_combine_kernel<E, 256, 256>), grid, block, 0, stream, p); break;
    case 192: hipLaunchKernelGGL((fa_splitkv_combine_kernel<E, 256, 192>), grid, block, 0, stream, p); break;
    case 128: hipLaunchKernelGGL((fa_splitkv_combine_kernel<E, 128, 128>), grid, block, 0, stream, p); break;
    case 96: hipLaunchKernelGGL((fa_splitkv_combine_kernel<E, 128, 96>), grid, block, 0, stream, p); break;
    case 64: hipLaunchKernelGGL((fa_splitkv_combine_kernel<E, 64, 64>), grid, block, 0, stream, p); break;
    case 32: hipLaunchKernelGGL((fa_splitkv_combine_kernel<E, 64, 32>), grid, block, 0, stream, p); break;
    default: return -2;
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int launch_splitkv_combine(const FwdK& p, int dtype_bf16, int d, hipStream_t stream) {
  const int64_t rows = (int64_t)p.b * p.h * p.sq;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  return dtype_bf16 ? launch_combine_e<__bf16>(p, d, grid, block, stream) : launch_combine_e<_Float16>(p, d, grid, block, stream);
}

// Rotary embedding (reference csrc/flash_attn/src/rotary.h, flash_fwd_kernel.h:640-720): one thread rotates 8 channel
// pairs of one (batch, row, head) = two 16-B chunks: interleaved pairs (2t, 2t+1) -> two adjacent chunks; non-interleaved
// pairs (t, t + rotary_dim/2) -> chunk c and chunk c + rotary_dim/16.
// Arithmetic in fp32, one rounding to the storage type; channels >= rotary_dim are copied.
template <typename E>
__global__ void __launch_bounds__(256) fa_rotary_kernel(const RotaryK p) {
  using V8 = typename ElemTraits<E>::v8;
  const int cpr = p.d / 8;                 // 16-B chunks per row
  const int rc = p.rotary_dim / 8;         // chunks in the rotated part (even)
  const int upr = rc / 2 + (cpr - rc);     // work units per row: chunk pairs + pass-through chunks
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)p.b * p.s * p.h * upr;
  if (idx >= total) return;
  const int u = (int)(idx % upr);
  const int h = (int)((idx / upr) % p.h);
  const int srow = (int)((idx / ((int64_t)upr * p.h)) % p.s);
  const int b = (int)(idx / ((int64_t)upr * p.h * p.s));
  const E* xr = (const E*)p.x + (int64_t)b * p.x_bs + (int64_t)srow * p.x_rs + (int64_t)h * p.x_hs;
  E* yr = (E*)p.y + (int64_t)b * p.y_bs + (int64_t)srow * p.y_rs + (int64_t)h * p.y_hs;
  if (u >= rc / 2) {  // channels past rotary_dim
    const int c = rc + (u - rc / 2);
    *reinterpret_cast<u32x4*>(yr + 8 * c) = *reinterpret_cast<const u32x4*>(xr + 8 * c);
    return;
  }
  int pos = (p.offsets ? p.offsets[b] : 0) + (p.per_token ? srow : 0);
  pos = min(pos, p.seqlen_ro - 1);
  const E* cr = (const E*)p.cos + (int64_t)pos * p.cos_rs;
  const E* sr = (const E*)p.sin + (int64_t)pos * p.cos_rs;
  const int c0 = p.interleaved ? 2 * u : u;           // first chunk of the pair
  const int c1 = p.interleaved ? 2 * u + 1 : u + rc / 2;
  const V8 a = bitcast_u32x4<V8>(*reinterpret_cast<const u32x4*>(xr + 8 * c0));
  const V8 bb = bitcast_u32x4<V8>(*reinterpret_cast<const u32x4*>(xr + 8 * c1));
  const V8 cv = bitcast_u32x4<V8>(*reinterpret_cast<const u32x4*>(cr + 8 * u));  // angles 8u .. 8u+7
  const V8 sv = bitcast_u32x4<V8>(*reinterpret_cast<const u32x4*>(sr + 8 * u));
  V8 oa, ob;
  if (p.interleaved) {  // channel 16u + 2t pairs with 16u + 2t + 1, angle 8u + t
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float x1 = (float)(t < 4 ? a[2 * t] : bb[2 * t - 8]), x2 = (float)(t < 4 ? a[2 * t + 1] : bb[2 * t - 7]);
      const float c = (float)cv[t], sn = (float)sv[t];
      const E o1 = (E)(x1 * c - x2 * sn), o2 = (E)(x1 * sn + x2 * c);
      if (t < 4) { oa[2 * t] = o1; oa[2 * t + 1] = o2; } else { ob[2 * t - 8] = o1; ob[2 * t - 7] = o2; }
    }
  } else {  // channel 8u + t pairs with 8u + t + rotary_dim/2, angle 8u + t
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float x1 = (float)a[t], x2 = (float)bb[t], c = (float)cv[t], sn = (float)sv[t];
      oa[t] = (E)(x1 * c - x2 * sn);
      ob[t] = (E)(x1 * sn + x2 * c);
    }
  }
  *reinterpret_cast<V8*>(yr + 8 * c0) = oa;
  *reinterpret_cast<V8*>(yr + 8 * c1) = ob;
}
int launch_rotary(const RotaryK& p, int dtype_bf16, hipStream_t stream) {
  const int cpr = p.d / 8, rc = p.rotary_dim / 8;
  const int64_t total = (int64_t)p.b * p.s * p.h * (rc / 2 + (cpr - rc));
  if (total <= 0) return 0;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype_bf16) hipLaunchKernelGGL(fa_rotary_kernel<__bf16>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(fa_rotary_kernel<_Float16>, grid, block, 0, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Varlen work list (FA3 has a scheduler pre-pass for the same reason, hopper/flash_prepare_scheduler.cu): one workgroup
// enumerates the non-empty blocks of the blocked side (query blocks for the forward / dQ, key blocks for dK/dV), estimates
// each block's work from the mask geometry and writes them heaviest first by a counting sort over 1024 work buckets
// (bucket width = 2^work_shift rows of the other side, chosen on the host so that the longest sequence spans the buckets).
// The order inside a bucket depends on atomics, the results of the attention kernels do not.
// Sequences of up to 8 blocks are handled one per thread (thousands of short sequences in parallel); longer ones are queued
// and walked by the whole workgroup, 1024 blocks at a time (a single 128k-token sequence is 1024 blocks).
__global__ void __launch_bounds__(1024) fa_varlen_schedule_kernel(const SchedK p) {
  __shared__ int hist[1024];
  __shared__ int long_seq[1024];
  __shared__ int total, n_long;
  const int t = threadIdx.x;
  hist[t] = 0;
  if (t == 0) { total = 0; n_long = 0; }
  __syncthreads();
  auto work_key = [&](int len_a, int len_o, int blkidx) {
    const int a0 = blkidx * p.blk, a1 = min(a0 + p.blk, len_a) - 1;
    int lo, hi;
    if (p.keys_blocked == 0) {  // a = queries, o = keys: visible keys of rows a0..a1 (mask.h:172-203 geometry)
      const int shift = len_o - len_a;
      hi = len_o - 1; lo = 0;
      if (p.wr >= 0) hi = min(hi, a1 + shift + p.wr);
      if (p.wl >= 0) lo = max(0, a0 + shift - p.wl);
    } else {                    // a = keys, o = queries: queries that see keys a0..a1
      const int shift = len_a - len_o;
      hi = len_o - 1; lo = 0;
      if (p.wr >= 0) lo = max(0, a0 - shift - p.wr);
      if (p.wl >= 0) hi = min(hi, a1 - shift + p.wl);
    }
    const int w = max(0, hi - lo + 1);
    return 1023 - min(1023, (w + (1 << p.work_shift) - 1) >> p.work_shift);  // bucket 0 = heaviest
  };
  auto seq_lens = [&](int b, int& len_a, int& len_o) {
    len_a = p.cu_a[b + 1] - p.cu_a[b];
    len_o = p.cu_o[b + 1] - p.cu_o[b];
    if (p.seqused_o) len_o = min(len_o, p.seqused_o[b]);
  };
  constexpr int SHORT = 8;
  for (int pass = 0; pass < 2; ++pass) {
    auto visit = [&](int b, int len_a, int len_o, int m) {
      const int key = work_key(len_a, len_o, m);
      const int pos = atomicAdd(&hist[key], 1);   // pass 0: count; pass 1: hist holds the bucket cursors
      if (pass == 1 && pos < p.bound) p.list[1 + pos] = make_int2(b, m);
    };
    for (int b = t; b < p.nb; b += 1024) {
      int len_a, len_o;
      seq_lens(b, len_a, len_o);
      const int nblk = (len_a + p.blk - 1) / p.blk;
      if (nblk <= SHORT) {
        for (int m = 0; m < nblk; ++m) visit(b, len_a, len_o, m);
      } else if (pass == 0) {
        const int slot = atomicAdd(&n_long, 1);
        if (slot < 1024) long_seq[slot] = b;
        else for (int m = 0; m < nblk; ++m) visit(b, len_a, len_o, m);   // (more than 1024 long sequences: this thread walks it)
      } else if (p.nb > 1024) {  // pass 1: was this sequence queued?  Only an overflowing queue leaves unqueued long sequences.
        bool queued = false;
        const int nl = min(n_long, 1024);
        for (int i = 0; i < nl && !queued; ++i) queued = long_seq[i] == b;
        if (!queued) for (int m = 0; m < nblk; ++m) visit(b, len_a, len_o, m);
      }
      if (pass == 0) atomicAdd(&total, nblk);
    }
    __syncthreads();
    const int nl = min(n_long, 1024);
    for (int i = 0; i < nl; ++i) {
      const int b = long_seq[i];
      int len_a, len_o;
      seq_lens(b, len_a, len_o);
      const int nblk = (len_a + p.blk - 1) / p.blk;
      for (int m = t; m < nblk; m += 1024) visit(b, len_a, len_o, m);
    }
    __syncthreads();
    if (pass == 0) {
      if (t == 0) {  // exclusive scan, heaviest bucket first
        int run = 0;
        for (int k = 0; k < 1024; ++k) { const int c = hist[k]; hist[k] = run; run += c; }
        p.list[0] = make_int2(min(total, p.bound), 0);
      }
      __syncthreads();
    }
  }
}
int launch_varlen_schedule(const SchedK& p, hipStream_t stream) {
  hipLaunchKernelGGL(fa_varlen_schedule_kernel, dim3(1), dim3(1024), 0, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

__global__ void fa_set_rng_kernel(uint64_t seed, uint64_t offset, uint64_t* dst) {
  dst[0] = seed;
  dst[1] = offset;
}
int launch_set_rng(uint64_t seed, uint64_t offset, uint64_t* dst, hipStream_t stream) {
  hipLaunchKernelGGL(fa_set_rng_kernel, dim3(1), dim3(1), 0, stream, seed, offset, dst);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// KV-cache append (reference flash_fwd_kernel.h:640-720, the Append_KV branch, without rotary): one thread moves
// 16 bytes of one new key row and the matching 16 bytes of the value row.
__global__ void __launch_bounds__(256) fa_kv_append_kernel(const KvAppendK p) {
  const int cpr = p.d / 8;                       // 16-B chunks per row
  const int per_b = p.s_new * p.h_k * cpr;
  const int b = blockIdx.y;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < per_b; i += gridDim.x * 256) {
    const int c = i % cpr, hk = (i / cpr) % p.h_k, t = i / (cpr * p.h_k);
    const int row = (p.seqlens_k ? p.seqlens_k[b] : 0) + t;
    int64_t koff, voff;
    if (p.block_table) {
      const int page = row / p.page_size;
      const int blk = p.block_table[(int64_t)b * p.block_table_bs + page];
      koff = (int64_t)blk * p.kc_bs + (int64_t)(row - page * p.page_size) * p.kc_rs;
      voff = (int64_t)blk * p.vc_bs + (int64_t)(row - page * p.page_size) * p.vc_rs;
    } else {
      const int bkv = p.kv_batch_idx ? p.kv_batch_idx[b] : b;
      koff = (int64_t)bkv * p.kc_bs + (int64_t)row * p.kc_rs;
      voff = (int64_t)bkv * p.vc_bs + (int64_t)row * p.vc_rs;
    }
    const unsigned short* ks = (const unsigned short*)p.knew + (int64_t)b * p.kn_bs + (int64_t)t * p.kn_rs + (int64_t)hk * p.kn_hs + c * 8;
    const unsigned short* vs = (const unsigned short*)p.vnew + (int64_t)b * p.vn_bs + (int64_t)t * p.vn_rs + (int64_t)hk * p.vn_hs + c * 8;
    unsigned short* kd = (unsigned short*)p.kcache + koff + (int64_t)hk * p.kc_hs + c * 8;
    unsigned short* vd = (unsigned short*)p.vcache + voff + (int64_t)hk * p.vc_hs + c * 8;
    *reinterpret_cast<u32x4*>(kd) = *reinterpret_cast<const u32x4*>(ks);
    *reinterpret_cast<u32x4*>(vd) = *reinterpret_cast<const u32x4*>(vs);
  }
}

int launch_kv_append(const KvAppendK& p, hipStream_t stream) {
  if (p.b <= 0 || p.s_new <= 0) return 0;
  const int per_b = p.s_new * p.h_k * (p.d / 8);
  dim3 grid((unsigned)std::min(256, (per_b + 255) / 256), (unsigned)p.b);
  hipLaunchKernelGGL(fa_kv_append_kernel, grid, dim3(256), 0, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

#endif  // FA_FWD_PART != 2

template <typename E, int D, int DV, int NW, int FEAT, bool PP>
static int launch_fwd_t(const FwdK& p, hipStream_t stream) {
  constexpr int smem = 4 * 64 * D * 2 + NW * 32 * 16;  // K/V double buffers (+ the row padding of the staged O epilogue)
  auto kern = fa_fwd_kernel<E, D, DV, NW, FEAT, PP>;
  static std::atomic<unsigned long long> attr_mask{0};
  if (ensure_dyn_lds(attr_mask, (const void*)kern, smem) != 0) return -1;
  const long long total = units_grid(p.n_units, p.unit_size);
  if (total <= 0) return 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(NW * 64), smem, stream, p);
  if (hipGetLastError() != hipSuccess) return -1;
  LastSchedule& ls = last_schedule();
  ls.fwd_kernel = 1; ls.fwd_nw = PP ? 16 : NW; ls.fwd_feat = FEAT; ls.fwd_splits = p.n_splits; ls.fwd_list = p.work_list != nullptr; ls.d = DV;
  ls.bf16 = std::is_same<E, __bf16>::value;
  snprintf(ls.name, sizeof(ls.name), "fa::fa_fwd_kernel<%s,%d,%d,feat%d,%s>", ls.bf16 ? "bf16" : "f16", DV, NW, FEAT, PP ? "pingpong" : "lockstep");
  return 0;
}

#if FA_FWD_PART != 2
int fwd_block_m(int nw) { return nw == 16 ? 256 : 32 * nw; }
#endif

template <typename E, int D, int FEAT>
static int launch_fwd_f(const FwdK& p, int nw, hipStream_t stream) {
  if constexpr (D > 128) {
    (void)nw;
    return launch_fwd_t<E, D, D, 4, FEAT, false>(p, stream);
  } else {
    if (nw == 8) return launch_fwd_t<E, D, D, 8, FEAT, false>(p, stream);
    if (nw == 4) return launch_fwd_t<E, D, D, 4, FEAT, false>(p, stream);
    if constexpr (FEAT == FEAT_NONE || FEAT == FEAT_ALL) {  // the ping-pong schedule is built plain and all-features only
      if (nw == 16) return launch_fwd_t<E, D, D, 8, FEAT, true>(p, stream);
    }
    return -2;
  }
}
template <typename E, int D>
static int launch_fwd_ed(const FwdK& p, int nw, hipStream_t stream) {
  int feat = feat_code(p.softcap > 0.f, p.alibi != nullptr, p.rng != nullptr);
  if (nw == 16 && feat != FEAT_NONE) feat = FEAT_ALL;
  switch (feat) {
    case FEAT_NONE: return launch_fwd_f<E, D, FEAT_NONE>(p, nw, stream);
    case FEAT_CAP: return launch_fwd_f<E, D, FEAT_CAP>(p, nw, stream);
    case FEAT_ALIBI: return launch_fwd_f<E, D, FEAT_ALIBI>(p, nw, stream);
    case FEAT_DROP: return launch_fwd_f<E, D, FEAT_DROP>(p, nw, stream);
    case FEAT_CAP | FEAT_DROP: return launch_fwd_f<E, D, (FEAT_CAP | FEAT_DROP)>(p, nw, stream);
    case FEAT_ALIBI | FEAT_DROP: return launch_fwd_f<E, D, (FEAT_ALIBI | FEAT_DROP)>(p, nw, stream);
    default: return launch_fwd_f<E, D, FEAT_ALL>(p, nw, stream);
  }
}
// trimmed head dims (32 / 96 / 192): 4-wave lock-step schedule; plain, or the run-time-checked all-features variant
template <typename E, int D, int DV>
static int launch_fwd_trim(const FwdK& p, hipStream_t stream) {
  if (p.softcap > 0.f || p.alibi != nullptr || p.rng != nullptr) return launch_fwd_t<E, D, DV, 4, FEAT_ALL, false>(p, stream);
  return launch_fwd_t<E, D, DV, 4, FEAT_NONE, false>(p, stream);
}
template <typename E>
static int launch_fwd_e(const FwdK& p, int d, int nw, hipStream_t stream) {
  switch (d) {
    case 128: return launch_fwd_ed<E, 128>(p, nw, stream);
    case 64: return launch_fwd_ed<E, 64>(p, nw, stream);
    case 256: return launch_fwd_ed<E, 256>(p, nw, stream);
    case 96: return nw == 4 ? launch_fwd_trim<E, 128, 96>(p, stream) : -2;
    case 32: return nw == 4 ? launch_fwd_trim<E, 64, 32>(p, stream) : -2;
    case 192: return nw == 4 ? launch_fwd_trim<E, 256, 192>(p, stream) : -2;
    default: return -2;
  }
}

#if FA_FWD_PART != 1
int launch_fwd_f16(const FwdK& p, int d, int nw, hipStream_t stream) { return launch_fwd_e<_Float16>(p, d, nw, stream); }
#endif
#if FA_FWD_PART != 2
int launch_fwd_bf16(const FwdK& p, int d, int nw, hipStream_t stream) { return launch_fwd_e<__bf16>(p, d, nw, stream); }
// nw: 4 / 8 = lock-step schedule with 4 / 8 waves per workgroup, 16 = 8-wave ping-pong schedule
int launch_fwd(const FwdK& p, int dtype_bf16, int d, int nw, hipStream_t stream) {
  // the scalar-base + 32-bit lane-offset tile loads need one tile's extent to fit 32 bits
  if ((uint64_t)64 * (uint64_t)(p.k_rs > p.v_rs ? p.k_rs : p.v_rs) * 2u >= (1ull << 31)) return -3;
  return dtype_bf16 ? launch_fwd_bf16(p, d, nw, stream) : launch_fwd_f16(p, d, nw, stream);
}
#endif

}  // namespace fa
