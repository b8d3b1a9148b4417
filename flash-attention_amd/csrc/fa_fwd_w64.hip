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
//
// Round 3: ONE code path, and fewer instructions.  With one wave per SIMD nothing overlaps a wave's own instruction stream: every instruction
// of the tile loop costs ~4.7 clocks whatever it is (measured in both directions, profiles/r03_fwd_w64_ablations.txt), and the once-per-block
// code runs at ~5 clocks per instruction.  Round 2's kernel had a lot of both (241 KB: four copies of the hand-placed iteration,
// compiler-ordered "generic" steps for the pipeline fill / drain, masked step variants, twelve copies of the cold rescale code; per block
// 23k clocks of prologue, 10-17k for iteration 0, 5.3k instead of 3.2k for every iteration with a masked or draining wave).  Now (53 KB):
//   * one step body (two hand-placed steps = an iteration; the loop holds an even and an odd iteration so that the K/V buffer parity is
//     a compile-time constant of the LDS reads' immediate offsets);
//   * the same body fills and drains the pipeline: chains that score no real key read a zero-filled tile (a buffer descriptor of zero
//     records "loads" zeros into LDS without touching memory; so do the rows of a partial last tile past the last key) and are masked;
//   * masks ride in the score chain's C operand: set_mask() puts -inf into the -m broadcast of the masked (row, key) pairs before a step
//     that straddles a mask boundary (two instructions per element from a per-lane bitmap, straight-line), the step itself -- its row-max
//     tree, exp2 and decision -- runs unchanged; the cold exit of a step is the rescale alone;
//   * an iteration's scalar preparation is eleven instructions (tile offsets as two multiplies, constant descriptors, an arithmetic mask
//     flag), LDS waits are paired (one explicit s_waitcnt per two fragment slots, which hipcc models and then omits its own);
//   * Q / O through buffer descriptors with 32-bit lane offsets (hoisted 64-bit addresses were spilled: a scratch reload waits on vmcnt(0),
//     i.e. on every tile DMA in flight), the next block's Q trickled in a piece per iteration, block id carried from that prefetch; O zeroed
//     by eight MFMAs, accumulator reads eight per asm statement (hipcc pads every asm statement with an s_nop).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "fa_device.h"
#include "fa_kernel_params.h"
#include "fa_launch.h"
#include "fa_fwd_w64_regs.h"
#define FA_W64_CLOB FA_W64_ACC_CLOBBERS
#include "fa_w64_asm.h"


// Schedule constants of the hand-placed step (each one the winner of an A/B recorded under profiles/; the timing ablations and the losing variants are
// applied as experiments/ablations/fa_fwd_w64.patch by tools/ablate_w64.sh -- the product source carries none of them):
#define FA_W64_AH 3        // LDS operand reads run this many fragment slots (2 gaps each) ahead of their MFMAs, one explicit wait per two slots
#define FA_W64_KDMA_G0 0   // gaps of a step that carry its LDS-DMA pieces (first gap, stride): where a piece is issued prices it between ~30 and ~180 clocks
#define FA_W64_KDMA_GS 2   // (MI355X_MICROARCH.md "LDS-DMA piece issue cost").  Round 6 A/B, profiles/r06_fwd_w64_dma_placement.txt: the K pieces in the EVEN gaps 0, 2, 4, 6 --
#define FA_W64_VDMA_G0 1   // the ones that make one probability, not two -- +1 % without a mask (1278 -> 1290 at S = 16k, 1225 -> 1240 at S = 4k), a tie under the causal mask; the
#define FA_W64_VDMA_GS 2   // P.V half's late gaps (17.., 24..) and every other V placement tried are ties or losses (the V pieces then land too close to the tile barrier)

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
// FEAT: 0 = plain, FEAT_ALIBI = causal ALiBi (right bound exactly on the diagonal: every visible key j of query i has j <= i + shift, so the bias
// -slope * |i + shift - j| is linear in j).  The bias rides in the same C operand as -m: C_r = -m + slope2 * (key_r - i - shift), moved from step
// to step by 32 in-place adds in the gaps behind the chain starts -- the only cost of the feature in the steady state.  The variant walks the key
// tiles DOWNWARDS from the diagonal (DESC), as the reference does (flash_fwd_kernel.h: n_block = n_max-1 .. n_min): walking upwards the biased
// scores grow by slope*32 per step, the running maximum moves at every step and the 540-instruction rescale runs every step (measured: 527
// TFLOP/s at config 3 against 614 for the lock-step kernel, profiles/r03_fwd_schedules.txt); downwards the maximum is found in the first tiles.
// set_mask / clear_mask rewrite the broadcast exactly and the rescale moves it, so the incremental adds' rounding does not accumulate past them.
//
// FEAT_CAP = softcap (round 5; reference: flash_fwd_kernel.h:357-368 + utils.h:395-409, scores -> softcap * tanh(scores / softcap)).  The map is not linear, so the
// maximum cannot be subtracted by the matrix pipe: Q carries scale/softcap * 2*log2e, the chains start from C = 0 (-inf where masked, as ever) and deliver
// y = 2*log2e * z, z = score*scale/softcap; with c = softcap*log2e the probability is 2^(c*tanh(z) - m) = 2^(off - 2c/(2^y + 1)), off = c - m one value per row.
// Seven vector instructions per score instead of two (exp2, add, rcp, two fused multiply-adds -- one of them only carries a masked score's -inf through tanh's
// saturation: -inf * 2^-100 --, exp2, row sum), staged over three gaps so that no instruction waits for the one before it; the row-max tree runs on y itself (tanh is
// monotone) and the decision maps the row maximum once per row and step.  Elements are spread 20 : 12 over the two halves of the step instead of 24 : 8 (the
// score half has the K reads and the DMA, the other half the packing and the tree).
//
// FEAT_DROP = dropout (round 5; reference: flash_fwd_kernel.h:357-368 + dropout.h; the random stream is fa_device.h drop_bytes, the same element -> byte map the
// lock-step kernel and both backward kernels use, so the mask is the same whatever kernel draws it).  Four consecutive keys of a row share one Philox2x32-7 call
// and they are the four accumulator rows 4*g .. 4*g+3 of a lane: eight calls per step and lane, 7 rounds x 3 instructions each (the round keys are per (batch,
// head): scalars), spread two rounds per gap over the step; a finished word waits until its elements are packed for the P.V product, where the dropped ones
// become zeros (byte compare + select per element).  The row sums keep the un-dropped probabilities; 1/(1-p) meets O in the epilogue.  No random-byte output
// (return_softmax): the lock-step kernel serves that.
//
// PAGED = keys and values live in a paged cache (FwdK::block_table; reference: the block_table path of compute_attn_1rowblock_splitkv, flash_fwd_kernel.h:505-1078,
// and of mha_varlen_fwd, flash_api.cpp:538-788).  A page holds a multiple of 256 keys, so a 64-key tile never straddles two: every tile gets its own buffer
// descriptor -- base = pool + table[b][page] * page_stride + the tile's rows inside the page, range = the tile's rows that exist -- made from scalars at the
// head of its iteration (~30 scalar instructions per iteration); the table entry of the tile after next is requested one iteration ahead.  Pools of any size
// (the 32-bit offsets only span a tile).  Plain attention only.
template <typename E, int D, int FEAT = 0, bool PAGED = false>
__global__ void __launch_bounds__(256, 1) fa_fwd_w64_kernel(const FwdK p) {
  constexpr bool ALIBI = FEAT == FEAT_ALIBI;
  static_assert(!PAGED || FEAT == 0, "the paged variant serves plain attention");
  constexpr bool SOFTCAP = FEAT == FEAT_CAP;
  constexpr bool DROP = FEAT == FEAT_DROP;
  constexpr bool DESC = ALIBI;   // iteration u scores key tile n_tiles - 1 - u instead of tile u
  static_assert(FEAT == 0 || FEAT == FEAT_ALIBI || FEAT == FEAT_CAP || FEAT == FEAT_DROP, "feature variants of this schedule: none, causal ALiBi, softcap, dropout");
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
  // decode_id: virtual block -> (batch, head, query block): the divisions, done ONCE per block (the id of the next block is carried
  // over from the Q prefetch); fill_blk: lengths and offsets of a block, cheap
  auto decode_id = [&](int vb, int round, Blk& k) __attribute__((always_inline)) -> bool {
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
    return true;
  };
  auto fill_blk = [&](Blk& k) __attribute__((always_inline)) -> bool {
    k.sq = p.sq; k.sk = p.sk; k.q_row0 = 0; k.k_row0 = 0;
    const int bkv = p.kv_batch_idx ? p.kv_batch_idx[k.b] : k.b;
    k.q_boff = (int64_t)k.b * p.q_bs; k.k_boff = (int64_t)bkv * p.k_bs; k.v_boff = (int64_t)bkv * p.v_bs; k.o_boff = (int64_t)k.b * p.o_bs;
    if (p.cu_q) { const int c0 = p.cu_q[k.b]; k.sq = p.cu_q[k.b + 1] - c0; k.q_row0 = c0; k.q_boff = 0; k.o_boff = 0; }
    if (p.seqused_q) k.sq = min(k.sq, p.seqused_q[k.b]);
    if (p.cu_k) { const int c0 = p.cu_k[k.b]; k.sk = p.cu_k[k.b + 1] - c0; k.k_row0 = c0; k.k_boff = 0; k.v_boff = 0; }
    if (p.seqused_k) k.sk = min(p.seqused_k[k.b] + p.seqused_add, (p.cu_k && !PAGED) ? k.sk : p.sk);   // (inside a packed batch never beyond the entry's slot, as the backward)
    if (p.leftpad_k) {
      const int lp = p.leftpad_k[k.b];
      k.sk = max(0, k.sk - lp);
      k.k_row0 += lp;
    }
    if constexpr (PAGED) { k.k_boff = 0; k.v_boff = 0; k.k_row0 = 0; }   // the page table supplies the rows, cu_seqlens_k / seqused_k only the lengths
    return k.m_block * BM < k.sq;
  };
  // this wave's 64 rows of a block's Q -> its rows of the LDS Q region (coalesced DMA, K-style swizzle on the source chunk)
  // Q block -> LDS Q region (K-style swizzle on the source chunk), 1-KiB pieces through a buffer descriptor of the block's rows: rows
  // past the sequence end are outside it and arrive as zeros.  Pieces are addressed by a RUN-TIME index with 32-bit offsets: with
  // compile-time indices hipcc hoists sixteen 64-bit per-lane addresses out of the persistent loop and spills them.
  constexpr int QDMA = (BM * ROW_BYTES) / 1024 / NW;   // pieces per wave
  auto q_srd_of = [&](const Blk& k) __attribute__((always_inline)) {
    const E* base = (const E*)p.q + k.q_boff + (k.q_row0 + (int64_t)k.m_block * BM) * p.q_rs + (int64_t)k.h * p.q_hs;
    const int rows = min(BM, k.sq - k.m_block * BM);
    const unsigned long long a = (unsigned long long)base;
    const unsigned nrec = rows > 0 ? (unsigned)(((unsigned long long)(rows - 1) * (unsigned long long)p.q_rs + D) * 2ull) : 0u;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi16 = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu;
    const unsigned nr = __builtin_amdgcn_readfirstlane(nrec);
    u32x4 srd = {lo, hi16, nr, 0x00020000u};
    return srd;
  };
  auto dma_q_piece = [&](const u32x4& srd, int i) __attribute__((always_inline)) {   // piece i (0 .. QDMA-1) of this wave's rows
    const int pc = wave * QDMA + i;
    const int row = pc * RPD + d_row;
    const unsigned vo = (unsigned)(row * (int)p.q_rs + ((d_pc ^ k_swz_w<D>(row)) << 3)) * 2u;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(Q_OFF + pc * 1024));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(vo), "s"(dst), "s"(srd) : "memory");
  };
  auto dma_q = [&](const u32x4& srd) __attribute__((always_inline)) {
#pragma unroll 1
    for (int i = 0; i < QDMA; ++i) dma_q_piece(srd, i);
  };
  int q_in_lds = -1;   // virtual block whose Q this wave has prefetched into the LDS Q region

  // ---- per-lane constants of every block: DMA source offsets of this lane's K / V pieces, LDS read bases
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
  const float thr = p.rescale_thr;
  const bool two_sided = p.wl >= 0;

  Blk cur_id;
  cur_id.b = 0; cur_id.h = 0; cur_id.m_block = 0;
  bool cur_ok = decode_id((int)blockIdx.x, 0, cur_id);
  for (int vb = blockIdx.x, round = 0; vb < n_virtual; vb += gridDim.x, ++round) {
  Blk blk = cur_id, nxt;
  const bool blk_ok = cur_ok && fill_blk(blk);
  nxt.b = 0; nxt.h = 0; nxt.m_block = 0;
  const bool nxt_id_ok = p.persist_total > 0 && decode_id(vb + (int)gridDim.x, round + 1, nxt);
  cur_id.b = __builtin_amdgcn_readfirstlane(nxt.b); cur_id.h = __builtin_amdgcn_readfirstlane(nxt.h); cur_id.m_block = __builtin_amdgcn_readfirstlane(nxt.m_block);
  cur_ok = nxt_id_ok;   // (scalar on purpose: carried in vector registers the three values end up in scratch)
  if (!blk_ok) continue;   // (uniform over the workgroup: no barrier is skipped by part of it)
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
  const int key_base = n_min * BN;  // first key of the block's first tile
  // tile (relative to n_min) scored in iteration u, first key of step i (= iteration i / 2, half i % 2)
  auto tile_of = [&](int u) __attribute__((always_inline)) { return DESC ? n_tiles - 1 - u : u; };
  auto step_key = [&](int i) __attribute__((always_inline)) { return key_base + BN * tile_of(i >> 1) + 32 * (i & 1); };
  float slope2 = 0.f;   // ALiBi slope of this head in log2 units
  if constexpr (ALIBI) slope2 = p.alibi[(int64_t)b * p.alibi_bs + h] * 1.4426950408889634f;

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
    lim_lo[qb] = (p.wl >= 0) ? max(0, my_row + shift - p.wl) : 0;   // (>= 0: the descending walk's drain step scores the zero tile left of key 0 -- it must not count as visible)
  }
  // ---- K/V tiles: global -> LDS by DMA through a buffer descriptor.  The LDS image is lane-linear, so the XOR swizzles
  // are applied to the per-lane SOURCE chunk.  A wave issues its DPW pieces of a tile from ONE statement: M0 = LDS base
  // of piece 0, pieces 1.. by the instruction offset (added to the LDS AND the memory address, hence the -1024*i folded
  // into each piece's lane offset).  Rows past the last key are clamped to the last key (finite data, masked to -inf).
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
  // Lanes whose offset lies outside a descriptor's range write ZEROS into LDS (tools/ubench/lds_dma_oob.hip): the rows of a partial
  // last tile past the last key need no clamping (finite data, and the key-length mask hides them), and a descriptor of zero
  // records zero-fills a whole tile without touching memory -- tiles past the last one, and the V tile the pipeline's first two
  // steps multiply by P = 0, are "loaded" that way.
  const u32x4 null_srd = {k_srd[0], k_srd[1], 0u, k_srd[3]};
  // (PAGED) table entry of a page (clamped into the sequence's pages: a tile past the end reads nothing, its descriptor has no records), and the descriptor of
  // absolute tile n: tile tin of page entry blk
  const int pg_tpp = __builtin_amdgcn_readfirstlane(PAGED ? p.page_size / BN : 1);   // tiles per page
  const int pg_last = __builtin_amdgcn_readfirstlane(sk > 0 ? (sk - 1) / max(p.page_size, 1) : 0);   // the sequence's last page
  const int* __restrict__ pg_table = p.block_table + (PAGED ? (int64_t)__builtin_amdgcn_readfirstlane(b) * p.block_table_bs : 0);
  auto page_entry = [&](int page) __attribute__((always_inline)) {
    const int pc = min(max(page, 0), pg_last);
    return __builtin_amdgcn_readfirstlane(pg_table[pc]);
  };
  // (all scalar: a page's bytes and a tile's bytes fit 32 bits -- checked by the launcher --, so the address is one 32 x 32 -> 64 multiply, one 32-bit multiply and
  // two 64-bit adds; at one wave per SIMD every instruction of the iteration's head is exposed)
  const unsigned pg_bytes_k = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(p.k_bs * 2)), pg_bytes_v = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(p.v_bs * 2));
  const unsigned tl_bytes_k = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(BN * p.k_rs * 2)), tl_bytes_v = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(BN * p.v_rs * 2));
  // (readfirstlane returns a SIGNED int: through unsigned before it is widened, or a low half with bit 31 set sign-extends over the high half)
  auto uniform64 = [](const void* ptr) __attribute__((always_inline)) {
    const unsigned long long a = (unsigned long long)ptr;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return ((unsigned long long)hi << 32) | (unsigned long long)lo;
  };
  const unsigned long long pg_base_k = uniform64(kp), pg_base_v = uniform64(vp);
  auto tile_srd = [&](auto isvc, int blk, int tin, int n) __attribute__((always_inline)) {
    constexpr bool ISV = decltype(isvc)::value != 0;
    const unsigned rs = (unsigned)(ISV ? p.v_rs : p.k_rs);
    const int rows = (n >= 0) ? min(BN, sk - n * BN) : 0;
    const unsigned long long a = (ISV ? pg_base_v : pg_base_k) + (unsigned long long)(unsigned)blk * (unsigned long long)(ISV ? pg_bytes_v : pg_bytes_k) + (unsigned long long)((unsigned)tin * (ISV ? tl_bytes_v : tl_bytes_k));
    const unsigned nrec = rows > 0 ? ((unsigned)(rows - 1) * rs + (unsigned)D) * 2u : 0u;
    u32x4 sd = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, nrec, 0x00020000u};
    return sd;
  };
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
  auto dma_tile = [&](auto isvc, int buf, int t) __attribute__((always_inline)) {  // t relative to n_min; t < 0 or >= n_tiles: zero fill
    constexpr bool ISV = decltype(isvc)::value != 0;
    const bool real = t >= 0 && t < n_tiles;
    const int64_t rs = ISV ? p.v_rs : p.k_rs;
    const unsigned lds_dst = (unsigned)((ISV ? 2 + buf : buf) * TILE_BYTES + wave * DPW * 1024);
    if constexpr (PAGED) {   // (prologue and idle iterations: everything made on the spot)
      const int n = n_min + t, page = max(n, 0) / pg_tpp;
      const u32x4 sd = real ? tile_srd(isvc, page_entry(page), max(n, 0) - page * pg_tpp, n) : null_srd;
      dma_pieces(sd, ISV ? voff_l : koff_l, lds_dst);
      return;
    }
    const unsigned toff = real ? (unsigned)(n_min + t) * (unsigned)(BN * 2) * (unsigned)rs : 0u;
    unsigned vo[DPW];
#pragma unroll
    for (int i = 0; i < DPW; ++i) vo[i] = (ISV ? voff_l[i] : koff_l[i]) + toff;
    dma_pieces(real ? (ISV ? v_srd : k_srd) : null_srd, vo, lds_dst);
  };

  // ---- prologue.  The previous block's epilogue staged its O tile over the K/V buffers: nobody may refill them before every
  // wave is through with it.  Q: already prefetched by this wave during the previous block (its own rows, so its own
  // vmcnt wait at the tile barriers made them visible), else loaded now.  K_0 rides under the Q conversion.
  // K_0 rides under the Q conversion; V buffer 1 -- which the first two steps of the pipeline multiply by P = 0 -- is zero-filled.
  __syncthreads();
  if (q_in_lds != vb) dma_q(q_srd_of(blk));
  // (round 4: only K_0 here.  Eight tile DMAs issued back to back cost ~160 clocks each -- the vector-memory queue fills --, 1300 clocks between the barrier
  // and the first LDS read of Q; the zero fill of V buffer 1 is issued between the two halves of the Q conversion instead, profiles/r04_fwd_w64_per_block_code.txt)
  if (n_tiles > 0) dma_tile(ICw<0>{}, 0, tile_of(0));
  if (q_in_lds != vb) {   // Q was not prefetched (first block of this workgroup): its pieces were requested first
    if (n_tiles > 0) { if constexpr (DPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }   // (all but K_0's pieces)
    else lds_dma_wait_all();
  }
  // this wave's B-operand fragments of Q, pre-multiplied by softmax_scale*log2(e) and rounded once to the input dtype, into
  // accumulator registers for the whole block
  {
    const float cq = SOFTCAP ? p.scale * 2.885390081777927f / p.softcap : p.scale_log2;   // (softcap: the chains deliver 2*log2e * score*scale/softcap)
    // (all reads of a query block first: read -> convert -> write one fragment at a time exposes the LDS latency sixteen times)
    u32x4 qraw[QB][KS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int qbase = Q_OFF + (wave * 64 + qb * 32 + qi) * ROW_BYTES + ((hi ^ k_swz_w<D>(qi)) << 4);
        qraw[qb][ks] = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)(qbase ^ (ks << 5));
      }
    auto load_q = [&](auto qbc, auto ksc) __attribute__((always_inline)) {
      constexpr int qb = decltype(qbc)::value, ks = decltype(ksc)::value;
      const V8 raw = bitcast_u32x4<V8>(qraw[qb][ks]);
      V8 sc;
#pragma unroll
      for (int j = 0; j < 8; j += 2) {   // (packed multiply: the conversion is ~400 instructions of un-overlapped per-block code)
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        f32x2 pr = {(float)raw[j], (float)raw[j + 1]};
        pr *= f32x2{cq, cq};
        sc[j] = (E)pr[0]; sc[j + 1] = (E)pr[1];
      }
      acc_write_frag<W64_Q_BASE + 4 * (qb * KS + ks)>(__builtin_bit_cast(u32x4, sc));
    };
    auto load_q_all = [&](auto qbc) __attribute__((always_inline)) {
      load_q(qbc, ICw<0>{}); load_q(qbc, ICw<1>{}); load_q(qbc, ICw<2>{}); load_q(qbc, ICw<3>{});
      if constexpr (KS == 8) { load_q(qbc, ICw<4>{}); load_q(qbc, ICw<5>{}); load_q(qbc, ICw<6>{}); load_q(qbc, ICw<7>{}); }
    };
    load_q_all(ICw<0>{});
    if (n_tiles > 0) dma_tile(ICw<1>{}, 1, -1);   // V buffer 1 <- zeros (the pipeline's first two steps multiply it by P = 0)
    load_q_all(ICw<1>{});
  }
  lds_dma_wait_all();   // K_0 (requested before the conversion)
  __syncthreads();
  // Next block's Q rows of this wave -> LDS under this block's tile loop, a few 1-KiB pieces per iteration (each iteration's
  // tile barrier waits for the pieces requested in it).  All of them at once, as in round 2, put 64 KB per CU -- 16 MB across
  // the chip, every workgroup at the same moment -- in front of the first tiles of the loop: 16k clocks to get the requests
  // out and another 10k of iteration 0 waiting behind them (profiles/r03_fwd_w64_stamps.txt).
  u32x4 qn_srd = null_srd;
  int qn_done = QDMA, qn_per_iter = QDMA;
  {
    q_in_lds = -1;
    if (nxt_id_ok && fill_blk(nxt)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the Q region have returned
      qn_srd = q_srd_of(nxt); qn_done = 0;
      qn_per_iter = n_tiles >= QDMA ? 1 : n_tiles >= QDMA / 2 ? 2 : n_tiles >= QDMA / 4 ? 4 : QDMA;
      if (n_tiles == 0) { dma_q(qn_srd); lds_dma_wait_all(); qn_done = QDMA; }   // no tile barrier will wait for it
      q_in_lds = vb + (int)gridDim.x;
    }
  }
  auto q_trickle = [&]() __attribute__((always_inline)) {   // called once per iteration, outside the steps (M0 is theirs inside)
    if (qn_done < QDMA) {
#pragma unroll 1
      for (int j = 0; j < qn_per_iter; ++j) dma_q_piece(qn_srd, qn_done + j);
      qn_done += qn_per_iter;
    }
  };

  // per-lane LDS read bases: K fragment of k-step ks at ka[ks] (+ buffer / half offsets as immediates), V d-block db at va[db]
  const int kbase = qi * ROW_BYTES + ((hi ^ k_swz_w<D>(qi)) << 4);
  const int tr_i = lane & 15, tr_half = (lane >> 4) & 1;
  const int tr_rr = tr_i >> 2, tr_cc = tr_i & 3;
  const int vbase = (4 * hi + tr_rr) * ROW_BYTES + (v_swz_w<D>(tr_rr) << 6) + tr_half * 32 + tr_cc * 8;
  // The buffer parity of an iteration (iteration u reads K buffer u & 1 and V buffer (u - 1) & 1) is a compile-time constant of the step: the
  // tile loop's body holds an even and an odd iteration, so the parity rides in the reads' immediate offsets (a run-time parity carried by the
  // bases cost twelve v_xor per iteration -- and at one wave per SIMD every instruction of the loop costs its ~4.7 clocks,
  // profiles/r03_fwd_w64_ablations.txt).
  int ka[KS], va[DB];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) ka[ks] = kbase ^ (ks << 5);
#pragma unroll
  for (int db = 0; db < DB; ++db) va[db] = vbase ^ (db << 6);

  // O = 2*DB tuples: a[0 : 32*DB) (query block qb, d-block db at tuple qb*DB + db), zeroed by the matrix pipe (0 . 0 + 0: eight
  // instructions; 128 v_accvgpr_write statements came with 120 pad s_nop from hipcc, ~1.2k clocks per block)
  {
    u32x4 zf = {0u, 0u, 0u, 0u};
    asm volatile("" : "+v"(zf));
    acc_zero_tuples_mfma<0>(zf, std::make_integer_sequence<int, 2 * DB>{});
  }
  float m_run[QB], l_run[QB][2], o_lag[QB];
  float thr_l[QB];          // per-lane decision threshold: -inf until the row has seen a key (any finite score moves m), then rescale_thr
  // (SOFTCAP) c = softcap*log2e (the capped scores live in [-c, c] log2 units), -2c, and per row off = c - m_base
  float capc = 0.f, cap_m2c = 0.f, capoff[QB] = {0.f, 0.f};
  if constexpr (SOFTCAP) {
    capc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.softcap * 1.4426950408889634f)));
    cap_m2c = -2.f * capc;
    capoff[0] = capc; capoff[1] = capc;
  }
  // (DROP) the seven round keys of this (batch, head) -- uniform --, the keep threshold, and each lane's row (the second word of the Philox counter)
  unsigned drop_kr[7] = {0u, 0u, 0u, 0u, 0u, 0u, 0u}, drop_thr = 255u, drop_row[QB] = {0u, 0u};
  if constexpr (DROP) {
    const unsigned k0_ = (unsigned)__builtin_amdgcn_readfirstlane((int)drop_bh_key(p.rng, b * p.h + h));
#pragma unroll
    for (int r = 0; r < 7; ++r) drop_kr[r] = k0_ + (unsigned)r * 0x9E3779B9u;
    drop_thr = (unsigned)__builtin_amdgcn_readfirstlane((int)p.drop_thr8);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) drop_row[qb] = (unsigned)(w_row0 + 32 * qb + qi);
  }
  constexpr float kCapTiny = 7.888609052210118e-31f;   // 2^-100: y * tiny vanishes for every finite y and keeps a masked score's -inf
  unsigned long long lag_mask = 0ull;   // wave-uniform, all ones or zero: some o_lag != 1 is waiting to be applied to O
  f32x16 negm[QB];     // the C operand of every score chain's first MFMA: -m broadcast (0 while m = -inf), plus the ALiBi bias of the step's keys
  // ALiBi: element r of a step has the bias slope2 * (step_key + 4*hi + acc_row(r, 0) - row - shift).  The key distance is formed in INTEGERS first
  // (arel = the lane's part, 4*hi - row - shift) and converted once: round 3 added two fp32 terms, slope2 * (4*hi - row - shift) + slope2 * key, which
  // cancel -- exact enough below 16k keys, ~1e-2 log2 units at 128k keys even for the keys next to the diagonal, the ones that matter
  // (tests/test_fwd_gpu.py: the 128k-key ALiBi case).  The in-place adds of the steady state only ever add +-slope2 * 32 / 96 to values whose size is
  // the bias itself: their rounding is relative to a distance-sized term, i.e. small exactly where the probabilities are not.
  int arel[QB] = {0, 0};
  if constexpr (ALIBI) {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) arel[qb] = 4 * hi - (w_row0 + 32 * qb + qi) - shift;
  }
  const float ainc = slope2 * 32.f;
  f32x16 sA[QB], sB[QB];
  u32x4 pfA[QB][2], pfB[QB][2];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = -INFINITY; l_run[qb][0] = 0.f; l_run[qb][1] = 0.f; o_lag[qb] = 1.f; thr_l[qb] = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) { negm[qb][r] = ALIBI ? slope2 * (float)(arel[qb] + step_key(0) + acc_row(r, 0)) : 0.f; sA[qb][r] = -INFINITY; sB[qb][r] = -INFINITY; }   // S_{-1} = -inf: P_{-1} = 0
#pragma unroll
    for (int t = 0; t < 2; ++t) { pfA[qb][t] = u32x4{0u, 0u, 0u, 0u}; pfB[qb][t] = u32x4{0u, 0u, 0u, 0u}; }
  }

  // Decision on the NEXT step's scores, held as s' = s - m_base (m_base = m, or 0 while m = -inf): the row moves its maximum
  // when max(s') > thr_l, i.e. when it grew by more than rescale_thr -- or, for a row that has not seen a key yet (thr_l = -inf),
  // as soon as any score is finite (same rule as fa_fwd_il.hip: (m_new - m_run) > thr with m_new = max(m_run, max s)).
  // The cold path moves m and l at once, re-bases the pending scores and the C broadcast, and rescales O one step later
  // (after P_i.V, computed at the old scale, has been accumulated) -- the lagged rescale of fa_fwd_il.hip.
  auto rescale = [&](auto qbc, bool grow, float tmax, f32x16& s_nxt) __attribute__((always_inline)) {
    constexpr int qb = decltype(qbc)::value;
    // (the base the pending scores are held against, recomputed from m: a masked step's C broadcast carries -inf in its masked elements)
    const float m_base = (m_run[qb] == -INFINITY) ? 0.f : m_run[qb];
    const float m_upd = grow ? (tmax + m_base) : m_run[qb];   // grow => tmax finite
    const float m_safe = (m_upd == -INFINITY) ? 0.f : m_upd;
    // (a row that sees its first key -- m = -inf -- has O = 0 and l = 0: its factor is 1, so the first decision of a block, which moves
    // every row, leaves nothing to apply to O one step later)
    const float alpha = (grow && m_run[qb] != -INFINITY) ? fast_exp2(m_run[qb] - m_safe) : 1.f;
    const float delta = m_safe - m_base;   // new base - old base (0 where the row did not move)
    m_run[qb] = m_upd;
    thr_l[qb] = grow ? thr : thr_l[qb];
    l_run[qb][0] *= alpha;
    l_run[qb][1] *= alpha;
    // in place, element by element through tied asm operands: a recomputed tuple would live in NEW registers and cost the
    // common path a 16-register copy at the join
    const float neg = -m_safe;
    if constexpr (SOFTCAP) capoff[qb] = capc - m_safe;   // (the pending scores are raw: nothing to re-base, the C broadcast stays 0 / -inf)
#pragma unroll
    for (int r = 0; r < (SOFTCAP ? 0 : 16); ++r) {
      float sv = s_nxt[r], nv = negm[qb][r];
      if constexpr (ALIBI) asm volatile("v_sub_f32 %0, %0, %2\n\tv_sub_f32 %1, %1, %2" : "+v"(sv), "+v"(nv) : "v"(delta), "v"(neg));   // (the bias stays; -inf stays -inf)
      else asm volatile("v_sub_f32 %0, %0, %2\n\tv_mov_b32 %1, %3" : "+v"(sv), "+v"(nv) : "v"(delta), "v"(neg));
      s_nxt[r] = sv;
      negm[qb][r] = nv;
    }
    // O of this query block (accumulator registers 16*DB*qb ..), two registers at a time; skipped when no factor is pending
    if (lag_mask != 0ull) acc_scale_range<16 * DB * qb>(o_lag[qb], std::make_integer_sequence<int, 16 * DB>{});
    o_lag[qb] = alpha;
  };
  // tmax = row maxima of s' (already combined across the lane halves)
  auto decide_and_rescale = [&](const float (&tmax)[QB], f32x16 (&s_nxt)[QB]) __attribute__((always_inline)) {
    // (the two compares and the OR in one statement: hipcc's own rendering of "any lane" is 10 instructions at every step boundary)
    unsigned long long grow_mask;
    asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cmp_gt_f32 %0, %3, %4\n\ts_or_b64 %0, %0, vcc"
                 : "=&s"(grow_mask) : "v"(tmax[0]), "v"(thr_l[0]), "v"(tmax[1]), "v"(thr_l[1]) : "vcc");
    // cold: ~4 KB of straight-line code per call site.  Left in line it sat between the steps of the tile loop and every
    // step paid an instruction-fetch miss jumping over it (18 of 63 clocks per MFMA, profiles/r03_fwd_w64_ablations.txt);
    // __builtin_expect moves it behind the loop.
    if (__builtin_expect((grow_mask | lag_mask) != 0ull, 0)) {   // the step's one cold exit
      const bool g0 = tmax[0] > thr_l[0], g1 = tmax[1] > thr_l[1];
      mfma_drain_acc();  // O is about to be read by the VALU
      rescale(ICw<0>{}, g0, tmax[0], s_nxt[0]);
      rescale(ICw<1>{}, g1, tmax[1], s_nxt[1]);
      lag_mask = __builtin_amdgcn_ballot_w64(o_lag[0] != 1.f || o_lag[1] != 1.f) != 0ull ? ~0ull : 0ull;
    }
  };
  // Masks ride in the C operand of a score chain's first MFMA: the broadcast -m (negm) gets -inf in the elements of masked (row, key) pairs, the
  // scores leave the matrix pipe as -inf there, and the step -- its row-max tree, its exp2, its decision -- runs unchanged.  set_mask(i) rewrites the
  // two broadcasts for step i (32 rows of a query block against 32 keys: all visible / none / element by element, from the step ranges computed once
  // per block), clear_mask() restores them; both run between steps, only in the iterations that straddle a mask boundary (a wave's diagonal tile,
  // window edges, the partial last tile, the drain's empty chains).  (Rounds 2-3 tried masked step VARIANTS, a pre-loaded chain start in separate
  // loops, and masking the finished scores in the step's cold exit + a second row-max tree: 30k / 30k / 26k clocks for the last five iterations
  // of a block under a causal mask against 17.6k with nothing masked, profiles/r03_fwd_w64_stamps.txt.)
  // (Round 4 tried a fast path for the common case -- a wave's tile-aligned causal diagonal and the drain behind it, mask rewrites from a per-lane constant
  // triangle, 112 instead of ~420 instructions per masked iteration: those iterations went from ~4180 to ~4120 clocks, the first iteration and the steady state
  // lost as much to the longer code.  A masked iteration's cost follows the code layout, not the instruction count: profiles/r04_fwd_w64_stamps.txt; removed.)
  // Straight-line on purpose, one path for every case (all visible / none / element by element fall out of the per-lane bounds): with a branch
  // per case the rewritten broadcasts of the cases meet at joins and hipcc copies 32 registers per call.
  auto set_mask = [&](int i_step) __attribute__((always_inline)) {
    const int k0m = step_key(i_step);
    float ninf = -INFINITY;
    asm volatile("" : "+v"(ninf));
    static_for<QB>([&](auto mqc) __attribute__((always_inline)) {
      constexpr int mq = decltype(mqc)::value;
      float nb = (SOFTCAP || m_run[mq] == -INFINITY) ? 0.f : -m_run[mq];
      // element r of this lane scores key k0m + 4*hi + acc_row(r, 0): the lane's visibility bitmap over those offsets, then two instructions per
      // element (sign-extended bit -> select mask -> bit-field insert)
      const int rel_hi = min(lim_hi[mq] - k0m - 4 * hi, 31), rel_lo = max(lim_lo[mq] - k0m - 4 * hi, 0);
      const unsigned ones = (rel_hi - rel_lo >= 31) ? 0xffffffffu : ((2u << ((rel_hi - rel_lo) & 31)) - 1u);
      unsigned bits = (rel_hi >= rel_lo) ? (ones << (rel_lo & 31)) : 0u;   // (rel_lo > 31 only with rel_hi < rel_lo)
      if constexpr (ALIBI) nb += slope2 * (float)(arel[mq] + k0m);   // the step's bias of this lane's element 0
      asm volatile("" : "+v"(nb), "+v"(bits));
#pragma unroll
      for (int r = 0; r < 16; r += 4) {
        float n0 = negm[mq][r], n1 = negm[mq][r + 1], n2 = negm[mq][r + 2], n3 = negm[mq][r + 3];
        unsigned t0, t1;
        if constexpr (ALIBI) {   // (the visible value differs per element: base + slope2 * offset)
          const float v0 = nb + slope2 * (float)acc_row(r, 0), v1 = nb + slope2 * (float)acc_row(r + 1, 0);
          const float v2 = nb + slope2 * (float)acc_row(r + 2, 0), v3 = nb + slope2 * (float)acc_row(r + 3, 0);
          asm volatile("v_bfe_i32 %4, %6, %c11, 1\n\tv_bfe_i32 %5, %6, %c12, 1\n\tv_bfi_b32 %0, %4, %7, %15\n\tv_bfi_b32 %1, %5, %8, %15\n\t"
                       "v_bfe_i32 %4, %6, %c13, 1\n\tv_bfe_i32 %5, %6, %c14, 1\n\tv_bfi_b32 %2, %4, %9, %15\n\tv_bfi_b32 %3, %5, %10, %15"
                       : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "=&v"(t0), "=&v"(t1)
                       : "v"(bits), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "i"(acc_row(r, 0)), "i"(acc_row(r + 1, 0)), "i"(acc_row(r + 2, 0)), "i"(acc_row(r + 3, 0)),
                         "v"(ninf));
        } else {
          asm volatile("v_bfe_i32 %4, %6, %c9, 1\n\tv_bfe_i32 %5, %6, %c10, 1\n\tv_bfi_b32 %0, %4, %7, %8\n\tv_bfi_b32 %1, %5, %7, %8\n\t"
                       "v_bfe_i32 %4, %6, %c11, 1\n\tv_bfe_i32 %5, %6, %c12, 1\n\tv_bfi_b32 %2, %4, %7, %8\n\tv_bfi_b32 %3, %5, %7, %8"
                       : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "=&v"(t0), "=&v"(t1)
                       : "v"(bits), "v"(nb), "v"(ninf), "i"(acc_row(r, 0)), "i"(acc_row(r + 1, 0)), "i"(acc_row(r + 2, 0)), "i"(acc_row(r + 3, 0)));
        }
        negm[mq][r] = n0; negm[mq][r + 1] = n1; negm[mq][r + 2] = n2; negm[mq][r + 3] = n3;
      }
    });
  };
  auto clear_mask = [&](int i_next) __attribute__((always_inline)) {   // i_next: the step whose keys the broadcast serves next (ALiBi)
    static_for<QB>([&](auto mqc) __attribute__((always_inline)) {
      constexpr int mq = decltype(mqc)::value;
      float nb = (SOFTCAP || m_run[mq] == -INFINITY) ? 0.f : -m_run[mq];
      if constexpr (ALIBI) nb += slope2 * (float)(arel[mq] + step_key(i_next));
      asm volatile("" : "+v"(nb));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float nv = negm[mq][r];
        float val = nb;
        if constexpr (ALIBI) val = nb + slope2 * (float)acc_row(r, 0);
        asm volatile("v_mov_b32 %0, %1" : "+v"(nv) : "v"(val));
        negm[mq][r] = nv;
      }
    });
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
  auto fast_step = [&](auto halfc, auto parc, f32x16 (&s_cur)[QB], f32x16 (&s_nxt)[QB],
                       const u32x4 (&pf_prev)[QB][2], u32x4 (&pf_cur)[QB][2], const u32x4& dma_srd, const unsigned (&dma_off)[DPW],
                       unsigned dma_toff, unsigned dma_dst, u32x4 (&kfr)[FA_W64_AH + 1], int drop_k0q = 0, auto modec = ICw<0>{}) __attribute__((always_inline)) {
    constexpr int half = decltype(halfc)::value, par = decltype(parc)::value;
    // MODE (round 5): a wave's FIRST iteration has no previous probabilities (P = 0 by construction: its P.V MFMAs, and in its first step the exp2 work, are dead) and
    // its LAST one scores no key of its own (the chains would be masked end to end): 1 = score chains only, 2 = score chains + this step's P, 3 = this step's P + P.V
    // (+ the pending rescale), 4 = P.V only.  The dead halves' MFMAs, LDS reads and vector work are not issued: two of a block's iterations cost about a third of one.
    constexpr int MODE = decltype(modec)::value;
    constexpr bool DO_QK = MODE <= 2, DO_P = MODE == 0 || MODE == 2 || MODE == 3, DO_PV = MODE == 0 || MODE >= 3;
    constexpr int QKG = 2 * KS, PVG = 4 * DB, NG = QKG + PVG;
    // (DROP) group G = 4*qb + g (elements 16*qb + 4*g .. + 3 = keys k0 + 8*g + 4*hi .. + 3 of the step whose probabilities this step makes, k0 = 4 * drop_k0q):
    // round r of its Philox call sits in gap ((7*G + r) * (NG - 2)) / 56 -- two rounds per gap at D = 128, the last group done one gap before its elements are packed
    auto drop_gap = [](int G, int r) constexpr { return ((7 * G + r) * (NG - 2)) / 56; };
    // (the word of group G is used where the group's first element is packed: gap QKG + (4*G) / CPG of the P.V half, CPG conversions per gap; the backward twin has the same check)
    constexpr bool drop_in_time = [&]() constexpr {
      constexpr int cpg = 16 / PVG > 0 ? 16 / PVG : 1;
      for (int G = 0; G < 8; ++G) if (drop_gap(G, 6) >= QKG + (2 * G) / cpg) return false;
      return true;
    }();
    static_assert(!DROP || drop_in_time, "a Philox word is finished before its first element is packed");
    unsigned ph_c0[8], ph_c1[8];
    constexpr int KOFF = par * TILE_BYTES + half * 32 * ROW_BYTES;                       // K_u: buffer u & 1
    constexpr int VOFF = (2 + (par ^ 1)) * TILE_BYTES + half * 32 * ROW_BYTES;           // V_{u-1}: buffer (u - 1) & 1
    constexpr int AH = FA_W64_AH, RING = AH + 1;  // operand reads run AH fragment slots (2 gaps each) ahead of their MFMAs
    constexpr int NF = KS + 2 * DB;       // fragment slots per step: KS K fragments, then 2*DB V fragments
    static_assert(RING == FA_W64_AH + 1, "the K fragment ring is the caller's (carried from a first step to its second step)");
    s16x4 vlo[RING], vhi[RING];
    auto rd_frag = [&](int f) __attribute__((always_inline)) {
      if (f < KS) {
        if constexpr (DO_QK) kfr[f % RING] = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)(ka[f] + KOFF);
      } else if (f < NF && DO_PV) {
        const int op = f - KS;
        const char FA_LDS* a0 = (const char FA_LDS*)(unsigned long)(unsigned)(va[op % DB] + VOFF + (16 * (op / DB)) * ROW_BYTES);
        vlo[f % RING] = lds_read_tr16(a0);
        vhi[f % RING] = lds_read_tr16(a0 + 8 * ROW_BYTES);
      }
    };
    // element e of P_i (e = 16*qb + r): done-by-gap schedule
    // (SOFTCAP: no element takes this path -- its staged form follows its own schedule, cap_a below)
    auto el_end = [](int x) constexpr { return SOFTCAP ? 32 : x <= QKG ? (24 * x) / QKG : (24 + ((x - QKG) * 16) / PVG > 32 ? 32 : 24 + ((x - QKG) * 16) / PVG); };
    float pe[QB][16];   // P_i as scalars (writing them back into the score tuples makes hipcc copy whole 16-register tuples)
    // (SOFTCAP) stage A of element e is issued in the gaps before x for e < cap_a(x): 20 elements under the score chains, 12 under the first PVG - 2 gaps of the
    // PV half; stages B and C follow one and two gaps later (the last element's stage C sits in the step's last gap)
    auto cap_a = [](int x) constexpr {
      if (x <= 0) return 0;
      if (x <= QKG) return (20 * x) / QKG;
      const int n = 20 + ((x - QKG) * 12 + (PVG - 3)) / (PVG - 2);
      return n > 32 ? 32 : n;
    };
    static_assert(!SOFTCAP || cap_a(NG - 2) == 32, "every element's last stage inside the step");
    float cap_a1[32], cap_d[32], cap_arg[32], cap_g[QB] = {0.f, 0.f}, cap_gt[QB] = {-INFINITY, -INFINITY};   // values in flight between the stages (a gap or two each)
    float tmax[QB] = {-INFINITY, -INFINITY}, tcopy[QB] = {0.f, 0.f};
    // gap (inside the PV half) schedule of the row-max work of query block mq
    constexpr int UPG = PVG >= 16 ? 1 : 2;        // max3 units per gap
    auto tree_g0 = [](int mq) constexpr { return 1 + mq; };   // the chain of block mq retired at gap QKG - 2 + mq
    auto hm_gap = [&](int mq) constexpr { return tree_g0(mq) + 8 / UPG; };           // cross-half combine right after the tree
    if constexpr (half != 1) {   // (second step: its first AH fragments were requested by the first step, see the end of the gap loop)
#pragma unroll
      for (int f = 0; f < AH; ++f) rd_frag(f);
    }
    __builtin_amdgcn_sched_barrier(0);
    static_for<NG>([&](auto xc) __attribute__((always_inline)) {
      constexpr int x = decltype(xc)::value;
      constexpr int f = x / 2, qb = x & 1;
      if constexpr (qb == 0) rd_frag(f + AH);
      // One wait per TWO fragment slots: before the MFMAs of an even slot f, wait until slot f + 1 has landed too (everything requested
      // after it may still be in flight: slots f + 2 .. f + AH, one LDS instruction per K fragment, two per transposed V fragment).  hipcc
      // models an explicit s_waitcnt and drops its own wait in front of slot f + 1 -- sixteen fewer instructions per iteration.
      if constexpr (qb == 0 && (f & 1) == 0 && f + 1 < NF) {
        constexpr auto ops = [](int g) constexpr { return g < KS ? (DO_QK ? 1 : 0) : g < NF ? (DO_PV ? 2 : 0) : 0; };
        constexpr int out = [&]() constexpr { int n = 0; for (int g = f + 2; g <= f + AH; ++g) n += ops(g); return n; }();
        if constexpr (f < KS ? DO_QK : DO_PV) __builtin_amdgcn_s_waitcnt(0xC07F | (out << 8));
      }
      if constexpr (x < QKG ? !DO_QK : !DO_PV) {
        // (a dead half: no MFMA in this gap)
        // MODE 1 / 2: the row-max tree (gaps QKG + 1 ..) reads the score chains a few instructions behind their last MFMAs -- in a full step sixteen P.V MFMAs sit
        // in between; the MFMAs are inline asm, hipcc pads nothing: the wait states of "matrix pipe writes a VGPR, VALU reads it" are spent here
        if constexpr (x == QKG && DO_QK) mfma_drain_acc();
      } else if constexpr (x < QKG) {
        if constexpr (f == 0) mfma_s_first<E, qb * KS>(s_nxt[qb], kfr[f % RING], negm[qb]);
        else mfma_s_acc<E, qb * KS + f>(s_nxt[qb], kfr[f % RING]);
      } else {
        constexpr int op = f - KS;
        const u32x4 vf = __builtin_bit_cast(u32x4, combine_tr<V8>(vlo[f % RING], vhi[f % RING]));
        mfma_o_acc<E, qb * DB + op % DB>(vf, pf_prev[qb][op / DB]);
      }
      // this step's DMA pieces (K_{u+1} in the first step of an iteration, V_u in the second): gaps G0, G0 + GS, ..
      constexpr int G0 = half == 0 ? FA_W64_KDMA_G0 : FA_W64_VDMA_G0, GS = half == 0 ? FA_W64_KDMA_GS : FA_W64_VDMA_GS;
      if constexpr (x >= G0 && (x - G0) % GS == 0 && (x - G0) / GS < DPW) {
        constexpr int pc = (x - G0) / GS;
        // (the tile's byte offset rides in the scalar-offset operand; the range check accounts for it: tools/ubench/lds_dma_oob.hip)
        if constexpr (pc == 0)
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds" : : "v"(dma_off[pc]), "s"(dma_dst), "s"(dma_srd), "s"(dma_toff) : "memory");
        else
          asm volatile("buffer_load_dwordx4 %0, %1, %3 offen offset:%c2 lds" : : "v"(dma_off[pc]), "s"(dma_srd), "i"(1024 * pc), "s"(dma_toff) : "memory");
      }
      if constexpr (ALIBI) {   // the C broadcasts move on to the next step's keys (the other half of the tile: +32 keys; the next tile down: -96): 32 in-place adds
        constexpr auto ab_end = [](int g) constexpr { return g < 4 ? 0 : (((g - 3) * 32 + (NG - 5)) / (NG - 4) > 32 ? 32 : ((g - 3) * 32 + (NG - 5)) / (NG - 4)); };
        const float step_add = (half == 1 && DESC) ? -3.f * ainc : ainc;
#pragma unroll
        for (int e = ab_end(x); e < ab_end(x + 1); ++e) {
          float nv = negm[e >> 4][e & 15];
          asm volatile("v_add_f32 %0, %0, %1" : "+v"(nv) : "v"(step_add));
          negm[e >> 4][e & 15] = nv;
        }
      }
      if constexpr (DROP) {
#pragma unroll
        for (int G = 0; G < 8; ++G)
#pragma unroll
          for (int r = 0; r < 7; ++r)
            if (drop_gap(G, r) == x) {
              if (r == 0) { ph_c0[G] = (unsigned)(drop_k0q + 2 * (G & 3)) + (unsigned)hi; ph_c1[G] = drop_row[G >> 2]; }
              const unsigned mh = __umulhi(0xD256D193u, ph_c0[G]), ml = 0xD256D193u * ph_c0[G];
              unsigned x3;   // mh ^ round key ^ c1 in ONE instruction (the key is a scalar operand; hipcc renders the C expression as two v_xor)
              asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(x3) : "v"(mh), "s"(drop_kr[r]), "v"(ph_c1[G]));
              ph_c0[G] = x3;
              ph_c1[G] = ml;
            }
      }
      if constexpr (SOFTCAP) {
        // three stages, oldest elements first: C = exp2 + row sum, B = rcp + the exponent, A = the mask carrier, exp2 and +1
#pragma unroll
        for (int e = cap_a(x - 2); e < cap_a(x - 1); ++e) {
          const int eq = e >> 4, r = e & 15;
          pe[eq][r] = fast_exp2(cap_arg[e]);
          l_run[eq][r & 1] += pe[eq][r];
        }
#pragma unroll
        for (int e = cap_a(x - 1); e < cap_a(x); ++e) cap_arg[e] = __builtin_fmaf(__builtin_amdgcn_rcpf(cap_d[e]), cap_m2c, cap_a1[e]);
#pragma unroll
        for (int e = cap_a(x); e < cap_a(x + 1); ++e) {
          const float yv = s_cur[e >> 4][e & 15];
          cap_a1[e] = __builtin_fmaf(yv, kCapTiny, capoff[e >> 4]);
          cap_d[e] = fast_exp2(yv) + 1.f;
        }
      }
      // exp2 + row sums (two running sums per query block, carried across steps)
#pragma unroll
      for (int e = (DO_P ? el_end(x) : 32); e < (DO_P ? el_end(x + 1) : 32); ++e) {
        const int eq = e >> 4, r = e & 15;
        pe[eq][r] = fast_exp2(s_cur[eq][r]);
        l_run[eq][r & 1] += pe[eq][r];
      }
      if constexpr (x >= QKG) {
        constexpr int y = x - QKG;  // gap inside the PV half
        // packing of P_i: PVG gaps, 16 conversions (one packed register each)
        constexpr int CPG = 16 / PVG > 0 ? 16 / PVG : 1;
#pragma unroll
        for (int c = (DO_P ? y * CPG : 16); c < (y + 1) * CPG && c < 16; ++c) {
          const int j = c >> 2, m = c & 3, cq = j >> 1, t = j & 1;
          using V2 = __attribute__((ext_vector_type(2))) E;
          V2 pr;
          float pv0 = pe[cq][8 * t + 2 * m], pv1 = pe[cq][8 * t + 2 * m + 1];
          if constexpr (DROP) {   // element r = 8*t + 2*m: byte r & 3 of its group's word; kept iff byte <= threshold (fa_kernel_params.h drop_thr8)
            const int r0 = 8 * t + 2 * m;
            const unsigned w = ph_c0[4 * cq + (r0 >> 2)];
            pv0 = (((w >> (8 * (r0 & 3))) & 0xffu) > drop_thr) ? 0.f : pv0;
            pv1 = (((w >> (8 * ((r0 + 1) & 3))) & 0xffu) > drop_thr) ? 0.f : pv1;
          }
          pr[0] = (E)pv0;
          pr[1] = (E)pv1;
          unsigned pw = __builtin_bit_cast(unsigned, pr);
          asm volatile("" : "+v"(pw));   // pinned to this gap (hipcc otherwise sinks the conversions into the next step's head)
          pf_cur[cq][t][m] = pw;
        }
        // the row-max tree of the fresh scores, the cross-half combine
#pragma unroll
        for (int mq = (DO_QK ? 0 : QB); mq < QB; ++mq) {
          const int g0 = tree_g0(mq);
          if (y >= g0 && y < g0 + 8 / UPG) {
#pragma unroll
            for (int u = (y - g0) * UPG; u < (y - g0 + 1) * UPG; ++u)  // 16 values in 8 ops: max3(s0,s1,s2), 6 x max3(t,.,.), max(t,s15)
              if (u == 7 && UPG == 1) {   // the tree's last maximum and the copy for the cross-half swap in one statement (no pad between them)
                float t_in = tmax[mq], t_out, t_cp;
                asm volatile("v_max_f32 %0, %2, %3\n\tv_mov_b32 %1, %0" : "=&v"(t_out), "=v"(t_cp) : "v"(t_in), "v"(s_nxt[mq][15]));
                tmax[mq] = t_out; tcopy[mq] = t_cp;
              } else
              tmax[mq] = u == 0 ? vmax3(s_nxt[mq][0], s_nxt[mq][1], s_nxt[mq][2])
                                : u < 7 ? vmax3(tmax[mq], s_nxt[mq][2 * u + 1], s_nxt[mq][2 * u + 2]) : vmax2(tmax[mq], s_nxt[mq][15]);
          }
          // cross-half combine in two statements a gap apart: the copy, then swap + max (v_permlane32_swap wants two wait states
          // after the write of its operand: the gap's other instructions provide them, no s_nop)
          if (y == hm_gap(mq) - 1 && UPG != 1) asm volatile("v_mov_b32 %0, %1" : "=v"(tcopy[mq]) : "v"(tmax[mq]));
          if (y == hm_gap(mq)) {
            // (MODE 1 / 2: the gap's other instructions are gone -- the two wait states are spent explicitly; tools/isa_mfma_hazards.py checks the distance)
            if constexpr (!DO_PV) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1" : "+v"(tmax[mq]), "+v"(tcopy[mq]));
            else asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1" : "+v"(tmax[mq]), "+v"(tcopy[mq]));
          }
          if constexpr (SOFTCAP) {   // the decision's operand: the row maximum through the cap, relative to the row's base -- c*tanh(z_max) - m_base, -inf kept
            constexpr int last = PVG - 1;
            const int g1 = hm_gap(mq) + 1 < last ? hm_gap(mq) + 1 : last, g2 = hm_gap(mq) + 2 < last ? hm_gap(mq) + 2 : last, g3 = hm_gap(mq) + 3 < last ? hm_gap(mq) + 3 : last;
            if (y == g1) cap_g[mq] = fast_exp2(tmax[mq]) + 1.f;
            if (y == g2) cap_g[mq] = __builtin_fmaf(__builtin_amdgcn_rcpf(cap_g[mq]), cap_m2c, capoff[mq]);
            if (y == g3) cap_gt[mq] = __builtin_fmaf(tmax[mq], kCapTiny, cap_g[mq]);
          }
        }
      }
      // carry: behind this step's last LDS wait (gap NG - 4: slot NF - 2, everything landed) the K ring is free -- request the first AH fragments of the SECOND
      // step's score chain (half 1 of the same K tile), one per gap; that step's waits count them exactly as if it had issued them itself
      if constexpr (half == 0 && x >= NG - AH && DO_QK) {   // (MODE 1 is followed by MODE 2, which scores; MODE 3 by MODE 4, which does not)
        constexpr int fn = x - (NG - AH);
        kfr[fn % RING] = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)(ka[fn] + par * TILE_BYTES + 32 * ROW_BYTES);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int mq = 0; mq < QB; ++mq)
      if (hm_gap(mq) >= PVG && DO_QK) tmax[mq] = vhalf_max(tmax[mq]);   // (no gap left for it)
    static_assert(!SOFTCAP || QB + 8 / UPG < PVG - 1, "softcap: the capped maximum is made inside the gaps");
    if constexpr (SOFTCAP) { tmax[0] = cap_gt[0]; tmax[1] = cap_gt[1]; }   // (the decision's operand: the capped maximum relative to the row's base)
    // (MODE 3: no fresh scores -- tmax = -inf moves nothing -- but a rescale factor decided one step ago is still pending: it has to meet O BEFORE the next step
    // accumulates the probabilities that were made at the new base.  MODE 4: nothing can be pending.)
    if constexpr (MODE != 4) decide_and_rescale(tmax, s_nxt);
  };

  // Iteration u (0 .. n_tiles) = two steps: 2u-1 and 2u score K_u (K buffer u & 1) and multiply by V_{u-1} (V buffer (u - 1) & 1);
  // K_{u+1} and V_u are DMA'd during it into the buffers it does not read.  ONE body runs every iteration a wave takes part in,
  // pipeline fill and drain included: a chain that scores no real key (step < 0 side of the fill never occurs -- the fill's first
  // chain is step 0; steps past the wave's last key, past the last tile) gets C = -inf in every element (set_mask), its scores are
  // -inf, its P is 0 and it never moves a maximum; the fill multiplies P = 0 by the zero tile the prologue put into V buffer 1.
  // That wastes four half-steps of MFMA work per wave and block, and replaces the compiler-ordered "generic" fill / drain steps of
  // round 2 -- which, with their own copies of the cold rescale code, made the kernel 240 KB and every once-per-block path an
  // instruction-cache miss stream (12-17k clocks for iteration 0 against 3.2k for a steady-state iteration).
  // Waves outside their range (rows past the sequence end, the early-finishing waves of a block under a causal mask, windows) only
  // issue their share of the tile DMAs and meet the barriers.
  // Per wave the iterations 0 .. n_tiles fall into three consecutive ranges -- idle | active | idle -- walked by two tight loops (entered
  // and the active loop between them); inside the active range the iterations outside [p_lo, p_hi] hold a step that straddles a mask
  // boundary: set_mask() / clear_mask() run around their steps.
  int u_first = n_tiles + 1, u_last = n_tiles, p_lo = n_tiles + 1, p_hi = n_tiles;   // all idle
  if (wave_valid && n_tiles > 0) {
    const int a_lo = max(0, (w_kmin - key_base) >> 5);               // first / last step with a key this wave can see
    const int a_hi = min(n_steps - 1, (w_kmax - key_base) >> 5);
    if (a_hi >= a_lo) {
      u_first = a_lo >> 1; u_last = min(n_tiles, (a_hi + 2) >> 1);
      const int f_lo = max(0, (w_full_lo - key_base + 31) >> 5);     // first step with no left-masked key
      const int f_hi = (w_full_hi - 31 - key_base) >> 5;             // last step with no right-masked key (< n_steps: w_full_hi < sk)
      p_lo = max(u_first, (f_lo + 1) >> 1);
      p_hi = min(u_last, (f_hi - 1) >> 1);
      if constexpr (DESC) {   // the same tiles, walked downwards: tile t is scored in iteration n_tiles - 1 - t, the drain follows the lowest visible tile
        const int t_lo = a_lo >> 1, t_hi = a_hi >> 1, pl_t = (f_lo + 1) >> 1, ph_t = (f_hi - 1) >> 1;
        u_first = n_tiles - 1 - t_hi; u_last = n_tiles - t_lo;
        p_lo = max(u_first, n_tiles - 1 - ph_t);
        p_hi = min(u_last, n_tiles - 1 - pl_t);
      }
      if (p_hi < p_lo) { p_lo = u_last + 1; p_hi = u_last; }         // no plain iteration: one masked range
    }
  }
  u_first = __builtin_amdgcn_readfirstlane(u_first); u_last = __builtin_amdgcn_readfirstlane(u_last);   // (wave-uniform by construction; said so)
  const unsigned wave_dst = (unsigned)(wave * DPW * 1024);
  auto iter_end = [&]() __attribute__((always_inline)) {
    lds_dma_wait_all();
    __syncthreads();
  };
  // Active iterations outside [m_lo, m_hi] hold a step that straddles a mask boundary
  const int m_lo = __builtin_amdgcn_readfirstlane(p_lo), m_hi = __builtin_amdgcn_readfirstlane(p_hi);
  const unsigned step_k = (unsigned)(BN * 2) * (unsigned)p.k_rs, step_v = (unsigned)(BN * 2) * (unsigned)p.v_rs;
  const int nmin_s = __builtin_amdgcn_readfirstlane(n_min), nts_s = __builtin_amdgcn_readfirstlane(n_tiles);
  // (PAGED) table entries and in-page tile indices of the tiles the next iteration requests: K tile u + 1 and V tile u (the K tile of the iteration before);
  // pg_nxt = the tile after those, whose entry is requested one iteration ahead
  int pg_blk_k = 0, pg_tin_k = 0, pg_blk_v = 0, pg_tin_v = 0, pg_page_n = 0, pg_tin_n = 0;
  auto pg_enter = [&](int u) __attribute__((always_inline)) {   // before the first active iteration u
    if constexpr (PAGED) {
      const int nv = n_min + u, nk = nv + 1, nn = nv + 2;
      pg_tin_v = nv % pg_tpp; pg_blk_v = page_entry(nv / pg_tpp);
      pg_tin_k = nk % pg_tpp; pg_blk_k = page_entry(nk / pg_tpp);
      pg_page_n = nn / pg_tpp; pg_tin_n = nn - pg_page_n * pg_tpp;
    }
  };
  // pmc: 0 = a full iteration, 1 = the wave's first (its P.V halves and the first step's exp2 work are dead: fast_step MODE 1 + 2), 2 = its last (its score
  // chains are dead: MODE 3 + 4; no mask rewrites either)
  auto step_pair = [&](auto parc, int u, auto pmc) __attribute__((always_inline)) {
    constexpr int par = decltype(parc)::value, PM = decltype(pmc)::value;
    using M0 = ICw<(PM == 1 ? 1 : PM == 2 ? 3 : 0)>; using M1 = ICw<(PM == 1 ? 2 : PM == 2 ? 4 : 0)>;
    // K_{u+1} rides in the first step, V_u in the second.  A tile past the block's last one is requested like any other: past the last key
    // the descriptor's range check zero-fills it, before that it is a real tile nobody looks at (its scores are masked, its V rows meet P = 0)
    // -- one tile of extra traffic per block against two descriptor selects per iteration.  Everything scalar the steps need is made HERE,
    // and made cheap: carried offsets, constant descriptors (37 scalar instructions per iteration in round 2's form of this, ~175 clocks of
    // a 3300-clock iteration).
    q_trickle();
    u32x4 kring[FA_W64_AH + 1];   // K fragment ring of the two steps (the second step's first fragments are requested by the first)
    const int us = __builtin_amdgcn_readfirstlane(u);   // (uniform by construction; said so)
    unsigned dst_k = (unsigned)((par ^ 1) * TILE_BYTES) + wave_dst, dst_v = (unsigned)((2 + par) * TILE_BYTES) + wave_dst;
    // (two scalar multiplies: carried offsets end up in vector registers.  Walking downwards the tile before the first one has a "negative" offset:
    // it wraps to the top of the 32-bit range, outside the descriptor, and is zero-filled like a tile past the last key)
    unsigned tk_ = (unsigned)(nmin_s + (DESC ? nts_s - 2 - us : us + 1)) * step_k, tv_ = (unsigned)(nmin_s + (DESC ? nts_s - 1 - us : us)) * step_v;
    int im32 = ((us - m_lo) | (m_hi - us)) >> 31;   // -1 outside [m_lo, m_hi] (arithmetic, not a compare + select: that one goes through a lane mask)
    dst_k = __builtin_amdgcn_readfirstlane(dst_k); dst_v = __builtin_amdgcn_readfirstlane(dst_v);
    im32 = __builtin_amdgcn_readfirstlane(im32);
    u32x4 srd_k = k_srd, srd_v = v_srd;
    if constexpr (PAGED) {
      // (the carried values are wave-uniform; said so -- assigned under the "active" branch they would live in vector registers and drag the arithmetic there)
      const int bk = __builtin_amdgcn_readfirstlane(pg_blk_k), tk = __builtin_amdgcn_readfirstlane(pg_tin_k);
      const int bv = __builtin_amdgcn_readfirstlane(pg_blk_v), tv = __builtin_amdgcn_readfirstlane(pg_tin_v);
      const int pn = __builtin_amdgcn_readfirstlane(pg_page_n), tn = __builtin_amdgcn_readfirstlane(pg_tin_n);
      srd_k = tile_srd(ICw<0>{}, bk, tk, nmin_s + us + 1);
      srd_v = tile_srd(ICw<1>{}, bv, tv, nmin_s + us);
      tk_ = 0u; tv_ = 0u;
      // next iteration: its V tile is this one's K tile; its K tile is pg_nxt, whose entry has a whole iteration to arrive
      pg_blk_v = bk; pg_tin_v = tk;
      pg_blk_k = page_entry(pn); pg_tin_k = tn;
      const bool wrap = tn + 1 == pg_tpp;
      pg_tin_n = wrap ? 0 : tn + 1;
      pg_page_n = wrap ? pn + 1 : pn;
    }
    asm volatile("" : "+s"(tk_), "+s"(tv_), "+s"(dst_k), "+s"(dst_v), "+s"(im32));
    // (each test on a freshly laundered scalar: as one hoisted boolean hipcc keeps a lane mask and spends five instructions per test)
    auto masked = [&]() __attribute__((always_inline)) { int c = im32; asm volatile("" : "+s"(c)); return c != 0; };
    if constexpr (PM != 2) { if (__builtin_expect(masked(), 0)) set_mask(2 * u); }
    fast_step(ICw<0>{}, parc, sA, sB, pfA, pfB, srd_k, koff_l, tk_, dst_k, kring, DROP ? step_key(2 * us - 1) >> 2 : 0, M0{});
    if constexpr (PM != 2) { if (__builtin_expect(masked(), 0)) set_mask(2 * u + 1); }
    fast_step(ICw<1>{}, parc, sB, sA, pfB, pfA, srd_v, voff_l, tv_, dst_v, kring, DROP ? step_key(2 * us) >> 2 : 0, M1{});
    if constexpr (PM != 2) { if (__builtin_expect(masked(), 0)) clear_mask(2 * u + 2); }
    iter_end();
  };
  auto idle_iter = [&](int u) __attribute__((always_inline)) {
    q_trickle();
    dma_tile(ICw<0>{}, (u & 1) ^ 1, tile_of(u + 1)); dma_tile(ICw<1>{}, u & 1, tile_of(u));
    iter_end();
  };
  if (n_tiles > 0) {
    // idle | active | idle.  The active range starts at an even iteration (one extra, fully masked, iteration for a wave whose window
    // starts at an odd tile: the block runs that iteration anyway) and leaves the two-iteration body in the middle or at its end.
    const bool active = u_first <= u_last;
    const int ua = active ? (u_first & ~1) : n_tiles + 1;
    int u = 0;
#pragma unroll 1
    for (; u < ua; ++u) idle_iter(u);
    if (active) {
      pg_enter(u);
      if constexpr (ALIBI) { if (u > 0) clear_mask(2 * u); }   // (the broadcasts were initialised for step 0)
      if constexpr (FEAT == 0) {
        // (round 5) the first and the last iteration of the wave's range are peeled: u_last > ua whenever the wave is active at all (a key seen in iteration u is
        // multiplied by V one iteration later); ua is even
        step_pair(ICw<0>{}, u, ICw<1>{}); ++u;
#pragma unroll 1
        for (;;) {
          if (u >= u_last) break;
          step_pair(ICw<1>{}, u, ICw<0>{}); ++u;
          if (u >= u_last) break;
          step_pair(ICw<0>{}, u, ICw<0>{}); ++u;
        }
        if (u & 1) step_pair(ICw<1>{}, u, ICw<2>{}); else step_pair(ICw<0>{}, u, ICw<2>{});
        ++u;
      } else {
#pragma unroll 1
        for (;;) {
          step_pair(ICw<0>{}, u, ICw<0>{}); ++u;
          if (u > u_last) break;
          step_pair(ICw<1>{}, u, ICw<0>{}); ++u;
          if (u > u_last) break;
        }
      }
    }
#pragma unroll 1
    for (; u <= n_tiles; ++u) idle_iter(u);
  }

  if (wave_valid) {
  mfma_drain_acc();
  u32x4 o_srd;
  {
    const int rows = min(BM, sq - m0);
    const unsigned long long a = (unsigned long long)(op + (int64_t)m0 * p.o_rs);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi16 = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffffu;
    const unsigned nr = __builtin_amdgcn_readfirstlane((unsigned)(((unsigned long long)(rows - 1) * (unsigned long long)p.o_rs + D) * 2ull));
    o_srd = u32x4{lo, hi16, nr, 0x00020000u};
  }
  // O tile through LDS (the K/V buffers are free after the last tile barrier; the Q region may already hold the next block's
  // rows and is not touched): whole-row stores
  static_for<QB>([&](auto qbc) __attribute__((always_inline)) {
    constexpr int qb = decltype(qbc)::value;
    f32x16 o_v[DB];
    static_for<DB>([&](auto dbc) __attribute__((always_inline)) {
      constexpr int db = decltype(dbc)::value;
      acc_read_tuple<16 * (qb * DB + db)>(o_v[db]);
#pragma unroll
      for (int r = 0; r < 16; ++r) o_v[db][r] = o_v[db][r];
    });
    const float l_tot = half_sum(l_run[qb][0] + l_run[qb][1]);
    const bool dead = (l_tot == 0.f) || (l_tot != l_tot);
    const float inv = (dead ? 1.f : 1.f / l_tot) * o_lag[qb] * (DROP ? p.rp_keep : 1.f);   // normalisation, the pending rescale factor (and 1/(1-p) under dropout) in one multiply per element
    const int row0 = w_row0 + 32 * qb;
    if (row0 < sq) {
      // 32 rows through this wave's staging rows, then whole-row stores through the block's O descriptor: rows past the sequence
      // end are outside it and dropped; one 32-bit lane offset (laundered: as a loop invariant of the persistent loop hipcc would
      // keep sixteen 64-bit store addresses alive across the tile loop and spill them) + a scalar offset per store
      {
        using V4 = typename T::v4;
        constexpr int RS = ROW_BYTES + 16, RPI = 64 / CPR;
        char FA_LDS* stage = lds + (wave * 64 + qb * 32) * RS;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            V4 ov;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) ov[jj] = (E)(o_v[db][4 * g + jj] * inv);
            *reinterpret_cast<V4 FA_LDS*>(stage + qi * RS + (32 * db + 8 * g + 4 * hi) * 2) = ov;
          }
        unsigned lane_off = (unsigned)(((wave * 64 + lane / CPR) * (int)p.o_rs + (lane % CPR) * 8) * 2);
        asm volatile("" : "+v"(lane_off));
#pragma unroll
        for (int i = 0; i < 32 / RPI; ++i) {
          const u32x4 x = *reinterpret_cast<const u32x4 FA_LDS*>(stage + (i * RPI + lane / CPR) * RS + (lane % CPR) * 16);
          const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)((qb * 32 + i * RPI) * (int)p.o_rs * 2));
          asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" : : "v"(x), "v"(lane_off), "s"(o_srd), "s"(soff) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staging rows are rewritten by the next query block
      }
      const int my_row = row0 + qi;
      if (my_row < sq && hi == 0) lsep[my_row] = dead ? INFINITY : (m_run[qb] * kLn2 + __logf(l_tot));
    }
  });
  }  // wave_valid
  }  // persistent block loop
}

template <typename E, int D, int FEAT = 0, bool PAGED = false>
static int launch_fwd_w64_t(const FwdK& p, hipStream_t stream) {
  constexpr int STAGE = 256 * (D * 2 + 16);
  constexpr int smem = ((4 * 64 * D * 2 > STAGE ? 4 * 64 * D * 2 : STAGE + 1023) / 1024 * 1024) + 256 * D * 2;  // K/V buffers | O staging, then the Q block
  auto kern = fa_fwd_w64_kernel<E, D, FEAT, PAGED>;
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
  ls.fwd_kernel = 3; ls.fwd_nw = 4; ls.fwd_feat = FEAT; ls.fwd_splits = 1; ls.fwd_list = p.work_list != nullptr; ls.d = D;
  ls.bf16 = std::is_same<E, __bf16>::value;
  if (FEAT) snprintf(ls.name, sizeof(ls.name), "fa::fa_fwd_w64_kernel<%s,%d,%s>", ls.bf16 ? "bf16" : "f16", D, FEAT == FEAT_CAP ? "softcap" : FEAT == FEAT_DROP ? "dropout" : "alibi");
  else if (PAGED) snprintf(ls.name, sizeof(ls.name), "fa::fa_fwd_w64_kernel<%s,%d,paged>", ls.bf16 ? "bf16" : "f16", D);
  else snprintf(ls.name, sizeof(ls.name), "fa::fa_fwd_w64_kernel<%s,%d>", ls.bf16 ? "bf16" : "f16", D);
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
  if (p.rng) {
    if (d == 128) return launch_fwd_w64_t<_Float16, 128, FEAT_DROP>(p, stream);
    if (d == 64) return launch_fwd_w64_t<_Float16, 64, FEAT_DROP>(p, stream);
    return -2;
  }
  if (p.softcap > 0.f) {
    if (d == 128) return launch_fwd_w64_t<_Float16, 128, FEAT_CAP>(p, stream);
    if (d == 64) return launch_fwd_w64_t<_Float16, 64, FEAT_CAP>(p, stream);
    return -2;
  }
  if (p.alibi) {
    if (d == 128) return launch_fwd_w64_t<_Float16, 128, FEAT_ALIBI>(p, stream);
    if (d == 64) return launch_fwd_w64_t<_Float16, 64, FEAT_ALIBI>(p, stream);
    return -2;
  }
  if (p.block_table) {
    if (d == 128) return launch_fwd_w64_t<_Float16, 128, 0, true>(p, stream);
    if (d == 64) return launch_fwd_w64_t<_Float16, 64, 0, true>(p, stream);
    return -2;
  }
  if (d == 128) return launch_fwd_w64_t<_Float16, 128>(p, stream);
  if (d == 64) return launch_fwd_w64_t<_Float16, 64>(p, stream);
  return -2;
}
#endif
#if FA_W64_PART != 2
int launch_fwd_w64_bf16(const FwdK& p, int d, hipStream_t stream) {
  if (p.rng) {
    if (d == 128) return launch_fwd_w64_t<__bf16, 128, FEAT_DROP>(p, stream);
    if (d == 64) return launch_fwd_w64_t<__bf16, 64, FEAT_DROP>(p, stream);
    return -2;
  }
  if (p.softcap > 0.f) {
    if (d == 128) return launch_fwd_w64_t<__bf16, 128, FEAT_CAP>(p, stream);
    if (d == 64) return launch_fwd_w64_t<__bf16, 64, FEAT_CAP>(p, stream);
    return -2;
  }
  if (p.alibi) {
    if (d == 128) return launch_fwd_w64_t<__bf16, 128, FEAT_ALIBI>(p, stream);
    if (d == 64) return launch_fwd_w64_t<__bf16, 64, FEAT_ALIBI>(p, stream);
    return -2;
  }
  if (p.block_table) {
    if (d == 128) return launch_fwd_w64_t<__bf16, 128, 0, true>(p, stream);
    if (d == 64) return launch_fwd_w64_t<__bf16, 64, 0, true>(p, stream);
    return -2;
  }
  if (d == 128) return launch_fwd_w64_t<__bf16, 128>(p, stream);
  if (d == 64) return launch_fwd_w64_t<__bf16, 64>(p, stream);
  return -2;
}
// 4 waves x 64 query rows per workgroup.  Plain attention (contiguous or paged keys / values), softcap, dropout, or ALiBi under a causal right bound (no split
// keys, one feature at a time, features only on contiguous keys / values).
int launch_fwd_w64(const FwdK& p, int dtype_bf16, int d, hipStream_t stream) {
  if (p.n_splits > 1) return -2;
  if (p.rng != nullptr && (p.randval != nullptr || p.softcap > 0.f || p.alibi != nullptr || p.block_table != nullptr)) return -2;   // dropout: alone, and without the random-byte output
  if (p.softcap > 0.f && p.alibi != nullptr) return -2;
  if (p.block_table != nullptr && (p.softcap > 0.f || p.alibi != nullptr || p.leftpad_k != nullptr || p.kv_batch_idx != nullptr || p.page_size % 64 != 0)) return -2;
  // (paged: a page's and a tile's bytes are formed in 32 bits -- pg_bytes_* / tl_bytes_* in the kernel)
  if (p.block_table != nullptr && ((uint64_t)(p.k_bs > p.v_bs ? p.k_bs : p.v_bs) * 2u >= (1ull << 32) || (uint64_t)p.page_size * (uint64_t)(p.k_rs > p.v_rs ? p.k_rs : p.v_rs) * 2u >= (1ull << 32))) return -2;
  if (p.alibi != nullptr && p.wr != 0) return -2;   // the bias is linear in the key only where no visible key lies right of the diagonal
  // buffer addressing: 32-bit byte offsets from the (batch, kv-head) base
  // (a paged cache is addressed tile by tile: the offsets only span 64 rows)
  const uint64_t span = ((uint64_t)(p.block_table ? 64 : (p.sk > 0 ? p.sk : 1)) + 128) * (uint64_t)(p.k_rs > p.v_rs ? p.k_rs : p.v_rs) * 2u;
  if (span >= (1ull << 32)) return -3;
  if (256ull * (uint64_t)(p.q_rs > p.o_rs ? p.q_rs : p.o_rs) * 2u >= (1ull << 32)) return -3;   // Q / O: 32-bit byte offsets over a block's 256 rows
  return dtype_bf16 ? launch_fwd_w64_bf16(p, d, stream) : launch_fwd_w64_f16(p, d, stream);
}
#endif

}  // namespace fa
