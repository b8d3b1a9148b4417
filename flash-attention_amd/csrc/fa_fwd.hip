// Fused attention forward for gfx950 (MI355X): S^T = K.Q^T -> online softmax -> O^T += V^T.P^T.
//
// Replaces the reference's flash_fwd_kernel / compute_attn_1rowblock
// (csrc/flash_attn/src/flash_fwd_kernel.h:54-501, softmax.h:128-187, mask.h:111-212,
// block_info.h:12-45) with a CDNA4-native structure -- it is not a translation of the CuTe code:
//
//   * one workgroup = NW waves, each wave owns 32 query rows for the whole key loop
//     (BM = 32*NW rows per workgroup); K/V tiles of BN = 64 keys are staged once per
//     workgroup through LDS (register-staged, double buffered, one barrier per tile);
//   * both contractions run "swapped" on v_mfma_f32_32x32x16: S^T[key][query] and
//     O^T[d][query].  In the accumulator layout column = lane&31, so EVERY per-query quantity
//     (running max m, running sum l, rescale factor, LSE) is lane-local: the row max is a
//     31-op in-register reduction plus one exchange with lane^32, and the O rescale is a
//     per-lane scalar multiply;
//   * P^T needs no shuffle to become the next MFMA's B operand: the k-index permutation the
//     accumulator layout induces on the keys is applied to V instead, by choosing which 4-key
//     groups each ds_read_b64_tr_b16 (LDS transpose read) fetches;
//   * K tile rows are XOR-swizzled at 16-B granularity (conflict-free ds_read_b128 A operands),
//     V tile rows at 64-B granularity (conflict-free transpose reads);
//   * causal / sliding-window / key-length masks are evaluated from per-lane key limits only on
//     tiles that straddle a boundary; tiles entirely outside a wave's visible range are skipped
//     per wave, and the block's key range is clipped (reference flash_fwd_kernel.h:90-94);
//   * 1-D grid with an XCD-aware remap so all query blocks of one (batch, head) -- and the
//     query heads sharing a KV head -- run on one XCD and share its L2; heavy (long-key-range)
//     blocks are scheduled first when the mask is right-bounded.
#include "fa_device.h"
#include "fa_kernel_params.h"
#include "fa_launch.h"

namespace fa {

template <int D> FA_DEVINL int k_swz(int row) { return D == 128 ? (row & 15) : ((row >> 1) & 7); }
template <int D> FA_DEVINL int v_swz(int row) { return D == 128 ? (row & 3) : ((row >> 1) & 1); }

template <typename E, int D, int NW>
__global__ void __launch_bounds__(NW * 64, 2) fa_fwd_kernel(const FwdK p) {
  using T = ElemTraits<E>;
  using V8 = typename T::v8;
  using V4 = typename T::v4;
  constexpr int BM = NW * 32, BN = 64, CPR = D / 8, NT = NW * 64;
  constexpr int ROW_BYTES = D * 2;
  constexpr int TILE_BYTES = BN * ROW_BYTES;
  constexpr int LD = (BN * CPR) / NT;  // 16-B chunks each thread moves per tile (K and V each)
  constexpr int KS = D / 16;           // k-steps of the QK^T contraction
  constexpr int DB = D / 32;           // 32-wide d blocks of the output
  static_assert(D == 64 || D == 128, "head dims built natively: 64, 128");
  static_assert(LD >= 1 && (BN * CPR) % NT == 0, "tile does not divide over the workgroup");
  constexpr float kLn2 = 0.6931471805599453f, kLog2e = 1.4426950408889634f;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char FA_LDS* lds = (char FA_LDS*)smem;  // K0 | K1 | V0 | V1

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, qi = lane & 31;

  // ---- which (batch, head, query block) -------------------------------------------------------
  const int total = p.nmb * p.b * p.h;
  const int w = xcd_remap(blockIdx.x, total);
  const int bh = w / p.nmb;
  const int mbr = w - bh * p.nmb;
  const int m_block = (p.wr >= 0) ? (p.nmb - 1 - mbr) : mbr;
  const int b = bh / p.h;
  const int h = bh - b * p.h;
  const int hk = h / p.hk_ratio;

  int sq = p.sq, sk = p.sk;
  int64_t q_row0 = 0, k_row0 = 0;  // first row of this sequence in the packed tensors
  int64_t q_boff = (int64_t)b * p.q_bs, k_boff = (int64_t)b * p.k_bs, v_boff = (int64_t)b * p.v_bs, o_boff = (int64_t)b * p.o_bs;
  if (p.cu_q) {  // varlen: rows cu[b] .. cu[b+1]-1  (reference block_info.h:17-36)
    const int c0 = p.cu_q[b];
    sq = p.cu_q[b + 1] - c0;
    q_row0 = c0;
    q_boff = 0;
    o_boff = 0;
  }
  if (p.cu_k) {
    const int c0 = p.cu_k[b];
    sk = p.cu_k[b + 1] - c0;
    k_row0 = c0;
    k_boff = 0;
    v_boff = 0;
  }
  if (p.seqused_k) sk = p.seqused_k[b];
  const int m0 = m_block * BM;
  if (m0 >= sq) return;

  const E* __restrict__ qp = (const E*)p.q + q_boff + q_row0 * p.q_rs + (int64_t)h * p.q_hs;
  const E* __restrict__ kp = (const E*)p.k + k_boff + k_row0 * p.k_rs + (int64_t)hk * p.k_hs;
  const E* __restrict__ vp = (const E*)p.v + v_boff + k_row0 * p.v_rs + (int64_t)hk * p.v_hs;
  E* __restrict__ op = (E*)p.o + o_boff + q_row0 * p.o_rs + (int64_t)h * p.o_hs;
  float* __restrict__ lsep = p.cu_q ? (p.lse + (int64_t)h * p.total_q + q_row0)
                                    : (p.lse + ((int64_t)b * p.h + h) * p.sq);

  // ---- key range of the block, per-wave and per-lane visibility limits --------------------------
  const int shift = sk - sq;  // bottom-right alignment
  const int blk_last = min(m0 + BM, sq) - 1;
  int kmax = sk - 1, kmin = 0;
  if (p.wr >= 0) kmax = min(kmax, blk_last + shift + p.wr);
  if (p.wl >= 0) kmin = max(0, m0 + shift - p.wl);
  const int n_min = kmin / BN;
  const int n_max = (kmax >= kmin) ? (kmax / BN + 1) : n_min;

  const int w_row0 = m0 + wave * 32;
  const int w_row1 = min(w_row0 + 31, sq - 1);
  const bool wave_valid = w_row0 < sq;
  const int w_kmax = (p.wr >= 0) ? min(sk - 1, w_row1 + shift + p.wr) : sk - 1;   // last key any row sees
  const int w_kmin = (p.wl >= 0) ? max(0, w_row0 + shift - p.wl) : 0;             // first key any row sees
  const int w_full_hi = (p.wr >= 0) ? min(sk - 1, w_row0 + shift + p.wr) : sk - 1;  // keys <= this: visible to all rows
  const int w_full_lo = (p.wl >= 0) ? (w_row1 + shift - p.wl) : 0;                  // keys >= this: visible to all rows

  const int my_row = w_row0 + qi;
  const bool row_valid = my_row < sq;
  const int lim_hi = (p.wr >= 0) ? min(sk - 1, my_row + shift + p.wr) : sk - 1;
  const int lim_lo = (p.wl >= 0) ? (my_row + shift - p.wl) : 0;

  const bool transform = (p.softcap > 0.f) || (p.alibi != nullptr);
  const float cs = transform ? kLog2e : p.scale_log2;  // multiplier taking S to the log2 domain
  const float slope = p.alibi ? p.alibi[(int64_t)b * p.alibi_bs + h] : 0.f;

  // ---- Q fragments (B operand of S^T = K.Q^T): lane = query row, 8 consecutive d per k-step -----
  V8 qf[KS];
  {
    const E* qrow = qp + (int64_t)my_row * p.q_rs + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = bitcast_u32x4<V8>(ld_global_16B(qrow + 16 * ks, row_valid));
  }

  // ---- staging: global -> registers -> LDS -------------------------------------------------------
  u32x4 kreg[LD], vreg[LD];
  auto load_tile = [&](int n) {
#pragma unroll
    for (int i = 0; i < LD; ++i) {
      const int idx = tid + i * NT;
      const int row = idx / CPR, ch = idx % CPR;
      const int key = n * BN + row;
      const bool ok = key < sk;
      kreg[i] = ld_global_16B(kp + (int64_t)key * p.k_rs + ch * 8, ok);
      vreg[i] = ld_global_16B(vp + (int64_t)key * p.v_rs + ch * 8, ok);
    }
  };
  auto store_tile = [&](int buf) {
    char FA_LDS* kb_ = lds + buf * TILE_BYTES;
    char FA_LDS* vb_ = lds + (2 + buf) * TILE_BYTES;
#pragma unroll
    for (int i = 0; i < LD; ++i) {
      const int idx = tid + i * NT;
      const int row = idx / CPR, ch = idx % CPR;
      *(u32x4 FA_LDS*)(kb_ + row * ROW_BYTES + ((ch ^ k_swz<D>(row)) << 4)) = kreg[i];
      *(u32x4 FA_LDS*)(vb_ + row * ROW_BYTES + (((((ch >> 2) ^ v_swz<D>(row)) << 2) | (ch & 3)) << 4)) = vreg[i];
    }
  };

  // per-lane LDS read offsets
  const int kread_base = qi * ROW_BYTES;          // + 32*kb rows, chunk (2ks+hi) ^ kswz
  const int kswz = k_swz<D>(qi);
  const int tr_i = lane & 15, tr_half = (lane >> 4) & 1;
  const int tr_rr = tr_i >> 2, tr_cc = tr_i & 3;
  const int vswz = v_swz<D>(tr_rr);
  int vread_base[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
    vread_base[db] = (4 * hi + tr_rr) * ROW_BYTES + ((db ^ vswz) << 6) + tr_half * 32 + tr_cc * 8;

  // ---- online-softmax state (per lane = per query row; both half-waves keep identical m) ----------
  f32x16 o_acc[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[db][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  if (n_min < n_max) {
    load_tile(n_min);
    store_tile(0);
    __syncthreads();
  }

  for (int n = n_min; n < n_max; ++n) {
    const int cur = (n - n_min) & 1;
    const int kv0 = n * BN;
    const bool has_next = (n + 1 < n_max);
    if (has_next) load_tile(n + 1);  // lands while this tile is being computed

    const bool active = wave_valid && (kv0 <= w_kmax) && (kv0 + BN - 1 >= w_kmin);
    if (active) {
      const char FA_LDS* kbuf = lds + cur * TILE_BYTES;
      const char FA_LDS* vbuf = lds + (2 + cur) * TILE_BYTES;

      // S^T[key][query] for the two 32-key halves of the tile
      f32x16 s[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const u32x4 kraw = *(const u32x4 FA_LDS*)(kbuf + kread_base + kb * 32 * ROW_BYTES + (((2 * ks + hi) ^ kswz) << 4));
          s[kb] = T::mfma(bitcast_u32x4<V8>(kraw), qf[ks], s[kb]);
        }
      }

      if (transform) {  // softcap / ALiBi: move to the scaled domain first (reference utils.h:395-409, alibi.h)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float y = s[kb][r] * p.scale;
            if (p.softcap > 0.f) y = p.softcap * tanhf(y / p.softcap);
            if (p.alibi) {
              const int key = kv0 + 32 * kb + acc_row(r, hi);
              y -= slope * fabsf((float)(my_row + shift - key));
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
      tmax = fmaxf(tmax, xchg_half(tmax));

      const float m_new = fmaxf(m_run, tmax);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;  // fully masked so far (softmax.h:76,154-156)
      const float alpha = fast_exp2((m_run - m_use) * cs);
      const float neg_mc = -m_use * cs;
      m_run = m_new;

      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = fast_exp2(__builtin_fmaf(s[kb][r], cs, neg_mc));
          s[kb][r] = pv;
          psum += pv;
        }
      l_run = l_run * alpha + psum;

      if (!__all(alpha == 1.f)) {
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o_acc[db][r] *= alpha;
      }

      // P^T as B operand: k-step (kb,t) <-> accumulator registers 8t..8t+7 of s[kb]
      V8 pf[4];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[kb * 2 + t][j] = (E)s[kb][8 * t + j];

      // O^T[d][query] += V^T[d][key] . P^T[key][query]
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          const char FA_LDS* a0 = vbuf + vread_base[db] + (16 * kt) * ROW_BYTES;
          const s16x4 lo = lds_read_tr16(a0);
          const s16x4 hi4 = lds_read_tr16(a0 + 8 * ROW_BYTES);
          o_acc[db] = T::mfma(combine_tr<V8>(lo, hi4), pf[kt], o_acc[db]);
        }
      }
    }

    if (has_next) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: normalise, store O (bf16/fp16) and LSE -------------------------------------------
  if (!wave_valid) return;
  const float l_tot = l_run + xchg_half(l_run);
  const bool dead = (l_tot == 0.f) || (l_tot != l_tot);  // no visible key (softmax.h:179-180)
  const float inv = dead ? 1.f : 1.f / l_tot;
  if (row_valid) {
    E* orow = op + (int64_t)my_row * p.o_rs;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        V4 ov;
#pragma unroll
        for (int j = 0; j < 4; ++j) ov[j] = (E)(o_acc[db][4 * g + j] * inv);
        *reinterpret_cast<V4*>(orow + 32 * db + 8 * g + 4 * hi) = ov;
      }
    if (hi == 0) lsep[my_row] = dead ? INFINITY : (m_run * cs * kLn2 + __logf(l_tot));
  }
}

template <typename E, int D, int NW>
static int launch_fwd_t(const FwdK& p, hipStream_t stream) {
  constexpr int smem = 4 * 64 * D * 2;
  auto kern = fa_fwd_kernel<E, D, NW>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return -1;
    attr_done = true;
  }
  const long long total = (long long)p.nmb * p.b * p.h;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(NW * 64), smem, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int fwd_block_m(int nw) { return 32 * nw; }

int launch_fwd(const FwdK& p, int dtype_bf16, int d, int nw, hipStream_t stream) {
#define FA_FWD_CASE(E_, D_, NW_) \
  if (d == D_ && nw == NW_) return launch_fwd_t<E_, D_, NW_>(p, stream);
  if (dtype_bf16) {
    FA_FWD_CASE(__bf16, 128, 8) FA_FWD_CASE(__bf16, 128, 4) FA_FWD_CASE(__bf16, 64, 8) FA_FWD_CASE(__bf16, 64, 4)
  } else {
    FA_FWD_CASE(_Float16, 128, 8) FA_FWD_CASE(_Float16, 128, 4) FA_FWD_CASE(_Float16, 64, 8) FA_FWD_CASE(_Float16, 64, 4)
  }
#undef FA_FWD_CASE
  return -2;
}

}  // namespace fa
