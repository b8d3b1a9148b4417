// Forward attention, "wide-wave" software-pipelined schedule for gfx950: one wave per SIMD, 64 query rows per
// wave (two 32-row blocks), the whole 512-entry register file per wave.
//
// Same arithmetic, LDS layouts and step structure as fa_fwd_il.hip.  Why another shape: timing ablations of the
// 32-rows-per-wave kernels (tools/ablate_fwd.sh) show the steady state is bound by instruction issue, not by the
// matrix pipe -- removing half the MFMAs saves only 25 % -- and per MFMA those kernels issue ~12 instructions
// (2 LDS operand reads, ~6 VALU, waits, scalar bookkeeping).  With two row blocks per wave every K / V fragment
// read from LDS feeds two MFMAs, the Q fragments of both blocks fit in registers (64 VGPRs), and the per-step
// bookkeeping is shared: ~6.5 instructions per MFMA.  A lone wave per SIMD has nobody to hide its stalls, so the
// steady-state step is hand-placed: per slot one LDS operand read four slots ahead, two MFMAs, and that slot's
// share (<= 5 per MFMA) of the softmax VALU work.
//
// Workgroup = 4 waves x 64 rows = 256 query rows, LDS = K/V double buffers (64 KB), 1 workgroup per CU.
#include <cstdlib>
#include <type_traits>

#include "fa_device.h"
#include "fa_kernel_params.h"
#include "fa_launch.h"

namespace fa {

template <int D> FA_DEVINL constexpr int k_swz_w(int row) { return D == 128 ? (row & 15) : ((row >> 1) & 7); }
template <int D> FA_DEVINL constexpr int v_swz_w(int row) { return D == 128 ? (row & 3) : ((row >> 1) & 1); }
template <int N> using ICw = std::integral_constant<int, N>;

template <typename E, int D>
__global__ void __launch_bounds__(256, 1) fa_fwd_w64_kernel(const FwdK p) {
  using T = ElemTraits<E>;
  using V8 = typename T::v8;
  using V4 = typename T::v4;
  constexpr int NW = 4, RB = 2, BM = NW * 32 * RB, BN = 64, CPR = D / 8;
  constexpr int ROW_BYTES = D * 2;
  constexpr int TILE_BYTES = BN * ROW_BYTES;
  constexpr int KS = D / 16;
  constexpr int DB = D / 32;
  constexpr float kLn2 = 0.6931471805599453f;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, qi = lane & 31;

  const int w = xcd_interleave(blockIdx.x, p.n_units, p.unit_size, p.unit_hpx);
  if (w < 0) return;
  const int bh = w / p.nmb;
  const int mbr = w - bh * p.nmb;
  const int m_block = (p.wr >= 0) ? (p.nmb - 1 - mbr) : mbr;
  const int b = bh / p.h;
  const int h = bh - b * p.h;
  const int hk = h / p.hk_ratio;

  int sq = p.sq, sk = p.sk;
  int64_t q_row0 = 0, k_row0 = 0;
  int64_t q_boff = (int64_t)b * p.q_bs, k_boff = (int64_t)b * p.k_bs, v_boff = (int64_t)b * p.v_bs, o_boff = (int64_t)b * p.o_bs;
  if (p.cu_q) { const int c0 = p.cu_q[b]; sq = p.cu_q[b + 1] - c0; q_row0 = c0; q_boff = 0; o_boff = 0; }
  if (p.cu_k) { const int c0 = p.cu_k[b]; sk = p.cu_k[b + 1] - c0; k_row0 = c0; k_boff = 0; v_boff = 0; }
  if (p.seqused_k) sk = p.seqused_k[b];
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
  const int key_base = n_min * BN;

  // the wave's 64 rows; per-lane limits for each of its two row blocks
  const int w_row0 = m0 + wave * 32 * RB;
  const int w_row1 = min(w_row0 + 32 * RB - 1, sq - 1);
  const bool wave_valid = w_row0 < sq;
  const int w_kmax = (p.wr >= 0) ? min(sk - 1, w_row1 + shift + p.wr) : sk - 1;
  const int w_kmin = (p.wl >= 0) ? max(0, w_row0 + shift - p.wl) : 0;
  const int w_full_hi = (p.wr >= 0) ? min(sk - 1, w_row0 + shift + p.wr) : sk - 1;
  const int w_full_lo = (p.wl >= 0) ? (w_row1 + shift - p.wl) : 0;
  int my_row[RB], lim_hi[RB], lim_lo[RB];
  bool row_valid[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    my_row[rb] = w_row0 + 32 * rb + qi;
    row_valid[rb] = my_row[rb] < sq;
    lim_hi[rb] = (p.wr >= 0) ? min(sk - 1, my_row[rb] + shift + p.wr) : sk - 1;
    lim_lo[rb] = (p.wl >= 0) ? (my_row[rb] + shift - p.wl) : 0;
  }
  const float cs = p.scale_log2;
  const float thr = p.rescale_thr;

  auto step_active = [&](int i) __attribute__((always_inline)) {
    const int k0 = key_base + 32 * i;
    return wave_valid && (i >= 0) && (i < n_steps) && (k0 <= w_kmax) && (k0 + 31 >= w_kmin);
  };
  auto step_needs_mask = [&](int i) __attribute__((always_inline)) {
    const int k0 = key_base + 32 * i;
    return (k0 + 31 > w_full_hi) || (k0 < w_full_lo);
  };

  // ---- K/V tiles by LDS-DMA (see fa_fwd_il.hip); LDS is addressed by byte offset (dynamic segment starts at 0)
  constexpr int RPD = 1024 / ROW_BYTES, NDMA = TILE_BYTES / 1024, DPW = NDMA / NW;
  static_assert(NDMA % NW == 0 && DPW >= 1, "tile does not divide over the waves");
  const int d_row = lane / CPR, d_pc = lane % CPR;
  unsigned koff_l[DPW], voff_l[DPW];
#pragma unroll
  for (int i = 0; i < DPW; ++i) {
    const int row = (wave * DPW + i) * RPD + d_row;
    const int kc = d_pc ^ k_swz_w<D>(row);
    const int vc = ((((d_pc >> 2) ^ v_swz_w<D>(row)) << 2) | (d_pc & 3));
    koff_l[i] = (unsigned)(row * (int)p.k_rs + kc * 8) * 2u;
    voff_l[i] = (unsigned)(row * (int)p.v_rs + vc * 8) * 2u;
  }
  auto dma_tile = [&](auto isvc, int buf, int t) __attribute__((always_inline)) {
    constexpr bool ISV = decltype(isvc)::value != 0;
    const int n = n_min + t;
    const int64_t rs = ISV ? p.v_rs : p.k_rs;
    const char* base = (const char*)((ISV ? vp : kp) + (int64_t)n * BN * rs);
    const char FA_LDS* dst = (const char FA_LDS*)(unsigned long)(unsigned)((ISV ? 2 + buf : buf) * TILE_BYTES + wave * DPW * 1024);
    if (n * BN + BN <= sk) {
#pragma unroll
      for (int i = 0; i < DPW; ++i) lds_dma_16B(base + (ISV ? voff_l[i] : koff_l[i]), dst + i * 1024);
    } else {
#pragma unroll
      for (int i = 0; i < DPW; ++i) {
        const int row = (wave * DPW + i) * RPD + d_row;
        const int grow = min(n * BN + row, sk - 1) - n * BN;
        const int c = ISV ? ((((d_pc >> 2) ^ v_swz_w<D>(row)) << 2) | (d_pc & 3)) : (d_pc ^ k_swz_w<D>(row));
        lds_dma_16B(base + ((int64_t)grow * rs + c * 8) * 2, dst + i * 1024);
      }
    }
  };

  // Q fragments of both row blocks (B operands of S^T = K.Q^T)
  V8 qreg[RB][KS];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const E* qrow = qp + (int64_t)my_row[rb] * p.q_rs + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qreg[rb][ks] = bitcast_u32x4<V8>(ld_global_16B(qrow + 16 * ks, row_valid[rb]));
  }

  const int kbase = qi * ROW_BYTES + ((hi ^ k_swz_w<D>(qi)) << 4);
  const int tr_i = lane & 15, tr_half = (lane >> 4) & 1;
  const int tr_rr = tr_i >> 2, tr_cc = tr_i & 3;
  const int vbase = (4 * hi + tr_rr) * ROW_BYTES + (v_swz_w<D>(tr_rr) << 6) + tr_half * 32 + tr_cc * 8;

  f32x16 o_acc[RB][DB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o_acc[rb][db][r] = 0.f;
  float m_run[RB], l_run[RB];
  f32x16 sA[RB], sB[RB];
  V8 pfA[RB][2], pfB[RB][2];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    m_run[rb] = -INFINITY;
    l_run[rb] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sA[rb][r] = 0.f; sB[rb][r] = 0.f; }
  }
  bool have_cur = false, have_prev = false;

  auto apply_mask = [&](f32x16 (&s)[RB], int i) __attribute__((always_inline)) {
    const int k0 = key_base + 32 * i;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int rel_hi = lim_hi[rb] - k0 - 4 * hi, rel_lo = lim_lo[rb] - k0 - 4 * hi;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int off = acc_row(r, 0);
        s[rb][r] = ((off <= rel_hi) && (off >= rel_lo)) ? s[rb][r] : -INFINITY;
      }
    }
  };
  // rescale everything still at the old scale exactly once (O, l, pending fp32 P of this step)
  auto rescale = [&](int rb, bool grow, float m_new, f32x16& pend, bool do_pend) __attribute__((always_inline)) {
    const float m_upd = grow ? m_new : m_run[rb];
    const float m_safe = (m_upd == -INFINITY) ? 0.f : m_upd;
    const float alpha = grow ? fast_exp2((m_run[rb] - m_safe) * cs) : 1.f;
    m_run[rb] = m_upd;
    l_run[rb] *= alpha;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o_acc[rb][db][r] *= alpha;
    if (do_pend) {
#pragma unroll
      for (int r = 0; r < 16; ++r) pend[r] *= alpha;
    }
  };
  auto finish_step = [&](f32x16 (&s_cur)[RB], const float (&tmax)[RB], V8 (&pf_cur)[RB][2], bool do_decide, bool do_sm)
      __attribute__((always_inline)) {
    if (do_decide) {
      float m_new[RB];
      bool grow[RB];
      bool any = false;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const float t = half_max(tmax[rb]);
        m_new[rb] = fmaxf(m_run[rb], t);
        grow[rb] = (m_new[rb] - m_run[rb]) * cs > thr;
        any = any || grow[rb];
      }
      if (__any(any)) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) rescale(rb, grow[rb], m_new[rb], s_cur[rb], do_sm);
      }
    }
    if (do_sm) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) pf_cur[rb][t][jj] = (E)s_cur[rb][8 * t + jj];
    }
  };

  // One 32-key step for both row blocks.  FAST: all three strands present (hand-placed slots); otherwise generic.
  auto step = [&](auto fastc, auto maskc, auto halfc, int kb_lane, int vb_lane, int i, f32x16 (&s_cur)[RB], f32x16 (&s_nxt)[RB],
                  const V8 (&pf_prev)[RB][2], V8 (&pf_cur)[RB][2]) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fastc)::value != 0;
    constexpr bool MASK = decltype(maskc)::value != 0;
    constexpr int HOFF = decltype(halfc)::value * 32 * ROW_BYTES;
    constexpr int NOP = 2 * DB, EPG = 16 / KS, AHEAD = 4, RING = AHEAD + 1;
    const bool do_qk = FAST || step_active(i + 1);
    const bool do_sm = FAST || have_cur;
    const bool do_pv = FAST || have_prev;
    u32x4 kfr[RING];
    s16x4 vlo[RING], vhi[RING];
    auto rd_slot = [&](int slot) __attribute__((always_inline)) {
      if (slot < KS) {
        if (do_qk) kfr[slot % RING] = *(const u32x4 FA_LDS*)(unsigned long)(unsigned)((kb_lane ^ (slot << 5)) + HOFF);
      } else if (slot < KS + NOP) {
        const int op_ = slot - KS;
        if (do_pv) {
          const char FA_LDS* a0 = (const char FA_LDS*)(unsigned long)(unsigned)((vb_lane ^ ((op_ % DB) << 6)) + HOFF + (16 * (op_ / DB)) * ROW_BYTES);
          vlo[slot % RING] = lds_read_tr16(a0);
          vhi[slot % RING] = lds_read_tr16(a0 + 8 * ROW_BYTES);
        }
      }
    };
    float neg_mc[RB], ps0[RB], ps1[RB], tmax[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      neg_mc[rb] = (m_run[rb] == -INFINITY) ? 0.f : -m_run[rb] * cs;
      ps0[rb] = 0.f;
      ps1[rb] = 0.f;
      tmax[rb] = -INFINITY;
    }
#pragma unroll
    for (int g = 0; g < AHEAD; ++g) rd_slot(g);
    if constexpr (FAST) __builtin_amdgcn_sched_barrier(0);
    // slots 0 .. KS-1: S_{i+1} += K.Q^T (both row blocks share the K fragment) + exp / row sums of S_i
#pragma unroll
    for (int g = 0; g < KS; ++g) {
      rd_slot(g + AHEAD);
      if (do_qk) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          f32x16 c = s_nxt[rb];
          if (g == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) c[r] = 0.f;
          }
          s_nxt[rb] = T::mfma(bitcast_u32x4<V8>(kfr[g % RING]), qreg[rb][g], c);
        }
      }
      if (do_sm) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int e = 0; e < EPG; e += 2) {
            const int r = g * EPG + e;
            const float p0 = fast_exp2(__builtin_fmaf(s_cur[rb][r], cs, neg_mc[rb]));
            const float p1 = fast_exp2(__builtin_fmaf(s_cur[rb][r + 1], cs, neg_mc[rb]));
            s_cur[rb][r] = p0;
            s_cur[rb][r + 1] = p1;
            ps0[rb] += p0;
            ps1[rb] += p1;
          }
      }
      if constexpr (FAST) __builtin_amdgcn_sched_barrier(0);
    }
    if (do_sm) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) l_run[rb] += ps0[rb] + ps1[rb];
    }
    if (!FAST && do_qk && step_needs_mask(i + 1)) apply_mask(s_nxt, i + 1);
    int rel_hi[RB], rel_lo[RB];
    if constexpr (FAST && MASK) {
      const int k0 = key_base + 32 * (i + 1);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) { rel_hi[rb] = lim_hi[rb] - k0 - 4 * hi; rel_lo[rb] = lim_lo[rb] - k0 - 4 * hi; }
    }
    // slots KS .. KS+2DB-1: O += V^T.P_{i-1} (both row blocks share the V fragment) + mask / row-max tree of S_{i+1}
#pragma unroll
    for (int g = 0; g < NOP; ++g) {
      rd_slot(KS + g + AHEAD);
      if (do_pv) {
        const V8 vf = combine_tr<V8>(vlo[(KS + g) % RING], vhi[(KS + g) % RING]);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) o_acc[rb][g % DB] = T::mfma(vf, pf_prev[rb][g / DB], o_acc[rb][g % DB]);
      }
      if (do_qk) {
        if constexpr (FAST && MASK) {
          if (g >= 1 && g < 3) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
              for (int r = (g - 1) * 8; r < (g - 1) * 8 + 8; ++r) {
                const int off = acc_row(r, 0);
                s_nxt[rb][r] = ((off <= rel_hi[rb]) && (off >= rel_lo[rb])) ? s_nxt[rb][r] : -INFINITY;
              }
          }
        }
        constexpr int G0 = (FAST && MASK) ? 3 : 2;
        if (g >= G0) {
          constexpr int SLOTS = NOP - G0;
          constexpr int PER = (8 + SLOTS - 1) / SLOTS;
#pragma unroll
          for (int t = 0; t < PER; ++t) {
            const int q_ = (g - G0) * PER + t;
            if (q_ < 8) {
#pragma unroll
              for (int rb = 0; rb < RB; ++rb) tmax[rb] = fmaxf(fmaxf(tmax[rb], s_nxt[rb][2 * q_]), s_nxt[rb][2 * q_ + 1]);
            }
          }
        }
      }
      if constexpr (FAST) __builtin_amdgcn_sched_barrier(0);
    }
    finish_step(s_cur, tmax, pf_cur, do_qk, do_sm);
    have_prev = do_sm;
    have_cur = do_qk;
  };

  if (n_tiles > 0) dma_tile(ICw<0>{}, 0, 0);
  lds_dma_wait_all();
  __syncthreads();

  int uf_lo = 1, uf_hi = 0;
  if (wave_valid && n_tiles > 0) {
    const int a_lo = max(0, (w_kmin - key_base) >> 5);
    const int a_hi = min(n_steps - 1, (w_kmax - key_base) >> 5);
    uf_lo = (a_lo + 3) >> 1;
    uf_hi = (a_hi - 1) >> 1;
  }
  auto iteration = [&](auto fastc, auto maskc, int u) __attribute__((always_inline)) {
    const int par = u & 1;
    if (u + 1 < n_tiles) dma_tile(ICw<0>{}, par ^ 1, u + 1);
    if (u < n_tiles) dma_tile(ICw<1>{}, par, u);
    const int kb_lane = kbase ^ (par * TILE_BYTES);
    const int vb_lane = vbase ^ ((2 + (par ^ 1)) * TILE_BYTES);
    // step 2u-1: S_{2u} from the first half of K_u, PV of step 2u-2 from the first half of V_{u-1}
    step(fastc, maskc, ICw<0>{}, kb_lane, vb_lane, 2 * u - 1, sA, sB, pfA, pfB);
    // step 2u: S_{2u+1} from the second half of K_u, PV of step 2u-1 from the second half of V_{u-1}
    step(fastc, maskc, ICw<1>{}, kb_lane, vb_lane, 2 * u, sB, sA, pfB, pfA);
    lds_dma_wait_all();
    __syncthreads();
  };
  if (n_tiles > 0) {
    int u = 0;
    const int head_end = min(max(uf_lo, 0), n_tiles + 1);
    for (; u < head_end; ++u) iteration(ICw<0>{}, ICw<0>{}, u);
    int um_lo, um_hi;
    {
      const int f_lo = (w_full_lo - key_base + 31) >> 5;
      const int f_hi = (w_full_hi - 31 - key_base) >> 5;
      um_lo = max(uf_lo, (f_lo + 1) >> 1);
      um_hi = min(uf_hi, (f_hi - 1) >> 1);
    }
    for (; u <= uf_hi && u < um_lo; ++u) iteration(ICw<1>{}, ICw<1>{}, u);
    for (; u <= um_hi; ++u) iteration(ICw<1>{}, ICw<0>{}, u);
    for (; u <= uf_hi; ++u) iteration(ICw<1>{}, ICw<1>{}, u);
    for (; u <= n_tiles; ++u) iteration(ICw<0>{}, ICw<0>{}, u);
  }

  if (!wave_valid) return;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const float l_tot = half_sum(l_run[rb]);
    const bool dead = (l_tot == 0.f) || (l_tot != l_tot);
    const float inv = dead ? 1.f : 1.f / l_tot;
    if (row_valid[rb]) {
      E* orow = op + (int64_t)my_row[rb] * p.o_rs;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          V4 ov;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) ov[jj] = (E)(o_acc[rb][db][4 * g + jj] * inv);
          *reinterpret_cast<V4*>(orow + 32 * db + 8 * g + 4 * hi) = ov;
        }
      if (hi == 0) lsep[my_row[rb]] = dead ? INFINITY : (m_run[rb] * cs * kLn2 + __logf(l_tot));
    }
  }
}

template <typename E, int D>
static int launch_fwd_w64_t(const FwdK& p, hipStream_t stream) {
  constexpr int smem = 4 * 64 * D * 2;
  auto kern = fa_fwd_w64_kernel<E, D>;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return -1;
    hipFuncAttributes fattr;  // LDS is addressed by byte offset: the dynamic segment must start at 0
    if (hipFuncGetAttributes(&fattr, (const void*)kern) != hipSuccess || fattr.sharedSizeBytes != 0) return -1;
    attr_done = true;
  }
  const long long total = units_grid(p.n_units, p.unit_size);
  if (total <= 0) return 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), smem, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// 4 waves x 64 rows: query block = 256 rows
int launch_fwd_w64(const FwdK& p, int dtype_bf16, int d, hipStream_t stream) {
  if ((uint64_t)64 * (uint64_t)(p.k_rs > p.v_rs ? p.k_rs : p.v_rs) * 2u >= (1ull << 31)) return -2;
  if (p.softcap > 0.f || p.alibi != nullptr) return -2;
  if (dtype_bf16) {
    if (d == 128) return launch_fwd_w64_t<__bf16, 128>(p, stream);
    if (d == 64) return launch_fwd_w64_t<__bf16, 64>(p, stream);
  } else {
    if (d == 128) return launch_fwd_w64_t<_Float16, 128>(p, stream);
    if (d == 64) return launch_fwd_w64_t<_Float16, 64>(p, stream);
  }
  return -2;
}

}  // namespace fa
