// EXPERIMENT (measured, not faster; not part of the default library): the dQ pass of the 5-contraction backward.  experiments/ds_spill.patch includes this
// text into a copy of flash-attention_amd/csrc/fa_bwd_w64.hip (inside namespace fa, after the 64-rows-per-wave dQ kernel whose helpers it uses) and puts
// the FA_BWD_MODE=2 dispatch back into a copy of fa_api.cpp (experiments/build_experiments.py).  Records: profiles/r02_bwd_5_vs_7_contractions.txt,
// profiles/r03_bwd_5_contractions_mall.txt.
// ------------------------------------------------------------------------------------------------------------------------
// dQ from spilled dS (BwdK::ds_ws): dQ^T[d][query] = sum_key K^T[d][key] . dS^T[key][query] -- ONE contraction instead of the
// three of the recomputing kernels.  8 waves x 32 query rows; K tiles (64 keys) shared through LDS, each wave's dS
// sub-tiles DMA'd into its private LDS rows and read back transposed (ds_read_b64_tr_b16 turns the writer's lane = key image
// into the lane = query B operand; fa_device.h ds_slot).  No score arithmetic at all, so this kernel also serves softcap,
// ALiBi and dropout: they are folded into dS by the dK/dV kernel.
// ------------------------------------------------------------------------------------------------------------------------
template <typename E, int D>
__global__ void __launch_bounds__(512, 1) fa_bwd_dq_ds_kernel(const BwdK p) {
  using T = ElemTraits<E>;
  using V8 = typename T::v8;
  // 8 waves x 32 rows, two waves per SIMD: the operands are all 8-byte transpose reads, which need several waves per SIMD in
  // flight to reach the LDS rate (MI355X_MICROARCH.md, LDS); per tile a wave reads the K tile (16 KB) + its dS (4 KB) for 16 MFMAs
  constexpr int NW = 8, BM = NW * 32, BN = 64, CPR = D / 8;
  constexpr int ROW_BYTES = D * 2, TILE_BYTES = BN * ROW_BYTES, DB = D / 32;
  constexpr int DS_WAVE = 2 * 2048;        // per wave and tile: the two 32-key sub-tiles of its 32 rows (contiguous in the workspace)
  constexpr int DS_BUF = NW * DS_WAVE;
  constexpr int OFF_DS = 2 * TILE_BYTES;   // LDS: K0 | K1 | dS0 | dS1

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char FA_LDS* lds = (char FA_LDS*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;

  int b, h, m_block;
  {
    const int w = xcd_interleave(blockIdx.x, p.q_units, p.q_unit_size, p.q_unit_hpx);
    if (w < 0) return;
    const int bh = w / p.nmb;
    const int mbr = w - bh * p.nmb;
    m_block = (p.wr >= 0) ? (p.nmb - 1 - mbr) : mbr;
    b = bh / p.h;
    h = bh - b * p.h;
  }
  const int hk = h / p.hk_ratio;
  const int sq = p.sq, sk = p.sk;
  const int m0 = m_block * BM;
  if (m0 >= sq) return;
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
  const E* __restrict__ ds_row = (const E*)p.ds_ws + (((((int64_t)b * p.h + h) * p.ds_nq32) + (w_row0 >> 5)) * p.ds_nk32 << 10) + lane * 8;

  constexpr int RPD = 1024 / ROW_BYTES, NDMA = TILE_BYTES / 1024, DPW = NDMA / NW;
  static_assert(NDMA % NW == 0 && DPW >= 1, "tile does not divide over the waves");
  auto load_tile = [&](int n, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      const int idx = wave * DPW + i;
      const int row = idx * RPD + lane / CPR;
      const int c = (lane % CPR) ^ swz16<D>(row);
      const int key = min(n * BN + row, sk - 1);   // rows past the last key: clamped copies, their dS is zero
      lds_dma_16B(kp + (int64_t)key * p.k_rs + c * 8, lds + buf * TILE_BYTES + idx * 1024);
    }
    if (wave_valid) {
      const E* src = ds_row + ((int64_t)(2 * n) << 10);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (2 * n + (j >> 1) < p.ds_nk32) lds_dma_16B(src + j * 512, lds + OFF_DS + buf * DS_BUF + wave * DS_WAVE + j * 1024);
    }
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

  if (n_min < n_max) {
    load_tile(n_min, 0);
    lds_dma_wait_all();
    __syncthreads();
  }
  auto tile = [&](auto curc, int n) __attribute__((always_inline)) {
    constexpr int cur = decltype(curc)::value;
    if (n + 1 < n_max) load_tile(n + 1, cur ^ 1);
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
    lds_dma_wait_all();
    __syncthreads();
  };
  for (int n = n_min; n < n_max; n += 2) {
    tile(ICw<0>{}, n);
    if (n + 1 < n_max) tile(ICw<1>{}, n + 1);
  }

  if (!wave_valid) return;
  E* dqtile = (E*)p.dq + (int64_t)b * p.dq_bs + (int64_t)w_row0 * p.dq_rs + (int64_t)h * p.dq_hs;
  store_tile_via_lds<E, D>(lds + wave * 32 * (ROW_BYTES + 16), dq_acc, p.scale, dqtile, p.dq_rs, sq - w_row0, lane);
}

template <typename E, int D>
static int launch_bwd_dq_ds_t(const BwdK& p, hipStream_t stream) {
  constexpr int LOOP = 2 * 64 * D * 2 + 2 * 8 * 4096, STAGE_OUT = 256 * (D * 2 + 16);
  constexpr int smem = LOOP > STAGE_OUT ? LOOP : STAGE_OUT;
  auto kern = fa_bwd_dq_ds_kernel<E, D>;
  static std::atomic<unsigned long long> attr_mask{0};
  if (ensure_dyn_lds(attr_mask, (const void*)kern, smem) != 0) return -1;
  const long long total = units_grid(p.q_units, p.q_unit_size);
  if (total <= 0) return 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(512), smem, stream, p);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// dQ pass of the 5-contraction backward (fixed-length batches, p.ds_ws filled by the dK/dV kernel, nmb sized for 256 rows)
int launch_bwd_dq_ds(const BwdK& p, int dtype_bf16, int d, hipStream_t stream) {
  if (!p.ds_ws || p.cu_q || p.cu_k) return -2;
  if (dtype_bf16) {
    if (d == 128) return launch_bwd_dq_ds_t<__bf16, 128>(p, stream);
    if (d == 64) return launch_bwd_dq_ds_t<__bf16, 64>(p, stream);
  } else {
    if (d == 128) return launch_bwd_dq_ds_t<_Float16, 128>(p, stream);
    if (d == 64) return launch_bwd_dq_ds_t<_Float16, 64>(p, stream);
  }
  return -2;
}

