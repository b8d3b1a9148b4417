// C-ABI entry points of libfa_gfx950.so (declared in include/fa_gfx950.h).
//
// Host layer of the hot path: validates the contract, normalises (causal, window) exactly as the
// reference host functions do, fills the device-side parameter block and enqueues the kernels.
// Behavioural spec: reference csrc/flash_attn/flash_api.cpp -- mha_fwd :368-536,
// mha_varlen_fwd :538-788, mha_bwd :800-1008, mha_varlen_bwd :1010-1241, set_params_fprop :44-177
// (window/causal normalisation :155-162 and :422-427).  No ATen here: buffers are caller-owned.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/fa_gfx950.h"
#include "fa_device.h"
#include "fa_kernel_params.h"
#include "fa_launch.h"

namespace {

thread_local char g_err[512] = {0};

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}

// Environment knobs are read once per process (getenv on every launch races with setenv and costs a libc lock);
// fa_knobs_reload() re-reads them.  Old snapshots are never freed (a handful of bytes per reload).
std::atomic<const fa::Knobs*> g_knobs{nullptr};
const fa::Knobs* read_knobs() {
  fa::Knobs* k = new fa::Knobs();
  k->fwd_nw = env_int("FA_FWD_NW", 0);
  const char* t = getenv("FA_RESCALE_THR");
  k->rescale_thr = (t && *t) ? (float)atof(t) : 8.f;
  if (!(k->rescale_thr >= 0.f) || k->rescale_thr > 16.f) k->rescale_thr = 0.f;
  k->varlen_list = env_int("FA_VARLEN_LIST", 1);
  k->il_sched = env_int("FA_IL_SCHED", 3);
  k->bwd_dq_nw = env_int("FA_BWD_DQ_NW", 0);
  k->bwd_mode = env_int("FA_BWD_MODE", 0);
  k->bwd_dkdv = env_int("FA_BWD_DKDV", 0);
  k->bwd_ds_cap_mb = env_int("FA_BWD_DS_CAP_MB", 8192);
  k->bwd_fused_check = env_int("FA_BWD_FUSED_CHECK", 0);
  k->bwd_c5_cap_mb = std::max(16, std::min(65536, env_int("FA_BWD_C5_CAP_MB", 1024)));
  k->fz_line = std::max(1, std::min(64, env_int("FA_FZ_LINE", 32)));
  k->bwd_gsplit = env_int("FA_BWD_GSPLIT", 1);
  k->w64_persist = env_int("FA_W64_PERSIST", 1);
  k->strict = env_int("FA_STRICT", 0);
  k->dkdv_prescale = env_int("FA_DKDV_PRESCALE", 0);
  k->bwd_fuse_delta = env_int("FA_BWD_FUSE_DELTA", 1);
  k->pack_gqa = env_int("FA_PACK_GQA", 1);
  if (k->strict) k->rescale_thr = 0.f;
  return k;
}

}  // namespace
namespace fa {
const Knobs& knobs() {
  const Knobs* k = g_knobs.load(std::memory_order_acquire);
  if (!k) {
    const Knobs* fresh = read_knobs();
    if (g_knobs.compare_exchange_strong(k, fresh, std::memory_order_acq_rel)) k = fresh;
    else delete fresh;
  }
  return *k;
}
LastSchedule& last_schedule() {
  thread_local LastSchedule ls{};
  return ls;
}
}  // namespace fa
namespace {

// Head dims with their own kernels (the reference builds the same set: static_switch.h:92-110).  32 / 96 / 192 are "trimmed"
// variants of the 64 / 128 / 256 kernels (fa_fwd.hip: same LDS pitch, fewer k-steps and output blocks) on the lock-step schedules.
bool head_dim_trimmed(int d) { return d == 32 || d == 96 || d == 192; }
bool head_dim_native(int d) { return d == 64 || d == 128 || d == 256 || head_dim_trimmed(d); }
// kernel head dim a call runs on: the next built size.  A head dim between the built sizes (a multiple of 8: 40, 72, 80, 104, 160, 224, ..)
// runs with a run-time column bound (FwdK / BwdK::d_chunks: the chunks behind the head dim read as zeros and are never stored; lock-step
// forward, 4-wave dQ kernel, dK/dV kernel, delta pre-pass) -- the reference rounds internally and masks columns the same way
// (flash_api.cpp:458,872, Is_even_K); no padded copies anywhere.
int head_dim_kernel(int d) { for (int n : {32, 64, 96, 128, 192}) if (d <= n) return n; return 256; }
int head_dim_pitch(int d) { return d <= 64 ? 64 : d <= 128 ? 128 : 256; }  // row pitch of tiles and of split-KV partial rows

// Reference flash_api.cpp:422-427 (+ :155-162): windows at least as wide as the key sequence are
// unbounded, a single query row needs no causal mask, causal means window_right = 0.
void normalize_window(int seqlen_q, int seqlen_k, bool has_alibi, int& is_causal, int& wl, int& wr) {
  if (wl >= seqlen_k) wl = -1;
  if (wr >= seqlen_k) wr = -1;
  if (seqlen_q == 1 && !has_alibi) is_causal = 0;
  if (is_causal) wr = 0;
  if (wl < 0) wl = -1;
  if (wr < 0) wr = -1;
}

// dropout fields shared by the forward and backward parameter blocks
int check_dropout(float p_dropout, const void* rng_state) {
  if (!(p_dropout >= 0.f) || p_dropout >= 1.f) return fail(FA_ERR_INVALID_ARGUMENT, "p_dropout must be in [0, 1)");
  if (p_dropout > 0.f && !rng_state) return fail(FA_ERR_INVALID_ARGUMENT, "p_dropout > 0 needs a device rng_state {seed, offset}");
  return FA_OK;
}
template <typename K> void fill_dropout(K& k, float p_dropout, const uint64_t* rng_state, int seqlen_k) {
  if (p_dropout > 0.f) {
    k.rng = rng_state;
    const float p_keep = 1.0f - p_dropout;  // single precision throughout, as csrc/flash_attn_ck/mha_fwd.cpp derives its uint8 threshold
    k.drop_thr8 = (uint32_t)std::floor(p_keep * 255.0f);
    k.rp_keep = 1.f / (1.f - p_dropout);
  }
}

// Split-KV schedule of the decode path (reference num_splits_heuristic, flash_api.cpp:275-347, re-derived for 256 CUs
// with two workgroups each): split only single-block query lengths whose (batch x head) grid leaves most CUs idle,
// into the fewest key splits that give every CU two workgroups; a split is a whole number of 64-key tiles.
// Head packing on the fixed-length forward and the KV-cache path (FwdK::pack_g; the reference packs only the single-row decode step by reshaping q,
// flash_api.cpp:429-437 -- FA3 generalises it as PackGQA, hopper/pack_gqa.h): when all g query heads of a KV group times the
// query rows fit one 128-row block, the group's heads become rows of that block and K/V are streamed once per KV head instead
// of once per query head.  Returns g (1 = no packing).
int pack_group(const FaFwdParams* a) {
  const int g = a->h_k > 0 ? a->h / a->h_k : 1;
  if (g <= 1 || !fa::knobs().pack_gqa || a->cu_seqlens_q || a->seqused_q || a->p_dropout > 0.f || a->seqlen_q < 1 || (long)a->seqlen_q * g > 128) return 1;
  return g;
}
int choose_splits(const FaFwdParams* a, int& split_tiles) {
  const int tiles = (a->seqlen_k + 63) / 64;
  split_tiles = tiles;
  if (a->seqlen_q > 128 || tiles < 2 || a->num_splits == 1) return 1;
  int want = a->num_splits;
  if (want <= 0) {
    const long units = (long)a->b * a->h / pack_group(a);   // workgroups of the unsplit grid
    if (units >= 384) return 1;
    want = (int)((512 + units - 1) / units);
    want = std::min(want, std::max(1, tiles / 4));  // at least 4 tiles (256 keys) per split
  }
  want = std::max(1, std::min(std::min(want, 64), tiles));
  split_tiles = (tiles + want - 1) / want;
  return (tiles + split_tiles - 1) / split_tiles;
}
int64_t splitkv_bytes(const FaFwdParams* a, int n_splits) {
  if (n_splits <= 1) return 0;
  return (int64_t)n_splits * a->b * a->h * a->seqlen_q * (head_dim_pitch(a->d) + 1) * (int64_t)sizeof(float);
}

// Forward schedule code: 64 = 64-rows-per-wave kernel (fa_fwd_w64.hip, 4 waves, 256-row blocks), 34 / 38 = software-pipelined
// kernel (fa_fwd_il.hip) with 4 / 8 waves per workgroup, 4 / 8 = lock-step kernel with 4 / 8 waves, 16 = 8-wave ping-pong
// (fa_fwd.hip).  wl / wr = normalised window.
// The 64-rows-per-wave forward addresses K and V through buffer descriptors with 32-bit byte offsets from the (batch, kv-head)
// base (fa_fwd_w64.hip: launch_fwd_w64): the whole key range of a sequence plus two tiles of overshoot must span < 4 GiB.
// Q and O are addressed the same way over the 256 rows of a query block (q_srd_of / dma_q_piece, the epilogue's lane_off / soff).
bool w64_span_ok(const FaFwdParams* a) {
  const uint64_t rs = (uint64_t)std::max<int64_t>(a->k_row_stride, a->v_row_stride);
  const uint64_t qo = (uint64_t)std::max<int64_t>(a->q_row_stride, a->o_row_stride);
  const uint64_t keys = a->block_table ? 64 : (a->seqlen_k > 0 ? a->seqlen_k : 1);   // (a paged cache is addressed tile by tile)
  // (paged: the kernel forms a page's and a tile's byte sizes in 32 bits -- fa_fwd_w64.hip pg_bytes_* / tl_bytes_*)
  if (a->block_table) {
    const uint64_t pg = (uint64_t)std::max<int64_t>(a->k_batch_stride, a->v_batch_stride) * 2u;
    if (pg >= (1ull << 32) || (uint64_t)std::max(a->page_block_size, 1) * rs * 2u >= (1ull << 32)) return false;
  }
  return (keys + 128) * rs * 2u < (1ull << 32) && 256ull * qo * 2u < (1ull << 32);
}
int fwd_schedule_nw(const FaFwdParams* a, int wl, int wr) {
  // Schedule (measured on MI355X, tools/ab_bench.py, profiles/r03_fwd_schedules.txt; FA_FWD_NW overrides):
  //   head dim 128, >= 512 query rows: the 64-rows-per-wave kernel (persistent 256-row workgroups, one per CU) from 8 key tiles of 64
  //   keys per query block on average, as long as its 256-row blocks still fill the chip (>= 200 of them; key loops of >= 32 tiles
  //   -- non-causal S >= 2048, causal S >= 4096, config 3 included -- take it regardless).  With round 3's per-block costs (one
  //   iteration body, 7k-clock prologue) it beats the 4-wave pipelined kernel by 7 % at S = 1024, 14-21 % at S = 2048 and 19-23 %
  //   from S = 4096 (1.20-1.24 vs 1.01-1.03 PFLOP/s non-causal); at S = 512 the two tied until round 5 peeled a wave's first and last
  //   iteration (fixed-length batches under a right bound now take it from 4 visible tiles on average, see below).  Shorter loops and small grids stay on the
  //   4-wave pipelined kernel (Q fragments in registers, two 128-row workgroups per CU hide each other's prologue / epilogue).
  //   FA_STRICT keeps the pipelined kernels (fp32 scaling of every score).  D = 64 has half the MFMA work per softmax element: the
  //   64-rows-per-wave kernel wins from 32 key tiles on average without a right bound (S >= 2048: +1 %, S >= 4096: +12-16 %) and from
  //   16 under a causal mask (S = 2048: +12 %, S = 1024: -3 %).
  //   K/V views whose key range spans >= 4 GiB (w64_span_ok) fall back from the 64-rows-per-wave kernel to the pipelined one,
  //   which addresses tile by tile.
  int nw = fa::knobs().fwd_nw;
  if (nw != 4 && nw != 8 && nw != 16 && nw != 34 && nw != 38 && nw != 64) {
    const bool right_bounded = (wr >= 0);
    const long avg_keys = right_bounded ? (a->seqlen_k + 1) / 2 : a->seqlen_k;
    const long span = (wl >= 0) ? std::min<long>(avg_keys, wl + (wr >= 0 ? wr : a->seqlen_k) + 256) : avg_keys;
    const long tiles = span / 64;
    const long blocks256 = a->cu_seqlens_q ? (long)a->h * (a->total_q / 256 + 1) : (long)a->b * a->h * ((a->seqlen_q + 255) / 256);
    const bool fills = blocks256 >= 200;
    const int fallback = a->seqlen_q > 128 ? 34 : 4;
    if (a->d == 128) {
      const bool long_loop = tiles >= 32 && a->seqlen_q >= 512;
      if (fa::knobs().strict) nw = long_loop ? (tiles >= 48 ? 38 : 34) : fallback;
      // (a left window bound keeps the old threshold: both ends of its short key range are masked iterations, and the early waves of a
      // block idle at both -- config 5, 20 tiles per block: 792 TFLOP/s on the pipelined kernel against 741-763 on this one)
      // (round 5: under a right bound from 4 tiles on average -- causal S = 512 -- since a wave's first and last iteration are peeled: 431 against 383 TFLOP/s there,
      // profiles/r05_fwd_w64_peel.txt; without a right bound 4-7 tiles means 256-448 keys in all, and a packed batch is sized here by its LONGEST sequence (its
      // short ones would leave most of a 256-row block empty): neither was measured, the old threshold stands for both)
      else nw = (long_loop || (tiles >= ((right_bounded && !a->cu_seqlens_q) ? 4 : 8) && wl < 0 && a->seqlen_q >= 512 && fills)) ? 64 : fallback;
    } else if (a->d == 64) {
      const long need = right_bounded ? 16 : 32;
      nw = (!fa::knobs().strict && a->seqlen_q >= 512 && (tiles >= 64 || (tiles >= need && fills))) ? 64 : fallback;
    } else {
      nw = fallback;
    }
    if (nw == 64 && !w64_span_ok(a)) nw = 38;   // (only the heuristic falls back: a forced FA_FWD_NW=64 reaches the launcher's -3 and its message)
  }
  return nw;
}
// query rows per workgroup of the schedule the forward will run (the lock-step variants serve softcap / ALiBi / dropout /
// head dim 256 / split keys)
int fwd_block_rows(const FaFwdParams* a, int nw, bool split) {
  if (a->d > 128 || !head_dim_native(a->d) || head_dim_trimmed(a->d) || split) return 128;
  if (nw == 64) return 256;
  if (nw == 34 || nw == 38) return 32 * (nw - 30);  // (the lock-step fallback of a pipelined schedule keeps the wave count)
  return nw == 16 ? 256 : 32 * nw;
}
// varlen work list: worth a pre-pass when a max_seqlen-sized grid would be mostly empty slots
int64_t varlen_list_entries(const FaFwdParams* a, int bm) {
  if (fa::knobs().varlen_list == 0) return 0;  // debugging switch: always the dense grid
  const int64_t dense = (int64_t)a->b * ((a->seqlen_q + bm - 1) / bm);
  const int64_t bound = (int64_t)a->total_q / bm + a->b;
  return (dense * 4 > bound * 5 && dense >= 64) ? bound : 0;
}

int check_common(int b, int h, int h_k, int d, int dtype, float softcap, bool forward = false) {
  if (b <= 0) return fail(FA_ERR_INVALID_ARGUMENT, "batch size must be positive");
  if (h <= 0 || h_k <= 0 || h % h_k != 0)
    return fail(FA_ERR_INVALID_ARGUMENT, "Number of heads in key/value must divide number of heads in query");
  if (d <= 0 || d > 256 || d % 8 != 0)
    return fail(FA_ERR_INVALID_ARGUMENT, "head dimension must be a multiple of 8 and at most 256");
  if (dtype != FA_DTYPE_FP16 && dtype != FA_DTYPE_BF16)
    return fail(FA_ERR_INVALID_ARGUMENT, "FlashAttention only supports fp16 and bf16 data type");
  if (softcap < 0.f) return fail(FA_ERR_INVALID_ARGUMENT, "softcap must be non-negative");
  return FA_OK;
}

// What the heuristic's pick becomes once the features have had their say (shared by do_fwd and fa_fwd_schedule_query):
//   64 = the 64-rows-per-wave kernel: plain attention, ALiBi under a right bound on the diagonal (its FEAT_ALIBI variant: the bias rides in the
//        score chains' C operand, key tiles walked downwards), or softcap (FEAT_CAP, round 5: seven vector instructions per score, staged over three
//        gaps), dropout without the random-byte output (FEAT_DROP, round 5: eight Philox calls per step spread two rounds per gap), and plain attention over
//        a paged cache (round 5: a buffer descriptor per 64-key tile); anything else that asked for it runs the 8-wave lock-step kernel on the same 256-row blocks;
//   34 / 38 = the software-pipelined kernel with 4 / 8 waves: plain attention only, else the lock-step kernel with the same wave count.
int resolve_fwd_features(const FaFwdParams* a, int nw, int wr, int n_splits, int pack, bool bounded, bool& w64, bool& il) {
  const bool shape_ok = n_splits == 1 && pack == 1 && !bounded;
  const bool base0 = !(a->p_dropout > 0.f) && shape_ok;
  const bool base = base0 && !(a->softcap > 0.f);
  const bool plain = base && !a->alibi_slopes;
  const bool w64_alibi = base && a->alibi_slopes && wr == 0;
  const bool w64_cap = base0 && a->softcap > 0.f && !a->alibi_slopes;
  const bool w64_drop = shape_ok && a->p_dropout > 0.f && !a->randval && !(a->softcap > 0.f) && !a->alibi_slopes && !a->block_table;   // (no random-byte output there)
  // (a paged cache: the plain variant only, and not together with a batch index or left padding -- the API rejects those combinations anyway)
  const bool paged_ok = !a->block_table || (plain && !a->cache_batch_idx && !a->leftpad_k && a->page_block_size % 64 == 0);
  w64 = nw == 64 && (plain || w64_alibi || w64_cap || w64_drop) && paged_ok;
  if (nw == 64 && !w64) nw = 8;
  il = (nw == 34 || nw == 38) && plain;
  if ((nw == 34 || nw == 38) && !il) nw -= 30;
  return nw;
}

int do_fwd(const FaFwdParams* a, void* stream, bool varlen, bool kvcache = false) {
  if (!a) return fail(FA_ERR_INVALID_ARGUMENT, "params is NULL");
  g_err[0] = 0;
  if (int rc = check_common(a->b, a->h, a->h_k, a->d, a->dtype, a->softcap, true)) return rc;
  if (!a->q || !a->k || !a->v || !a->o || !a->softmax_lse)
    return fail(FA_ERR_INVALID_ARGUMENT, "q, k, v, o and softmax_lse must be non-NULL");
  if (varlen != (a->cu_seqlens_q != nullptr) || varlen != (a->cu_seqlens_k != nullptr))
    return fail(FA_ERR_INVALID_ARGUMENT, varlen ? "fa_varlen_fwd needs cu_seqlens_q and cu_seqlens_k"
                                                : "fa_fwd takes fixed-length batches (cu_seqlens must be NULL)");
  if (a->seqlen_q < 0 || a->seqlen_k < 0) return fail(FA_ERR_INVALID_ARGUMENT, "negative sequence length");
  if (!kvcache && (a->cache_batch_idx || a->seqused_k_add))
    return fail(FA_ERR_INVALID_ARGUMENT, "cache_batch_idx / seqused_k_add are fa_fwd_kvcache arguments");
  if (!kvcache && !varlen && (a->block_table || a->leftpad_k))
    return fail(FA_ERR_INVALID_ARGUMENT, "block_table / leftpad_k are fa_varlen_fwd and fa_fwd_kvcache arguments");
  if (a->leftpad_k && a->block_table)
    return fail(FA_ERR_INVALID_ARGUMENT, "We don't support Paged KV and leftpad_k running at the same time yet");
  if (a->block_table) {
    if (a->cache_batch_idx) return fail(FA_ERR_INVALID_ARGUMENT, "Paged KVcache does not support cache_batch_idx");
    if (a->page_block_size <= 0 || a->page_block_size % 256 != 0)
      return fail(FA_ERR_INVALID_ARGUMENT, "Paged KV cache block size must be divisible by 256");
  }
  if (int rc = check_dropout(a->p_dropout, a->rng_state)) return rc;
  if (kvcache && a->p_dropout > 0.f) return fail(FA_ERR_INVALID_ARGUMENT, "fa_fwd_kvcache is an inference path: p_dropout must be 0");
  if (a->randval && !(a->p_dropout > 0.f)) return fail(FA_ERR_INVALID_ARGUMENT, "return_softmax is only supported when p_dropout > 0.0");
  if (a->num_splits > 1 && !kvcache) return fail(FA_ERR_INVALID_ARGUMENT, "num_splits > 1 is not supported");
  if (a->seqlen_q == 0 || a->total_q == 0) return FA_OK;  // nothing to write

  fa::FwdK k{};
  k.n_splits = 1;
  k.pack_g = 1;
  k.q = a->q; k.k = a->k; k.v = a->v; k.o = a->o; k.lse = a->softmax_lse;
  k.q_bs = a->q_batch_stride; k.q_rs = a->q_row_stride; k.q_hs = a->q_head_stride;
  k.k_bs = a->k_batch_stride; k.k_rs = a->k_row_stride; k.k_hs = a->k_head_stride;
  k.v_bs = a->v_batch_stride; k.v_rs = a->v_row_stride; k.v_hs = a->v_head_stride;
  k.o_bs = a->o_batch_stride; k.o_rs = a->o_row_stride; k.o_hs = a->o_head_stride;
  k.cu_q = a->cu_seqlens_q; k.cu_k = a->cu_seqlens_k; k.seqused_k = a->seqused_k; k.seqused_q = a->seqused_q;
  k.kv_batch_idx = a->cache_batch_idx; k.block_table = a->block_table; k.block_table_bs = a->block_table_batch_stride;
  k.page_size = a->page_block_size; k.seqused_add = a->seqused_k_add; k.leftpad_k = a->leftpad_k;
  k.alibi = a->alibi_slopes; k.alibi_bs = a->alibi_batch_stride;
  k.b = a->b; k.h = a->h; k.h_k = a->h_k; k.hk_ratio = a->h / a->h_k;
  k.sq = a->seqlen_q; k.sk = a->seqlen_k; k.total_q = a->total_q;
  int causal = a->is_causal, wl = a->window_left, wr = a->window_right;
  normalize_window(a->seqlen_q, a->seqlen_k, a->alibi_slopes != nullptr, causal, wl, wr);
  k.wl = wl; k.wr = wr;
  k.scale = a->softmax_scale;
  k.scale_log2 = a->softmax_scale * 1.4426950408889634f;
  k.softcap = a->softcap;
  fill_dropout(k, a->p_dropout, a->rng_state, a->seqlen_k);
  if (a->p_dropout > 0.f) { k.randval = a->randval; k.rv_bs = a->randval_batch_stride; k.rv_hs = a->randval_head_stride; k.rv_rs = a->randval_row_stride; }
  // Deferred O rescale: the running max only moves when a row's max grew by more than this many log2
  // units (P stays <= 2^thr; fp32 accumulators and the relative precision of bf16/fp16 P are unaffected).
  // 0 reproduces the reference's rescale-on-any-growth rule exactly.  Default 8 (measured +4..10 %,
  // parity suite unchanged); override with FA_RESCALE_THR.
  // fp16 P must stay below 65504: the threshold is capped at 15 there (bf16 has fp32's exponent range).
  k.rescale_thr = fa::knobs().rescale_thr;
  if (a->dtype == FA_DTYPE_FP16 && k.rescale_thr > 15.f) k.rescale_thr = 15.f;

  int nw = fwd_schedule_nw(a, wl, wr);
  // (fa_fwd too: a short query chunk with grouped heads is the same problem without a cache -- FA3's PackGQA covers prefill, hopper/pack_gqa.h;
  // packed varlen batches keep one block per head)
  const int pack = (kvcache || !varlen) ? pack_group(a) : 1;
  if (pack > 1) {  // grouped query heads become rows of one block (4-wave lock-step kernel)
    k.pack_g = pack; k.h = a->h_k; k.hk_ratio = 1; k.sq = a->seqlen_q * pack;
    nw = 4;
  }
  const int dk = head_dim_kernel(a->d);   // kernel head dim; != a->d: run-time column bound
  const bool bounded = dk != a->d;
  if (bounded) k.d_chunks = a->d / 8;
  if (dk > 128 || head_dim_trimmed(dk) || bounded) nw = 4;  // head dim 256: one 4-wave lock-step workgroup per CU (512-register budget); trimmed / bounded dims: 4-wave lock-step
  // decode: split the keys over several workgroups when (batch x heads) cannot fill the chip
  if (kvcache) {
    int split_tiles = 0;
    const int ns = choose_splits(a, split_tiles);
    if (ns > 1) {
      const int64_t need = splitkv_bytes(a, ns);
      if (a->workspace && a->workspace_bytes >= need) {
        k.n_splits = ns; k.split_tiles = split_tiles;
        k.o_accum = (float*)a->workspace;
        k.lse_accum = k.o_accum + (int64_t)ns * a->b * a->h * a->seqlen_q * head_dim_pitch(a->d);
        nw = 4;
      } else if (a->num_splits > 1) {
        return fail(FA_ERR_WORKSPACE, "fa_fwd_kvcache: num_splits = %d needs a workspace of %lld bytes (fa_fwd_workspace_bytes)", a->num_splits, (long long)need);
      }  // auto schedule without a workspace: run unsplit
    }
  }
  bool w64 = false, il = false;
  nw = resolve_fwd_features(a, nw, k.wr, k.n_splits, pack, bounded, w64, il);
  const int bm = w64 ? 256 : il ? 32 * (nw - 30) : fa::fwd_block_m(nw);
  k.nmb = (k.sq + bm - 1) / bm;
  if (varlen && !kvcache) {  // uneven packed batch: enumerate the non-empty query blocks, heaviest first
    const int64_t entries = varlen_list_entries(a, bm);
    if (entries > 0 && a->workspace && a->workspace_bytes >= (entries + 1) * 8) {
      fa::SchedK sk{};
      sk.cu_a = a->cu_seqlens_q; sk.cu_o = a->cu_seqlens_k; sk.seqused_o = a->seqused_k;
      sk.list = (int2*)a->workspace; sk.nb = a->b; sk.blk = bm; sk.bound = (int)entries; sk.wl = wl; sk.wr = wr; sk.keys_blocked = 0;
      sk.work_shift = fa::sched_work_shift(a->seqlen_k);
      if (fa::launch_varlen_schedule(sk, (hipStream_t)stream) != 0)
        return fail(FA_ERR_LAUNCH, "schedule kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
      k.work_list = (const int2*)a->workspace;
      k.work_bound = (int)entries;
    }
  }
  fa::choose_units(a->b, a->h_k, k.hk_ratio, k.nmb * k.n_splits, k.n_units, k.unit_size, k.unit_hpx);
  int rc = w64  ? fa::launch_fwd_w64(k, a->dtype == FA_DTYPE_BF16, dk, (hipStream_t)stream)
           : il ? fa::launch_fwd_il(k, a->dtype == FA_DTYPE_BF16, dk, nw - 30, (hipStream_t)stream)
                : fa::launch_fwd(k, a->dtype == FA_DTYPE_BF16, dk, nw, (hipStream_t)stream);
  if (rc == 0) fa::last_schedule().fwd_pack = k.pack_g;
  if (rc == 0 && k.n_splits > 1) rc = fa::launch_splitkv_combine(k, a->dtype == FA_DTYPE_BF16, dk, (hipStream_t)stream);
  if (rc == -2) return fail(FA_ERR_UNSUPPORTED, "no forward kernel for head dim %d", a->d);
  if (rc == -3)
    return fail(FA_ERR_UNSUPPORTED, w64 ? "k/v key range (or a 256-row q/o block) too large for the 64-rows-per-wave forward: (seqlen_k + 128) * k/v row_stride * 2 and 256 * q/o row_stride * 2 bytes must be < 4 GiB "
                                          "(FA_FWD_NW=64 was forced; the default schedule falls back to the pipelined kernel)"
                                        : "k/v row stride too large: one 64-key tile (64 * row_stride * 2 bytes) must span less than 2 GiB "
                                          "(the kernels address a tile with 32-bit lane offsets)");
  if (rc != 0) return fail(FA_ERR_LAUNCH, "forward kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
  return FA_OK;
}

// dQ schedule (fa_launch.h Knobs::bwd_dq_nw).  Measured on MI355X (profiles/r02_bwd_schedules.txt): the 64-rows-per-wave
// kernel wins from ~2k keys at head dim 128 (config 3: dQ 955 vs 997 us, S = 16k non-causal 1383 vs 1584 us) and loses on
// short sequences, where its 256-row blocks leave CUs idle.
// What the 64-per-wave backward kernels cover besides plain attention: ALiBi under a causal right bound (the bias is linear in the key there) in both of them;
// softcap in both at head dim 128, dropout in the dQ kernel at head dim 128 -- one feature at a time (round 5; profiles/r05_bwd_features_w64.txt)
bool bwd_w64_features_ok(const FaBwdParams* a, bool dq_kernel) {
  const int n = (a->softcap > 0.f) + (a->p_dropout > 0.f) + (a->alibi_slopes != nullptr);
  if (n > 1) return false;
  if (a->softcap > 0.f) return a->d == 128;                   // (head dim 64: the 4-wave feature kernel measured 2-4 % ahead)
  if (a->p_dropout > 0.f) return dq_kernel && a->d == 128;
  return !a->alibi_slopes || a->is_causal || a->window_right == 0;
}

int bwd_dq_schedule(const FaBwdParams* a) {
  // trimmed head dims (32 / 96 / 192), head dims between the built sizes and head dim 256 only have the 4-wave, 128-row dQ kernel
  // (fa_bwd.hip: launch_dq_f): the block size fill_bwd / bwd_list_entries derive from the schedule has to be that kernel's, whatever the knob says
  if (head_dim_trimmed(head_dim_kernel(a->d)) || !head_dim_native(a->d) || a->d > 128) return 4;
  const int knob = fa::knobs().bwd_dq_nw;
  if (knob == 4 || knob == 8 || knob == 64) return knob;
  const bool plain = bwd_w64_features_ok(a, true);
  if (a->d == 64) {
    // round 4 (profiles/r04_bwd_schedules.txt): at head dim 64 the 64-rows-per-wave kernel wins from ~2k visible keys per query row ON AVERAGE -- non-causal
    // S >= 2048 (+5.5 .. +7.5 % on the whole backward), causal S >= 8192 (+5.6 .. +7.4 %); it ties at causal S = 4096 and loses below
    // mean visible keys per query row: bottom-right aligned, row i sees keys up to i + (sk - sq) + window_right
    const bool right_bounded = a->is_causal || a->window_right >= 0;
    const long wr = a->is_causal ? 0 : a->window_right;
    long avg_keys = a->seqlen_k;
    if (right_bounded) { avg_keys = (long)a->seqlen_k - a->seqlen_q / 2 + wr; if (avg_keys > a->seqlen_k) avg_keys = a->seqlen_k; if (avg_keys < 0) avg_keys = 0; }
    // (late round 6: without a right bound from 1536 keys -- whole backward S = 1536 567 | 583, S = 1792 578 | 596 TFLOP/s, a tie at S = 1024)
    return (plain && a->window_left < 0 && a->seqlen_q >= 512 && avg_keys >= (right_bounded ? 2048 : 1536)) ? 64 : 4;
  }
  if (a->d != 128 || !plain || a->seqlen_q < 512) return 4;
  if (a->seqlen_k >= 2048) return 64;
  // (late round 6: WITHOUT a right bound the 64-rows-per-wave kernel already leads from 768 keys once its 256-row blocks fill the chip -- whole backward, TFLOP/s, 32 rows
  // per wave | 64: S = 768 569 | 591, S = 1024 631 | 650, S = 1280 667 | 702, S = 1536 677 | 720 (B16 H32: 699 | 751), S = 1600 on 448 blocks 569 | 619, GQA 32/8 S = 1024 on
  // 256 blocks 328 | 345; S = 512 500 | 485.  Under a causal mask a tie at S = 1600 (516 | 522) and behind on small grids (S = 1536 on 96 blocks 174 | 165): profiles/r06_bwd_c5.txt (9))
  const bool right_bounded = a->is_causal || a->window_right >= 0;
  const long blocks256 = a->cu_seqlens_q ? 0 : (long)a->b * a->h * ((a->seqlen_q + 255) / 256);
  const bool featureless = a->softcap == 0.f && a->p_dropout == 0.f && !a->alibi_slopes;   // (the feature variants were not measured below 2k keys)
  return (featureless && !right_bounded && a->window_left < 0 && a->seqlen_k >= 768 && blocks256 >= 256) ? 64 : 4;
}

// dK/dV schedule (fa_launch.h Knobs::bwd_dkdv): 64 = four waves x 64 keys (fa_bwd_dkdv_w64.hip; plain attention or causal ALiBi at head dim 64 / 128, softcap at head dim 128), 8 = eight waves x 32 keys
// (fa_bwd.hip: every feature variant, head dim 256, trimmed head dims).  Measured (profiles/r05_bwd_dkdv_w64.txt): at head dim 128 the 64-keys-per-wave kernel
// wins from 2k query rows per key block (+1 % at S = 2048, +3 .. +6 % on the whole backward from S = 4096, GQA included) and loses below (its pipeline fill /
// drain and 160 KB of LDS per workgroup cost more than they save on a short walk); at head dim 64 it ties or loses everywhere.
int bwd_dkdv_schedule(const FaBwdParams* a) {
  const bool plain = bwd_w64_features_ok(a, false);
  if (!plain || !head_dim_native(a->d) || head_dim_trimmed(head_dim_kernel(a->d)) || (a->d != 128 && a->d != 64)) return 8;
  const int knob = fa::knobs().bwd_dkdv;
  if (knob == 8 || knob == 64) return knob;
  if (fa::knobs().dkdv_prescale) return 8;   // (FA_DKDV_PRESCALE=1 names a variant of the eight-wave kernel)
  // (a key block's walk: the query rows that can see it -- bounded by the window under a two-sided mask; packed batches are sized by their longest sequence
  // and stay on the eight-wave kernel below 2k rows of it)
  long walk = a->seqlen_q;
  const int wr = a->is_causal ? 0 : a->window_right;
  const int ratio = a->h / a->h_k;
  if (a->window_left >= 0 && wr >= 0) walk = std::min<long>(walk, (long)a->window_left + wr + 256) * ratio;   // (the query heads of a group are walked one after the other: config 5, 4 x 1280 rows, measured +1.5 % on this kernel)
  // (round 6, late: ... and so they are without a window -- a key block of a GQA group walks ratio x Sq rows, half of them on average under a right bound.  Measured on the
  // pair, TFLOP/s eight-wave | this kernel: causal S = 1024 ratio 4 158-172 | 165-182 (+5 %), ratio 8 168-172 | 178-182 (+6 %), S = 768 ratio 4 +2 %, S = 512 ratio 4 / 8 -2 %;
  // no mask S = 1024 ratio 4 296-306 | 322-332 (+8.5 %), S = 1536 ratio 4 +8 %, S = 640 ratio 8 +10 %: from 2k walked rows on average, not below 640 rows per head)
  else if (a->window_left < 0 && ratio > 1 && a->seqlen_q >= 640 && !a->cu_seqlens_q) walk = std::max<long>(walk, (long)a->seqlen_q * ratio / (wr >= 0 ? 2 : 1));
  // (late round 6: without any mask from 1536 walked rows -- S = 1536 688 | 720, S = 1792 667 | 695 TFLOP/s on the whole backward, S = 1280 / 1024 +1 ... +2 %; under a causal mask a tie below 2k)
  return (a->d == 128 && walk >= ((wr < 0 && a->window_left < 0) ? 1536 : 2048)) ? 64 : 8;
}

int fill_bwd(const FaBwdParams* a, bool varlen, fa::BwdK& k) {
  if (!a) return fail(FA_ERR_INVALID_ARGUMENT, "params is NULL");
  g_err[0] = 0;
  if (int rc = check_common(a->b, a->h, a->h_k, a->d, a->dtype, a->softcap)) return rc;
  if (!a->dout || !a->q || !a->k || !a->v || !a->o || !a->softmax_lse || !a->dq || !a->dk || !a->dv || !a->softmax_d)
    return fail(FA_ERR_INVALID_ARGUMENT, "dout, q, k, v, o, softmax_lse, dq, dk, dv and softmax_d must be non-NULL");
  if (varlen != (a->cu_seqlens_q != nullptr) || varlen != (a->cu_seqlens_k != nullptr))
    return fail(FA_ERR_INVALID_ARGUMENT, varlen ? "fa_varlen_bwd needs cu_seqlens_q and cu_seqlens_k"
                                                : "fa_bwd takes fixed-length batches (cu_seqlens must be NULL)");
  k = fa::BwdK{};
  k.dout = a->dout; k.q = a->q; k.k = a->k; k.v = a->v; k.o = a->o; k.lse = a->softmax_lse;
  k.dq = a->dq; k.dk = a->dk; k.dv = a->dv; k.delta = a->softmax_d;
  k.do_bs = a->do_batch_stride; k.do_rs = a->do_row_stride; k.do_hs = a->do_head_stride;
  k.q_bs = a->q_batch_stride; k.q_rs = a->q_row_stride; k.q_hs = a->q_head_stride;
  k.k_bs = a->k_batch_stride; k.k_rs = a->k_row_stride; k.k_hs = a->k_head_stride;
  k.v_bs = a->v_batch_stride; k.v_rs = a->v_row_stride; k.v_hs = a->v_head_stride;
  k.o_bs = a->o_batch_stride; k.o_rs = a->o_row_stride; k.o_hs = a->o_head_stride;
  k.dq_bs = a->dq_batch_stride; k.dq_rs = a->dq_row_stride; k.dq_hs = a->dq_head_stride;
  k.dk_bs = a->dk_batch_stride; k.dk_rs = a->dk_row_stride; k.dk_hs = a->dk_head_stride;
  k.dv_bs = a->dv_batch_stride; k.dv_rs = a->dv_row_stride; k.dv_hs = a->dv_head_stride;
  k.cu_q = a->cu_seqlens_q; k.cu_k = a->cu_seqlens_k; k.seqused_q = a->seqused_q; k.seqused_k = a->seqused_k;
  k.alibi = a->alibi_slopes; k.alibi_bs = a->alibi_batch_stride;
  k.b = a->b; k.h = a->h; k.h_k = a->h_k; k.hk_ratio = a->h / a->h_k;
  k.sq = a->seqlen_q; k.sk = a->seqlen_k; k.total_q = a->total_q; k.total_k = a->total_k;
  int causal = a->is_causal, wl = a->window_left, wr = a->window_right;
  normalize_window(a->seqlen_q, a->seqlen_k, a->alibi_slopes != nullptr, causal, wl, wr);
  k.wl = wl; k.wr = wr;
  k.scale = a->softmax_scale;
  k.scale_log2 = a->softmax_scale * 1.4426950408889634f;
  k.softcap = a->softcap;
  if (int rc = check_dropout(a->p_dropout, a->rng_state)) return rc;
  fill_dropout(k, a->p_dropout, a->rng_state, a->seqlen_k);
  k.dq_nw = bwd_dq_schedule(a);
  if (!head_dim_native(a->d)) k.d_chunks = a->d / 8;   // run-time column bound of the next built size's kernels
  const int bwd_bm = a->d > 128 ? 128 : fa::bwd_block_m(k.dq_nw);
  k.nmb = (a->seqlen_q + bwd_bm - 1) / bwd_bm;
  k.nnb = (a->seqlen_k + fa::bwd_block_n(a->d) - 1) / fa::bwd_block_n(a->d);
  fa::choose_units(a->b, a->h_k, a->h / a->h_k, k.nmb, k.q_units, k.q_unit_size, k.q_unit_hpx);
  fa::choose_units(a->b, a->h_k, 1, k.nnb, k.k_units, k.k_unit_size, k.k_unit_hpx);
  return FA_OK;
}

// varlen backward work lists: query blocks for the dQ kernel, key blocks for the dK/dV kernel (entries each, 0 = dense grids)
void bwd_list_entries(const FaBwdParams* a, int64_t& q_entries, int64_t& k_entries) {
  q_entries = k_entries = 0;
  if (!a->cu_seqlens_q || !a->cu_seqlens_k || fa::knobs().varlen_list == 0) return;
  const int bm = a->d > 128 ? 128 : fa::bwd_block_m(bwd_dq_schedule(a)), bn = fa::bwd_block_n(a->d);
  const int64_t dq_dense = (int64_t)a->b * ((a->seqlen_q + bm - 1) / bm), dq_bound = (int64_t)a->total_q / bm + a->b;
  const int64_t dk_dense = (int64_t)a->b * ((a->seqlen_k + bn - 1) / bn), dk_bound = (int64_t)a->total_k / bn + a->b;
  if (dq_dense * 4 > dq_bound * 5 && dq_dense >= 64) q_entries = dq_bound;
  if (dk_dense * 4 > dk_bound * 5 && dk_dense >= 64) k_entries = dk_bound;
}

// (Round 2's two-launch dS-spill backward, FA_BWD_MODE=2 -- a tie with the recomputing pair, profiles/r02_bwd_5_vs_7_contractions.txt, on O(S^2) scratch -- is
// superseded by FA_BWD_MODE=5 below: the same idea on the 64-per-wave kernels with a bounded workspace; git history keeps the old experiment.)
// Fused backward (FA_BWD_MODE=3, fa_bwd.hip fa_bwd_fused_kernel): bytes of the dS workspace (256-B aligned) when the call qualifies, else 0; the sync
// area (fa_kernel_params.h FZ_*) sits behind it.  Same conditions as launch_bwd_fused.
// Round 6: it is the DEFAULT (FA_BWD_MODE=0) where it was measured ahead of the recomputing pair and its workspace stays within bounds (bwd_fused_plan below; profiles/r06_bwd_c5.txt (5), timed
// WITHOUT the status read -- a stream sync per call, which the default path does not do and which cost the launch 9 % at S = 1024): head dim 128, Sq = Sk, at least
// 32 (batch, kv head) units, under a causal mask from 512 to 4096 rows (+8 % / +18 % / +15 % at S = 512 / 1024 / 2048 on the sweep's shapes, +7 ... +9 % at S = 3072 and
// +9 ... +13 % at S = 4096 on the grids that fit the 1 GiB -- 32 heads) and -- until the pair's dQ half got the 64-rows-per-wave kernel below 2k keys, see the return below -- without a mask from 512 to 1536 rows (+1.5 % / +5 % / +5.6 % / +3 ... +5 % at 512 / 768 / 1024 / 1536).
// Where the workspace would exceed 1 GiB it ties or loses anyway (config 3: a tie on 4 GiB; S = 4096 on 64 heads +3 % on 2 GiB; S = 8192 -8 %); without a mask from
// S = 2048 -6 ... -10 %; head dim 64 without a mask -20 % (causal +3 % / -1 %: left to the pair); small grids (16 units) -3 %.
// FA_BWD_MODE=3 forces it wherever it applies (cap FA_BWD_DS_CAP_MB), -1 / 1 never.
bool bwd_fused_by_table(const FaBwdParams* a) {
  int causal = a->is_causal, wl = a->window_left, wr = a->window_right;
  normalize_window(a->seqlen_q, a->seqlen_k, false, causal, wl, wr);
  if (a->d != 128 || wl >= 0 || a->seqlen_q != a->seqlen_k || (long)a->b * a->h_k < 32) return false;
  if (fa::knobs().bwd_dq_nw != 0 || fa::knobs().bwd_dkdv != 0 || fa::knobs().dkdv_prescale || fa::knobs().strict) return false;
  const int s = a->seqlen_q;
  // (late round 6, sync-free timings at large batches: from 256 to 511 rows the launch leads once the grid is large -- causal S = 256: 512 units a tie, 1024 +4 %, 1536 +7 %,
  // 2048 +10 %; S = 320 / 384 on 1024 units +15 % / +13 %, S = 384 / 448 on 512 units +7 % / +9 %, S = 384 on 256 units a tie; S = 128 -5 %; no mask S = 256 / 384 on 2048 units
  // +3 % / +7 %, S = 384 on 512 units +1 %: profiles/r06_bwd_c5.txt (8))
  if (s >= 256 && s < 512) { const long us = (long)a->b * a->h_k * s; return wr == 0 ? us >= 196608 : (wr < 0 && us >= 786432); }
  // (without a mask from 512 rows: the pair, whose dQ half takes the 64-rows-per-wave kernel from 768 keys since late round 6 -- bwd_dq_schedule -- and is then level with
  // this launch or ahead of it at no workspace: S = 768 569 fused | 591 pair, 1024 658 | 650, 1280 696 | 702, 1536 698 | 720, B16 H32 S = 1536 739 | 751; S = 512 a tie on either dQ kernel)
  return wr == 0 && s >= 512 && s <= 4096;
}
struct FusedPack { int np64, c1, jb, head_tiles; };
FusedPack fused_pack(int sq, int sk, int wr) {   // the row packing of the dS workspace (as C5Plan's)
  FusedPack f;
  f.np64 = (sk + 63) / 64;
  f.c1 = wr >= 0 ? (int)std::min<int64_t>(2 * f.np64, ((int64_t)31 + (sk - sq) + wr) / 32 + 2) : 2 * f.np64;
  f.jb = std::max(0, 2 * f.np64 - f.c1);
  f.head_tiles = fa::ds_row_start((sq + 31) / 32, f.c1, f.jb, f.np64);
  return f;
}
// The fused launch's plan (round 6, late): the batch is cut into chunks of whole batch entries so that a chunk's packed dS fits the cap, one launch per chunk on the
// SAME workspace (stream order: launch c + 1 starts when launch c is over).  By the table (mode 0) the cap is 1.25 GiB (the packed triangle of the shapes whose half
// square is exactly 1 GiB comes to 1.03 - 1.08 GiB: B8 S2048 H32 +15 %, B4 S4096 H16 +3 %, B4 S3072 H32 +7 %), a chunk keeps >= 32 (batch, kv head) units, and only
// sequences up to 2048 rows are chunked -- there the launch is 12 ... 25 % ahead of the pair at any grid size (B32 S1024 H32: 524 against 418 TFLOP/s, B16 S2048 H32
// 652 / 583, B64 S512 H32 366 / 313, no mask B32 S1024 H32 698 / 649; profiles/r06_bwd_c5.txt (7)); from 2048 to 4096 rows it is only ahead of the pair ON THE
// SAME GRID (+3 ... +9 %) and the pair gains more from a larger grid than that, so there the whole batch has to fit.  FA_BWD_MODE=3: cap FA_BWD_DS_CAP_MB, chunked whenever needed.
struct FusedPlan { int nb, n_chunks; int64_t ds_bytes, sync_bytes; };   // batch entries per chunk, chunks, bytes of a chunk's dS / sync area
bool bwd_fused_plan(const FaBwdParams* a, FusedPlan& pl) {
  pl = FusedPlan{0, 0, 0, 0};
  const int mode = fa::knobs().bwd_mode;
  if ((mode != 3 && mode != 0) || a->cu_seqlens_q || a->cu_seqlens_k || a->seqused_q || a->seqused_k || (a->d != 128 && a->d != 64)) return false;
  if (a->seqlen_q <= 0 || a->seqlen_k < a->seqlen_q || a->window_left >= 0 || a->softcap > 0.f || a->alibi_slopes || a->p_dropout > 0.f || a->b <= 0) return false;
  if (mode == 0 && !bwd_fused_by_table(a)) return false;
  // (packed rows, fa_device.h ds_row_start -- a causal mask stores its triangle: about half of B*H*Sq*Sk*2 bytes)
  int causal = a->is_causal, wl = a->window_left, wr = a->window_right;
  normalize_window(a->seqlen_q, a->seqlen_k, false, causal, wl, wr);
  const FusedPack fp = fused_pack(a->seqlen_q, a->seqlen_k, wr);
  const int64_t per_batch = (int64_t)a->h * fp.head_tiles * 2048;
  const int64_t cap = mode == 0 ? ((int64_t)1280 << 20) : ((int64_t)fa::knobs().bwd_ds_cap_mb << 20);
  int64_t nb = std::min<int64_t>(a->b, cap / per_batch);
  if (nb < 1) return false;
  const int n_chunks = (int)((a->b + nb - 1) / nb);
  nb = (a->b + n_chunks - 1) / n_chunks;   // (even chunks)
  if (mode == 0 && n_chunks > 1 && (a->seqlen_q > 2048 || (a->b - (n_chunks - 1) * nb) * a->h_k < 32)) return false;   // (the last chunk is the smallest)
  pl.nb = (int)nb; pl.n_chunks = n_chunks;
  pl.ds_bytes = (nb * per_batch + 255) & ~(int64_t)255;
  pl.sync_bytes = fa::fz_sync_words(nb * a->h * ((a->seqlen_q + 255) / 256), fa::knobs().fz_line) * 4;
  return true;
}
int64_t bwd_fused_ds_bytes(const FaBwdParams* a) { FusedPlan pl; return bwd_fused_plan(a, pl) ? pl.ds_bytes : 0; }
int64_t bwd_fused_sync_bytes(const FaBwdParams* a) { FusedPlan pl; return bwd_fused_plan(a, pl) ? pl.sync_bytes : 0; }

// ---- 5-contraction backward (round 6; reference: ONE pass forms S, dP and dS and all three gradients follow from it, csrc/flash_attn/src/flash_bwd_kernel.h:457-733) ----
// The 64-keys-per-wave dK/dV items write dS (input dtype) to a workspace, dQ = dS.K is one contraction instead of the recomputing dQ kernel's three.  The workspace is
// bounded: the (batch, kv head) units are cut into chunks of whole XCD rounds (8 units), launch c runs the dK/dV items of chunk c and the dQ items of chunk c - 1
// in one grid (stream order is the hand-off, nothing spins), two slots alternate.  Rows of a slot are packed (fa_device.h ds_row_start): a causal mask stores its triangle only.
struct C5Plan {
  int hpx, rounds, rounds_per_chunk, n_chunks;
  int nq32, nk32, np64, c1, jb, head_tiles, nmb, nnb;
  int64_t slot_bytes;
};
bool bwd_c5_plan(const FaBwdParams* a, C5Plan& pl) {
  const int mode = fa::knobs().bwd_mode;
  if (mode != 0 && mode != 5) return false;
  if (a->cu_seqlens_q || a->cu_seqlens_k || a->seqused_q || a->seqused_k || (a->d != 128 && a->d != 64)) return false;
  if (a->softcap > 0.f || a->alibi_slopes || a->p_dropout > 0.f || a->seqlen_q <= 0 || a->seqlen_k < a->seqlen_q || a->b <= 0) return false;
  if (fa::knobs().bwd_dkdv == 8 || fa::knobs().dkdv_prescale) return false;   // (knobs that name the eight-wave dK/dV kernel)
  int causal = a->is_causal, wl = a->window_left, wr = a->window_right;
  normalize_window(a->seqlen_q, a->seqlen_k, false, causal, wl, wr);
  if (wl >= 0) return false;
  pl.nq32 = (a->seqlen_q + 31) / 32; pl.nk32 = (a->seqlen_k + 31) / 32;
  pl.nmb = (a->seqlen_q + 255) / 256; pl.nnb = (a->seqlen_k + 255) / 256;
  // rows of 64-key pairs (fa_device.h ds_row_start): a - 1 = key sub-tiles row block 0 sees (keys <= 31 + (sk - sq) + wr)
  pl.np64 = (a->seqlen_k + 63) / 64;
  pl.c1 = wr >= 0 ? (int)std::min<int64_t>(2 * pl.np64, ((int64_t)31 + (a->seqlen_k - a->seqlen_q) + wr) / 32 + 2) : 2 * pl.np64;
  pl.jb = std::max(0, 2 * pl.np64 - pl.c1);
  pl.head_tiles = fa::ds_row_start(pl.nq32, pl.c1, pl.jb, pl.np64);
  const int n_units = a->b * a->h_k, ratio = a->h / a->h_k;
  pl.hpx = (a->h_k % 8 == 0) ? a->h_k / 8 : 0;
  pl.rounds = (n_units + 7) / 8;
  const int64_t round_bytes = (int64_t)8 * ratio * pl.head_tiles * 2048;
  const int64_t slot_cap = ((int64_t)fa::knobs().bwd_c5_cap_mb << 20) / 2;
  if (round_bytes > slot_cap || round_bytes >= ((int64_t)1 << 32)) return false;
  int rpc = (int)std::min<int64_t>(pl.rounds, slot_cap / round_bytes);
  while ((int64_t)rpc * round_bytes >= ((int64_t)1 << 32)) --rpc;   // (32-bit buffer offsets inside a slot)
  pl.n_chunks = (pl.rounds + rpc - 1) / rpc;
  pl.rounds_per_chunk = (pl.rounds + pl.n_chunks - 1) / pl.n_chunks;   // balanced
  pl.slot_bytes = ((int64_t)pl.rounds_per_chunk * round_bytes + 255) & ~(int64_t)255;
  if (mode == 5) return true;
  return false;   // (the measured table goes here)
}

// the dK/dV launch of either schedule (-2 from the 64-keys-per-wave launcher = not covered after all: nothing was enqueued)
// GQA group split (late round 6).  The dK/dV kernels own (batch, kv head, key block) items and walk the group's query heads inside: with few kv heads and a small batch
// the grid does not fill the chip (B2 S1024 H32/2: 16 workgroups on 256 CUs, 55 TFLOP/s).  The group is then split into gs = 2^n virtual kv heads -- consecutive query
// heads each, BwdK::kv_in_shift tells the kernels which real K / V head to read --, their partial dK / dV go to a workspace [b][sk][h_k * gs][d] in the input dtype and
// one small kernel sums them (the reference's own form: per-query-head dK / dV summed by at::sum_out, flash_api.cpp:1000-1004).  Fixed-length batches, head dims the
// kernels hold natively; gs = the smallest power of two (<= 8) that brings the grid to 1024 workgroups, or the whole group.
struct GsplitPlan { int gs, shift; int64_t bytes; };
bool bwd_gsplit_plan(const FaBwdParams* a, GsplitPlan& pl) {
  pl = GsplitPlan{1, 0, 0};
  if (fa::knobs().bwd_gsplit == 0 || a->cu_seqlens_q || a->cu_seqlens_k || a->seqused_q || a->seqused_k || a->h_k <= 0 || a->h % a->h_k != 0 || a->d % 8 != 0 || !head_dim_native(a->d)) return false;
  const int ratio = a->h / a->h_k;
  if (ratio < 2 || a->seqlen_k <= 0 || a->seqlen_q <= 0 || a->b <= 0) return false;
  const long wgs = (long)a->b * a->h_k * ((a->seqlen_k + fa::bwd_block_n(a->d) - 1) / fa::bwd_block_n(a->d));
  int gs = 1, shift = 0;
  // (under a right bound alone the items are uneven -- 1024 of them; with a left window or no mask they are uniform and short walks pay for the split's extra K / V loads
  // and partials: config 5, 512 items, 0.994 -> 1.055 ms with a split in two; B1 S8192 H32/8 window 1024 0.502 -> 0.578 -- so there only a grid under 256 items is split)
  const long target = ((a->is_causal || a->window_right >= 0) && a->window_left < 0) ? 1024 : 256;
  if (fa::knobs().bwd_gsplit > 1) { while (gs * 2 <= fa::knobs().bwd_gsplit && ratio % (gs * 2) == 0) { gs *= 2; ++shift; } }   // (forced, for tests)
  else { while (wgs * gs < target && gs < 8 && ratio % (gs * 2) == 0) { gs *= 2; ++shift; } }   // (measured, profiles/r06_bwd_gsplit.txt: past 8 virtual heads nothing is gained; 512 uneven causal items on 256 CUs still gain 10 % from a split in two)
  if (gs < 2) return false;
  pl.gs = gs; pl.shift = shift;
  pl.bytes = 2 * (((int64_t)a->b * a->seqlen_k * a->h_k * gs * a->d * 2 + 255) & ~(int64_t)255);
  return true;
}

int launch_dkdv_any(const FaBwdParams* a, const fa::BwdK& k_in, int bf, int dk_, hipStream_t s) {
  fa::BwdK k = k_in;
  FaBwdParams a2 = *a;
  GsplitPlan gp;
  const bool split = !k.ds_ws && !k.k_list && bwd_gsplit_plan(a, gp) && a->workspace && a->workspace_bytes >= gp.bytes;
  if (split) {
    const int hk2 = a->h_k * gp.gs;
    a2.h_k = hk2;
    k.h_k = hk2; k.hk_ratio = a->h / hk2; k.kv_in_shift = gp.shift;
    k.dk = a->workspace; k.dv = (char*)a->workspace + gp.bytes / 2;
    k.dk_bs = k.dv_bs = (int64_t)a->seqlen_k * hk2 * a->d; k.dk_rs = k.dv_rs = (int64_t)hk2 * a->d; k.dk_hs = k.dv_hs = a->d;
    fa::choose_units(a->b, hk2, 1, k.nnb, k.k_units, k.k_unit_size, k.k_unit_hpx);
  }
  int rc = -2, nw = 64;
  if (!k.ds_ws && bwd_dkdv_schedule(&a2) == 64) rc = fa::launch_bwd_dkdv_w64(k, bf, a->d, s);
  if (rc == -2) { nw = a->d > 128 ? 4 : 8; rc = fa::launch_bwd_dkdv(k, bf, dk_, s); }
  fa::last_schedule().bwd_dkdv_nw = nw;
  if (rc == 0 && split) {
    rc = fa::launch_bwd_gsum(k.dk, k_in.dk, bf, a->b, a->seqlen_k, a->h_k, gp.gs, a->d, k_in.dk_bs, k_in.dk_rs, k_in.dk_hs, s);
    if (rc == 0) rc = fa::launch_bwd_gsum(k.dv, k_in.dv, bf, a->b, a->seqlen_k, a->h_k, gp.gs, a->d, k_in.dv_bs, k_in.dv_rs, k_in.dv_hs, s);
  }
  return rc;
}

int do_bwd(const FaBwdParams* a, void* stream, bool varlen) {
  fa::BwdK k;
  if (int rc = fill_bwd(a, varlen, k)) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (C5Plan pl; !varlen && a->total_q != 0 && bwd_c5_plan(a, pl) && a->workspace && a->workspace_bytes >= 2 * pl.slot_bytes) {
    // delta pre-pass, then n_chunks + 1 mixed launches
    const int bf = a->dtype == FA_DTYPE_BF16;
    k.nmb = pl.nmb; k.nnb = pl.nnb;
    k.k_units = a->b * a->h_k; k.k_unit_size = pl.nnb; k.k_unit_hpx = pl.hpx;
    k.ds_nq32 = pl.nq32; k.ds_nk32 = pl.np64; k.ds_c1 = pl.c1; k.ds_jb = pl.jb; k.ds_head_tiles = pl.head_tiles;
    k.c5_slot_bytes = (uint32_t)pl.slot_bytes;
    int rc = fa::launch_bwd_delta(k, bf, a->d, s);
    for (int c = 0; rc == 0 && c <= pl.n_chunks; ++c) {
      const int pj0 = c * pl.rounds_per_chunk, pj1 = std::min(pl.rounds, pj0 + pl.rounds_per_chunk);
      const int cj0 = (c - 1) * pl.rounds_per_chunk, cj1 = std::min(pl.rounds, cj0 + pl.rounds_per_chunk);
      k.ds_ws = (char*)a->workspace + (int64_t)(c & 1) * pl.slot_bytes;
      k.ds_rd = (const char*)a->workspace + (int64_t)((c + 1) & 1) * pl.slot_bytes;
      k.c5_np = c < pl.n_chunks ? 8 * (pj1 - pj0) * pl.nnb : 0;
      k.c5_nc = c >= 1 ? 8 * (cj1 - cj0) * (a->h / a->h_k) * pl.nmb : 0;
      k.c5_pbid0 = 8 * pj0 * pl.nnb; k.c5_pj0 = pj0; k.c5_cj0 = cj0;
      rc = fa::launch_bwd_c5(k, bf, a->d, s);
    }
    if (rc == 0) {
      fa::LastSchedule& ls = fa::last_schedule();
      ls.bwd_dkdv_nw = 64; ls.bwd_dq_nw = 64; ls.bwd_spill = 5; ls.bwd_list = 0;
      return FA_OK;
    }
    if (rc != -2) return fail(FA_ERR_LAUNCH, "backward kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    if (int rc2 = fill_bwd(a, varlen, k)) return rc2;   // does not apply after all (nothing was enqueued but the delta pre-pass): the default path, from a clean parameter block
  }
  if (const int64_t fz = varlen ? 0 : bwd_fused_ds_bytes(a); fz > 0 && a->workspace && a->workspace_bytes >= fz + bwd_fused_sync_bytes(a) && a->total_q != 0) {
    // delta pre-pass, then ONE launch: dK / dV and dQ = dS.K
    k.ds_ws = a->workspace;
    k.ds_nq32 = (a->seqlen_q + 31) / 32;
    k.ds_nk32 = (a->seqlen_k + 31) / 32;
    { const FusedPack fp = fused_pack(a->seqlen_q, a->seqlen_k, k.wr); k.ds_np64 = fp.np64; k.ds_c1 = fp.c1; k.ds_jb = fp.jb; k.ds_head_tiles = fp.head_tiles; }
    k.nmb = (a->seqlen_q + 255) / 256;
    k.fuse_sync = (int32_t*)((char*)a->workspace + fz);
    k.fuse_line = fa::knobs().fz_line;
    const int bf = a->dtype == FA_DTYPE_BF16;
    int rc = fa::launch_bwd_delta(k, bf, a->d, s);   // (the whole batch at once)
    FusedPlan fpl;
    bwd_fused_plan(a, fpl);
    const fa::BwdK whole = k;
    for (int b0 = 0; rc == 0 && b0 < a->b; b0 += fpl.nb) {   // one launch per chunk of batch entries, all on the same workspace
      const int nb = std::min(fpl.nb, a->b - b0);
      auto at = [&](const void* base, int64_t bs) { return (const void*)((const char*)base + (int64_t)b0 * bs * 2); };
      k = whole;
      k.b = nb;
      k.dout = at(whole.dout, whole.do_bs); k.q = at(whole.q, whole.q_bs); k.k = at(whole.k, whole.k_bs); k.v = at(whole.v, whole.v_bs); k.o = at(whole.o, whole.o_bs);
      k.dq = (void*)at(whole.dq, whole.dq_bs); k.dk = (void*)at(whole.dk, whole.dk_bs); k.dv = (void*)at(whole.dv, whole.dv_bs);
      k.lse = whole.lse + (int64_t)b0 * a->h * a->seqlen_q; k.delta = whole.delta + (int64_t)b0 * a->h * a->seqlen_q;
      fa::choose_units(nb, a->h_k, 1, k.nnb, k.k_units, k.k_unit_size, k.k_unit_hpx);
      k.fuse_items = nb * a->h * k.nmb;
      k.fuse_keep_err = b0 > 0;   // (the launch's error flag accumulates over the chunks: only the first launch zeroes it)
      rc = fa::launch_bwd_fused(k, bf, a->d, s);
    }
    k = whole;
    if (rc == 0) {
      fa::LastSchedule& ls = fa::last_schedule();
      ls.bwd_dkdv_nw = 8; ls.bwd_dq_nw = 8; ls.bwd_spill = 3; ls.bwd_list = 0;
      return FA_OK;
    }
    if (rc != -2) return fail(FA_ERR_LAUNCH, "backward kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    if (int rc2 = fill_bwd(a, varlen, k)) return rc2;   // does not apply after all: the default path, from a clean parameter block
  }
  if (varlen) {
    int64_t qe, ke;
    bwd_list_entries(a, qe, ke);
    const int64_t need = (qe ? (qe + 1) * 8 : 0) + (ke ? (ke + 1) * 8 : 0);
    if (need > 0 && a->workspace && a->workspace_bytes >= need) {
      char* ws = (char*)a->workspace;
      fa::SchedK sk{};
      sk.nb = a->b; sk.wl = k.wl; sk.wr = k.wr;
      if (qe) {
        sk.cu_a = a->cu_seqlens_q; sk.cu_o = a->cu_seqlens_k; sk.list = (int2*)ws; sk.bound = (int)qe; sk.keys_blocked = 0;
        sk.blk = a->d > 128 ? 128 : fa::bwd_block_m(k.dq_nw);
        sk.work_shift = fa::sched_work_shift(a->seqlen_k);
        if (fa::launch_varlen_schedule(sk, s) != 0) return fail(FA_ERR_LAUNCH, "schedule kernel launch failed");
        k.q_list = (const int2*)ws; k.q_bound = (int)qe;
        ws += (qe + 1) * 8;
      }
      if (ke) {
        sk.cu_a = a->cu_seqlens_k; sk.cu_o = a->cu_seqlens_q; sk.list = (int2*)ws; sk.bound = (int)ke; sk.keys_blocked = 1;
        sk.blk = fa::bwd_block_n(a->d);
        sk.work_shift = fa::sched_work_shift(a->seqlen_q);
        if (fa::launch_varlen_schedule(sk, s) != 0) return fail(FA_ERR_LAUNCH, "schedule kernel launch failed");
        k.k_list = (const int2*)ws; k.k_bound = (int)ke;
      }
    }
  }
  const int bf = a->dtype == FA_DTYPE_BF16;
  // Nothing to differentiate: the caller's dq/dk/dv hold no rows (Sq == 0 / Sk == 0 with fixed
  // shapes are handled by the binder, which zero-fills as flash_api.cpp:992-999 does).
  if (a->seqlen_q == 0 || a->seqlen_k == 0 || a->total_q == 0 || a->total_k == 0) return FA_OK;
  const int dk_ = head_dim_kernel(a->d);   // kernel head dim (!= a->d: BwdK::d_chunks)
  // Round 4: where the 64-rows-per-wave dQ kernel runs (plain attention, head dim 64 / 128, long key loops) it computes softmax_d = rowsum(dO * O) of its own rows
  // in its prologue and goes FIRST; the dK/dV kernel behind it reads softmax_d from memory as before and the delta pre-pass is not launched
  // (FA_BWD_FUSE_DELTA=0: the three-launch order).  -2 from the launcher = the schedule does not apply after all: nothing was enqueued, fall through.
  if (k.dq_nw == 64 && fa::knobs().bwd_fuse_delta) {
    k.fuse_delta = 1;
    int rc = fa::launch_bwd_dq_w64(k, bf, a->d, s);
    if (rc == 0) {
      fa::last_schedule().bwd_dq_nw = 64;
      rc = launch_dkdv_any(a, k, bf, dk_, s);
      fa::last_schedule().bwd_spill = 0; fa::last_schedule().bwd_list = (k.q_list != nullptr) + 2 * (k.k_list != nullptr);
      if (rc != 0) return fail(FA_ERR_LAUNCH, "backward kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
      return FA_OK;
    }
    if (rc != -2) return fail(FA_ERR_LAUNCH, "backward kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
    k.fuse_delta = 0;
  }
  int rc = fa::launch_bwd_delta(k, bf, dk_, s);
  if (rc == 0) {
    rc = launch_dkdv_any(a, k, bf, dk_, s);
  }
  if (rc == 0) rc = fa::launch_bwd_dq(k, bf, dk_, s);
  if (rc == 0) { fa::last_schedule().bwd_spill = 0; fa::last_schedule().bwd_list = (k.q_list != nullptr) + 2 * (k.k_list != nullptr); }
  if (rc == -2) return fail(FA_ERR_UNSUPPORTED, "no backward kernel for head dim %d", a->d);
  if (rc != 0) return fail(FA_ERR_LAUNCH, "backward kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
  return FA_OK;
}

}  // namespace

extern "C" {

int fa_abi_version(void) { return FA_ABI_VERSION; }
int fa_sizeof_fwd_params(void) { return (int)sizeof(FaFwdParams); }
int fa_sizeof_bwd_params(void) { return (int)sizeof(FaBwdParams); }
int fa_sizeof_kvappend_params(void) { return (int)sizeof(FaKvAppendParams); }
int fa_sizeof_rotary_params(void) { return (int)sizeof(FaRotaryParams); }
const char* fa_last_error(void) { return g_err; }
void fa_knobs_reload(void) { g_knobs.store(read_knobs(), std::memory_order_release); }
int fa_last_schedule(int32_t* out, int n) {
  const fa::LastSchedule& ls = fa::last_schedule();
  const int32_t v[FA_SCHEDULE_FIELDS] = {ls.fwd_kernel, ls.fwd_nw, ls.fwd_feat, ls.fwd_splits, ls.fwd_list, ls.d, ls.bf16, ls.bwd_dq_nw, ls.bwd_list, ls.bwd_spill, ls.fwd_pack, ls.bwd_dkdv_nw};
  for (int i = 0; i < n && i < FA_SCHEDULE_FIELDS; ++i) out[i] = v[i];
  return FA_SCHEDULE_FIELDS;
}
const char* fa_last_kernel_name(void) { return fa::last_schedule().name; }
// (the schedule resolution of do_fwd for the entry points without a KV cache: same heuristic, same feature resolution, no split keys)
int fa_fwd_schedule_query(const FaFwdParams* a, int varlen) {
  if (!a) return fail(FA_ERR_INVALID_ARGUMENT, "params is NULL");
  if (int rc = check_common(a->b, a->h, a->h_k, a->d, a->dtype, a->softcap, true)) return rc;
  int causal = a->is_causal, wl = a->window_left, wr = a->window_right;
  normalize_window(a->seqlen_q, a->seqlen_k, a->alibi_slopes != nullptr, causal, wl, wr);
  int nw = fwd_schedule_nw(a, wl, wr);
  const int pack = !varlen ? pack_group(a) : 1;
  if (pack > 1) nw = 4;
  const int dk = head_dim_kernel(a->d);
  const bool bounded = dk != a->d;
  if (dk > 128 || head_dim_trimmed(dk) || bounded) nw = 4;
  bool w64 = false, il = false;
  return resolve_fwd_features(a, nw, wr, 1, pack, bounded, w64, il);
}
int fa_bwd_dq_schedule_query(const FaBwdParams* a) {
  if (!a) return fail(FA_ERR_INVALID_ARGUMENT, "params is NULL");
  if (int rc = check_common(a->b, a->h, a->h_k, a->d, a->dtype, a->softcap)) return rc;
  return bwd_dq_schedule(a);
}

int fa_bwd_plan_query(const FaBwdParams* a, int32_t* out, int n) {
  if (!a || !out) return fail(FA_ERR_INVALID_ARGUMENT, "params / out is NULL");
  if (int rc = check_common(a->b, a->h, a->h_k, a->d, a->dtype, a->softcap)) return rc;
  int32_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  C5Plan pl;
  if (!a->cu_seqlens_q && bwd_c5_plan(a, pl)) {
    v[0] = 5; v[1] = pl.n_chunks; v[2] = pl.rounds_per_chunk; v[3] = pl.head_tiles; v[4] = pl.np64; v[5] = pl.c1; v[6] = pl.jb; v[7] = (int32_t)(pl.slot_bytes >> 20);
  } else if (!a->cu_seqlens_q && bwd_fused_ds_bytes(a) > 0) {
    v[0] = 3;
    FusedPlan fpl; bwd_fused_plan(a, fpl); v[1] = fpl.n_chunks; v[2] = fpl.nb; v[7] = (int32_t)((fpl.ds_bytes + fpl.sync_bytes) >> 20);
  }
  if (v[0] == 0) { GsplitPlan gp; if (bwd_gsplit_plan(a, gp)) { v[3] = gp.gs; v[7] = (int32_t)(gp.bytes >> 20); } }   // (the pair: its dK/dV half on a split GQA group)
  for (int i = 0; i < n && i < 8; ++i) out[i] = v[i];
  return 8;
}

int fa_fwd(const FaFwdParams* params, void* stream) { return do_fwd(params, stream, false); }
int fa_varlen_fwd(const FaFwdParams* params, void* stream) { return do_fwd(params, stream, true); }
int fa_fwd_kvcache(const FaFwdParams* params, void* stream) { return do_fwd(params, stream, false, true); }

int fa_kvcache_append(const FaKvAppendParams* a, void* stream) {
  if (!a) return fail(FA_ERR_INVALID_ARGUMENT, "params is NULL");
  g_err[0] = 0;
  if (!a->knew || !a->vnew || !a->kcache || !a->vcache) return fail(FA_ERR_INVALID_ARGUMENT, "knew, vnew, kcache and vcache must be non-NULL");
  if (a->d <= 0 || a->d % 8 != 0) return fail(FA_ERR_INVALID_ARGUMENT, "head dimension must be a multiple of 8");
  if (a->dtype != FA_DTYPE_FP16 && a->dtype != FA_DTYPE_BF16) return fail(FA_ERR_INVALID_ARGUMENT, "FlashAttention only supports fp16 and bf16 data type");
  if (a->block_table && (a->page_block_size <= 0 || a->page_block_size % 256 != 0))
    return fail(FA_ERR_INVALID_ARGUMENT, "Paged KV cache block size must be divisible by 256");
  fa::KvAppendK k{};
  k.knew = a->knew; k.vnew = a->vnew; k.kcache = a->kcache; k.vcache = a->vcache;
  k.kn_bs = a->knew_batch_stride; k.kn_rs = a->knew_row_stride; k.kn_hs = a->knew_head_stride;
  k.vn_bs = a->vnew_batch_stride; k.vn_rs = a->vnew_row_stride; k.vn_hs = a->vnew_head_stride;
  k.kc_bs = a->kcache_batch_stride; k.kc_rs = a->kcache_row_stride; k.kc_hs = a->kcache_head_stride;
  k.vc_bs = a->vcache_batch_stride; k.vc_rs = a->vcache_row_stride; k.vc_hs = a->vcache_head_stride;
  k.seqlens_k = a->seqlens_k; k.kv_batch_idx = a->cache_batch_idx; k.block_table = a->block_table;
  k.block_table_bs = a->block_table_batch_stride; k.page_size = a->page_block_size;
  k.b = a->b; k.s_new = a->seqlen_new; k.h_k = a->h_k; k.d = a->d;
  if (fa::launch_kv_append(k, (hipStream_t)stream) != 0)
    return fail(FA_ERR_LAUNCH, "kv append launch failed: %s", hipGetErrorString(hipGetLastError()));
  return FA_OK;
}

int fa_rotary(const FaRotaryParams* a, void* stream) {
  if (!a) return fail(FA_ERR_INVALID_ARGUMENT, "params is NULL");
  g_err[0] = 0;
  if (!a->x || !a->y || !a->cos || !a->sin) return fail(FA_ERR_INVALID_ARGUMENT, "x, y, cos and sin must be non-NULL");
  if (a->dtype != FA_DTYPE_FP16 && a->dtype != FA_DTYPE_BF16) return fail(FA_ERR_INVALID_ARGUMENT, "FlashAttention only supports fp16 and bf16 data type");
  if (a->d <= 0 || a->d % 8 != 0) return fail(FA_ERR_INVALID_ARGUMENT, "head dimension must be a multiple of 8");
  if (a->rotary_dim <= 0 || a->rotary_dim > a->d) return fail(FA_ERR_INVALID_ARGUMENT, "rotary_dim must be <= headdim");
  if (a->rotary_dim % 16 != 0) return fail(FA_ERR_INVALID_ARGUMENT, "Only rotary dimensions divisible by 16 are currently supported");
  if (a->seqlen_ro <= 0) return fail(FA_ERR_INVALID_ARGUMENT, "cos/sin must have at least one row");
  fa::RotaryK k{};
  k.x = a->x; k.y = a->y; k.cos = a->cos; k.sin = a->sin; k.offsets = a->seqlen_offsets;
  k.x_bs = a->x_batch_stride; k.x_rs = a->x_row_stride; k.x_hs = a->x_head_stride;
  k.y_bs = a->y_batch_stride; k.y_rs = a->y_row_stride; k.y_hs = a->y_head_stride; k.cos_rs = a->cos_row_stride;
  k.b = a->b; k.s = a->s; k.h = a->h; k.d = a->d; k.rotary_dim = a->rotary_dim; k.seqlen_ro = a->seqlen_ro;
  k.interleaved = a->interleaved; k.per_token = a->per_token;
  if (fa::launch_rotary(k, a->dtype == FA_DTYPE_BF16, (hipStream_t)stream) != 0)
    return fail(FA_ERR_LAUNCH, "rotary launch failed: %s", hipGetErrorString(hipGetLastError()));
  return FA_OK;
}

int fa_set_rng_state(uint64_t seed, uint64_t offset, uint64_t* rng_state, void* stream) {
  g_err[0] = 0;
  if (!rng_state) return fail(FA_ERR_INVALID_ARGUMENT, "rng_state is NULL");
  if (fa::launch_set_rng(seed, offset, rng_state, (hipStream_t)stream) != 0)
    return fail(FA_ERR_LAUNCH, "rng_state launch failed: %s", hipGetErrorString(hipGetLastError()));
  return FA_OK;
}

int64_t fa_fwd_workspace_bytes(const FaFwdParams* params) {
  if (!params) return 0;
  if (params->cu_seqlens_q) {  // varlen forward: the work list of an uneven packed batch
    int causal = params->is_causal, wl = params->window_left, wr = params->window_right;
    normalize_window(params->seqlen_q, params->seqlen_k, params->alibi_slopes != nullptr, causal, wl, wr);
    const int nw = params->d > 128 ? 4 : fwd_schedule_nw(params, wl, wr);
    const int64_t entries = varlen_list_entries(params, fwd_block_rows(params, nw, false));
    return entries > 0 ? (entries + 1) * 8 : 0;
  }
  int split_tiles = 0;
  return splitkv_bytes(params, choose_splits(params, split_tiles));
}

int64_t fa_bwd_workspace_bytes(const FaBwdParams* params) {
  // no fp32 dq accumulator; an uneven packed batch gets work lists, a fixed-length batch the dS workspace of the 5-contraction path
  if (!params) return 0;
  int64_t qe, ke;
  bwd_list_entries(params, qe, ke);
  if (const int64_t fz = bwd_fused_ds_bytes(params); fz > 0) return fz + bwd_fused_sync_bytes(params);
  if (C5Plan pl; bwd_c5_plan(params, pl)) return 2 * pl.slot_bytes;
  if (GsplitPlan gp; bwd_gsplit_plan(params, gp)) return gp.bytes;   // (a split GQA group's partial dK / dV; optional like the dS area: without it the unsplit kernels run)
  return (qe ? (qe + 1) * 8 : 0) + (ke ? (ke + 1) * 8 : 0);   // (work lists: varlen only)
}
int fa_bwd(const FaBwdParams* params, void* stream) { return do_bwd(params, stream, false); }
int fa_bwd_fused_status(const FaBwdParams* params, void* stream) {
  if (!params || !params->workspace || fa::last_schedule().bwd_spill != 3) return FA_OK;   // (this thread's last backward did not take the fused path: the sync area was never initialised)
  // Reading the flag synchronises the stream.  Where the fused backward is the DEFAULT (round 6: the table of bwd_fused_by_table) the binders' call returns at once --
  // a host sync per backward is not acceptable there, and the flag guards a wait that cannot time out while the launch's workgroups run (the publisher a consumer
  // polls for is between two of its own instructions); FA_BWD_MODE=3 (the opt-in form) and FA_BWD_FUSED_CHECK=1 read it.
  if (fa::knobs().bwd_fused_check < 0 || (fa::knobs().bwd_mode != 3 && !fa::knobs().bwd_fused_check)) return FA_OK;   // (-1: never, also for FA_BWD_MODE=3 -- timing the launch without the sync)
  const int64_t fz = bwd_fused_ds_bytes(params);
  if (fz <= 0 || params->workspace_bytes < fz + bwd_fused_sync_bytes(params)) return FA_OK;   // the call did not (could not) take the fused path
  int32_t flag = 0;
  if (hipMemcpyAsync(&flag, (const char*)params->workspace + fz + fa::FZ_ERR * 4, sizeof(flag), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
      hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
    return fail(FA_ERR_LAUNCH, "fused backward: could not read the launch's error flag: %s", hipGetErrorString(hipGetLastError()));
  if (flag != 0) return fail(FA_ERR_LAUNCH, "fused backward (FA_BWD_MODE=3): a dQ hand-off timed out, a block of dq was not written -- discard this call's gradients");
  return FA_OK;
}
int fa_varlen_bwd(const FaBwdParams* params, void* stream) { return do_bwd(params, stream, true); }

}  // extern "C"
