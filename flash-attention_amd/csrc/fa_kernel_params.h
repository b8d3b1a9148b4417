// Device-side parameter blocks (passed by value to the kernels).  Host code fills them from the
// public C-ABI structs of include/fa_gfx950.h after validation/normalisation (fa_api.cpp).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace fa {

struct FwdK {
  const void* q;
  const void* k;
  const void* v;
  void* o;
  float* lse;
  int64_t q_bs, q_rs, q_hs;
  int64_t k_bs, k_rs, k_hs;
  int64_t v_bs, v_rs, v_hs;
  int64_t o_bs, o_rs, o_hs;
  const int32_t* cu_q;       // nullptr => fixed length
  const int32_t* cu_k;
  const int32_t* seqused_k;  // optional: keys in use per batch entry (KV cache: cache_seqlens)
  const int32_t* seqused_q;  // optional: queries in use per batch entry (padded batches: rows past it are neither read nor written)
  const int32_t* kv_batch_idx;   // optional: batch entry -> row of the KV cache (cache_batch_idx)
  const int32_t* block_table;    // optional: paged KV cache, (b, max_blocks) page indices
  int64_t block_table_bs;
  int32_t page_size;             // keys per page (multiple of 64)
  int32_t seqused_add;           // added to seqused_k (freshly appended keys)
  const int32_t* leftpad_k;      // optional: first cache row of each batch entry (seqused_k counts from row 0)
  const float* alibi;        // optional
  int64_t alibi_bs;
  int32_t b, h, h_k, hk_ratio;
  int32_t sq, sk;            // fixed: exact; varlen: max
  int32_t total_q;
  int32_t nmb;               // query blocks per sequence (grid sizing)
  int32_t n_units, unit_size, unit_hpx;  // XCD work mapping (fa_device.h xcd_interleave); grid = 8*ceil(n_units/8)*unit_size
  int32_t wl, wr;            // normalised window, < 0 = unbounded
  float scale;               // softmax_scale
  float scale_log2;          // softmax_scale * log2(e)
  float softcap;             // 0 = off
  float rescale_thr;         // O rescale deferred until a row max grows by more than this (log2 units)
  int32_t pack_g;            // >= 1.  g > 1 (fa_fwd_kernel, KV-cache path): the g query heads of a KV group are packed into the rows: h = h_k,
                             // hk_ratio = 1, sq = g * (true query count), row r = query r / g of query head head * g + r % g
  int32_t d_chunks;          // fa_fwd_kernel: > 0 => only the first d_chunks 16-byte chunks of a row exist in memory (head dim = 8 * d_chunks <
                             // the kernel's DV); the chunks behind them are read as zeros and never stored (KV-cache path, head dims between
                             // the built sizes: a cache cannot be padded on the fly)
  int32_t persist_total;     // fa_fwd_w64: > 0 => persistent launch, blocks 0 .. persist_total-1 are walked by gridDim.x workgroups
  int32_t work_bound;        // entries the list can hold (grid = work_bound * h)
  const int2* work_list;     // varlen: {count,0}, then {batch, query block} pairs, heaviest first (nullptr => dense grid)
  // split-KV (decode): n_splits > 1 => workgroup (.., split) scans key tiles [split*split_tiles, (split+1)*split_tiles) and
  // writes a normalised fp32 partial output + its log-sum-exp; fa_splitkv_combine_kernel merges them
  int32_t n_splits, split_tiles;
  float* o_accum;            // (n_splits, b, h, sq, d) fp32
  float* lse_accum;          // (n_splits, b, h, sq) fp32, -inf for an empty partial
  // dropout (rng == nullptr => off): element (b, h, i, j) is kept iff its random byte <= drop_thr8 (fa_device.h drop_bytes: Philox2x32-7 keyed per (batch, head))
  const uint64_t* rng;       // device {seed, offset}
  uint8_t* randval;          // optional: random bytes out
  int64_t rv_bs, rv_hs, rv_rs;
  uint32_t drop_thr8;        // floor(255 * (1 - p_dropout))
  float rp_keep;             // 1 / (1 - p_dropout)
};

// Fused backward (BwdK::fuse_sync, fa_bwd.hip fa_bwd_fused_kernel): int32 words of the sync area.  An error flag, 16 words reserved for the cycle statistics of
// experiments/ablations/fa_bwd.patch, then eight control blocks -- one per XCD x, whose key-block items are those numbered x + 8k: the ready queue's tail, head and count of published
// and unclaimed dQ items, the count of key-block items finished, the next key-block item to hand out, a 128-byte line each -- then one arrival counter per
// dQ item (batch, head, 256-query block), BwdK::fuse_line words apart (32 = a line each), then the eight queues (item + 1; 0 = not published yet),
// fuse_items words each.  Zeroed before every launch.
enum { FZ_ERR = 0, FZ_STATS = 32, FZ_CTRL = 64, FZ_CTRL_STRIDE = 160, FZ_TAIL = 0, FZ_HEAD = 32, FZ_AVAIL = 64, FZ_DONE = 96, FZ_NEXT = 128, FZ_COUNTERS = FZ_CTRL + 8 * FZ_CTRL_STRIDE };
inline constexpr long long fz_sync_words(long long items, int line) { return FZ_COUNTERS + items * line + 8 * items; }

struct BwdK {
  const void* dout;
  const void* q;
  const void* k;
  const void* v;
  const void* o;
  const float* lse;
  void* dq;
  void* dk;
  void* dv;
  float* delta;              // softmax_d
  int64_t do_bs, do_rs, do_hs;
  int64_t q_bs, q_rs, q_hs;
  int64_t k_bs, k_rs, k_hs;
  int64_t v_bs, v_rs, v_hs;
  int64_t o_bs, o_rs, o_hs;
  int64_t dq_bs, dq_rs, dq_hs;
  int64_t dk_bs, dk_rs, dk_hs;
  int64_t dv_bs, dv_rs, dv_hs;
  const int32_t* cu_q;
  const int32_t* cu_k;
  const int32_t* seqused_q;  // optional: queries / keys in use per batch entry (padded batches), as FwdK
  const int32_t* seqused_k;
  const float* alibi;
  int64_t alibi_bs;
  int32_t b, h, h_k, hk_ratio;
  int32_t sq, sk;
  int32_t total_q, total_k;
  int32_t nmb, nnb;          // query / key blocks per sequence
  int32_t q_units, q_unit_size, q_unit_hpx;  // XCD work mapping of the dQ kernel
  int32_t k_units, k_unit_size, k_unit_hpx;  // XCD work mapping of the dK/dV kernel
  int32_t wl, wr;
  float scale;
  float scale_log2;
  float softcap;
  int32_t q_bound, k_bound;
  const int2* q_list;        // varlen work lists of the dQ kernel (query blocks) and the dK/dV kernel (key blocks)
  const int2* k_list;
  const uint64_t* rng;       // dropout, as in FwdK
  uint32_t drop_thr8;
  float rp_keep;
  // dS spill (5-contraction backward): the dK/dV kernel writes every dS sub-tile (32 queries x 32 keys, 2 KB, rounded to the
  // input dtype) here and the dQ pass is one contraction dS.K; NULL = the dQ kernel recomputes S, dP and dS (7 contractions)
  void* ds_ws;               // [b][h][ds_nq32][ds_nk32][2 KB]
  int32_t ds_nq32, ds_nk32;  // 32-row / 32-key blocks per sequence
  int32_t d_chunks;          // > 0 => only the first d_chunks 16-byte chunks of a row exist in memory (head dim = 8 * d_chunks < the kernels' DV, a head
                             // dim between the built sizes): the chunks behind them are read as zeros and never stored (as FwdK::d_chunks); the
                             // 4-wave dQ kernel, the dK/dV kernel and the delta pre-pass take it
  int32_t dq_nw;             // dQ schedule: 4 / 8 waves x 32 rows, 64 = 4 waves x 64 rows (nmb and the query work list are sized for it)
  int32_t* fuse_sync;        // fused backward (FA_BWD_MODE=3): sync area behind the dS workspace (FZ_* above), NULL otherwise
  int32_t fuse_items;        // dQ items of the fused backward = b * h * nmb (256-row blocks)
  int32_t fuse_line;         // words between two arrival counters of the sync area
  int32_t fuse_total;        // key-block items of the fused backward (= the dK/dV kernel's grid)
  int32_t kv_in_shift;       // dK/dV kernels: K / V are read from kv head (hk >> kv_in_shift) while dK / dV, the head walk and everything else use hk -- a GQA group split into
                             // 2^kv_in_shift virtual kv heads whose partial dK / dV the host sums afterwards (fa_api.cpp bwd_gsplit_plan; 0 = off)
  int32_t fuse_keep_err;     // host side: this launch is not the call's first chunk -- its memset leaves the error flag (word FZ_ERR = 0) alone
  // 5-contraction backward (round 6, fa_bwd_dkdv_w64.hip fa_bwd_c5_kernel): one launch = the dK/dV items of one chunk of (batch, kv head) units, which write dS to the
  // slot ds_ws, followed by the dQ = dS.K items of the chunk BEFORE it, which read the slot ds_rd the previous launch filled (stream order is the hand-off).  Units are
  // dealt to the XCDs in rounds of eight (k_units / k_unit_size / k_unit_hpx); a chunk is a range of rounds.
  const void* ds_rd;         // slot read by this launch's dQ items
  uint32_t c5_slot_bytes;    // bytes of a slot (range of the buffer descriptors)
  int32_t c5_np, c5_nc;      // dK/dV items and dQ items of this launch
  int32_t c5_pbid0;          // number of the launch's first dK/dV item in the whole dK/dV grid
  int32_t c5_pj0, c5_cj0;    // first round of the dK/dV items' chunk / of the dQ items' chunk
  int32_t ds_np64;           // fused backward (round 6): 64-key pairs per packed row of the dS workspace (ds_nk32 stays the number of 32-key sub-tiles of a sequence)
  int32_t ds_c1, ds_jb, ds_head_tiles;   // row packing of the workspace (fa_device.h ds_row_start(i, ds_c1, ds_jb, ds_nk32): on this path ds_nk32 counts 64-key PAIRS of sub-tiles); sub-tiles per head = ds_head_tiles
  int32_t fuse_delta;        // 64-rows-per-wave dQ kernel only: 1 = compute softmax_d = rowsum(dO * O) of its own rows in the prologue (and write it for the
                             // dK/dV kernel, which is then launched BEHIND the dQ kernel); 0 = read it (fa_bwd_delta_kernel ran first)
};

}  // namespace fa
