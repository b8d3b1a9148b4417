/*
 * fa_gfx950.h -- C ABI of the MI355X (gfx950 / CDNA4) fused-attention library `libfa_gfx950.so`.
 *
 * This is the drop-in boundary of the hot path: plain pointers, sizes and element strides, no
 * torch / ATen types.  Each entry point replaces one host function of the reference's backend
 * module `flash_attn_2_cuda` (the module `flash_attn/flash_attn_interface.py:13-23` imports):
 *
 *   fa_fwd         <->  mha_fwd         csrc/flash_attn/flash_api.cpp:368-536   (pybind "fwd",        :1537)
 *   fa_varlen_fwd  <->  mha_varlen_fwd  csrc/flash_attn/flash_api.cpp:538-788   (pybind "varlen_fwd", :1538)
 *   fa_bwd         <->  mha_bwd         csrc/flash_attn/flash_api.cpp:800-1008  (pybind "bwd",        :1539)
 *   fa_varlen_bwd  <->  mha_varlen_bwd  csrc/flash_attn/flash_api.cpp:1010-1241 (pybind "varlen_bwd", :1540)
 *   fa_kvcache_append + fa_fwd_kvcache  <->  mha_fwd_kvcache  csrc/flash_attn/flash_api.cpp:1243-1532 (pybind "fwd_kvcache", :1541)
 *
 * The two parameter blocks below are this library's counterpart of the reference's kernel POD
 * `Flash_fwd_params` / `Flash_bwd_params` (csrc/flash_attn/src/flash.h:47-189).  The caller owns
 * every buffer (allocation stays with the framework's caching allocator); the library only
 * enqueues kernels on `stream` and never synchronises.  The torch extension
 * `flash_attn_2_cuda` (flash-attention_amd/csrc/torch_binding.cpp) and the ctypes loader
 * (flash-attention_amd/flash_attn_amd/_cabi.py) are the two in-tree binders; INTEGRATION.md shows
 * the stub a reference maintainer would add.
 *
 * Conventions
 *   - all strides are in ELEMENTS (not bytes); the last (head-dim) stride is 1;
 *   - q (B,Sq,H,D), k/v (B,Sk,Hk,D), o like q; varlen: q (total_q,H,D), k/v (total_k,Hk,D) and
 *     batch strides are ignored; sequence b owns rows cu_seqlens[b] .. cu_seqlens[b+1]-1
 *     (reference csrc/flash_attn/src/block_info.h:12-45);
 *   - softmax_lse is fp32 (B,H,Sq) or, varlen, (H,total_q); natural log of sum_j exp(scale*s_ij);
 *     rows without any visible key give out = 0, lse = +inf (reference softmax.h:179-180);
 *   - window_left / window_right < 0 mean unbounded; is_causal forces window_right = 0; masks are
 *     aligned to the bottom-right corner (reference flash_attn_interface.py:1175-1189);
 *   - head dim d: any multiple of 8 up to 256, forward and backward.  Kernels are built for 32 / 64 / 96 / 128 / 192 / 256 (the reference's
 *     set, static_switch.h:92-110); a size in between runs the next built size's kernels with a run-time column bound: the 16-byte chunks
 *     behind d are read as zeros and never stored, so q / k / v / out / gradients keep their own row pitch -- no padded copies (the reference
 *     rounds internally the same way, flash_api.cpp:458,872);
 *   - return value 0 = enqueued; negative = FA_ERR_* (message via fa_last_error()).
 */
#ifndef FA_GFX950_H_
#define FA_GFX950_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FA_ABI_VERSION 6

enum { FA_DTYPE_FP16 = 0, FA_DTYPE_BF16 = 1 };

enum {
  FA_OK = 0,
  FA_ERR_INVALID_ARGUMENT = -1, /* shape/dtype/stride contract violated                     */
  FA_ERR_UNSUPPORTED = -2,      /* valid in the reference but not built here (see message)   */
  FA_ERR_LAUNCH = -3,           /* hipLaunchKernel failed                                     */
  FA_ERR_WORKSPACE = -4         /* workspace missing or too small                             */
};

typedef struct FaFwdParams {
  const void* q;
  const void* k;
  const void* v;
  void* o;
  float* softmax_lse;           /* (B,H,Sq) or varlen (H,total_q) */
  int64_t q_batch_stride, q_row_stride, q_head_stride;
  int64_t k_batch_stride, k_row_stride, k_head_stride;
  int64_t v_batch_stride, v_row_stride, v_head_stride;
  int64_t o_batch_stride, o_row_stride, o_head_stride;
  const int32_t* cu_seqlens_q;  /* NULL => fixed-length batch          */
  const int32_t* cu_seqlens_k;  /* NULL => fixed-length batch          */
  const int32_t* seqused_k;     /* optional (B): keys actually used     */
  const float* alibi_slopes;    /* optional (H) or (B,H) fp32           */
  int64_t alibi_batch_stride;   /* 0 when slopes are (H)                */
  int32_t b, h, h_k, d;
  int32_t seqlen_q, seqlen_k;   /* fixed: exact; varlen: max_seqlen_*   */
  int32_t total_q;              /* varlen: rows of q; fixed: b*seqlen_q */
  int32_t dtype;                /* FA_DTYPE_*                           */
  int32_t is_causal;
  int32_t window_left, window_right;
  float softmax_scale;
  float softcap;                /* 0 = off                              */
  int32_t seqused_k_add;        /* added to seqused_k[b] (keys appended in the same call)            */
  /* KV-cache extensions (NULL / 0 otherwise) */
  const int32_t* cache_batch_idx;   /* fa_fwd_kvcache, optional (B): batch entry -> row of the cache */
  const int32_t* block_table;       /* fa_fwd_kvcache / fa_varlen_fwd, optional (B, max_blocks) int32: paged K/V, k/v are
                                       (num_blocks, page, Hk, D) and k/v_batch_stride is the page stride */
  int64_t block_table_batch_stride;
  int32_t page_block_size;          /* keys per page, multiple of 256 (reference flash_api.cpp:1318)  */
  int32_t num_splits;               /* fa_fwd_kvcache: 0 = heuristic, 1 = no split, >1 = split the keys this many ways */
  float p_dropout;                  /* probability to DROP, in [0, 1); 0 = off                         */
  int32_t reserved0;
  /* dropout (p_dropout > 0): */
  const uint64_t* rng_state;        /* device, 2 x u64 {seed, offset} (reference rng_state, flash_api.cpp:496-515) */
  uint8_t* randval;                 /* optional out: the random byte of every (query, key) pair; a pair is KEPT iff
                                       byte <= floor(255*(1-p_dropout)) (the ROCm backend's return_softmax payload,
                                       csrc/flash_attn_ck/mha_fwd.cpp:275-279); (B,H,Sq,Sk) or varlen (H,total_q,max_seqlen_k) */
  int64_t randval_batch_stride, randval_head_stride, randval_row_stride;
  /* split-KV scratch (fa_fwd_kvcache): fa_fwd_workspace_bytes() bytes, 256-B aligned; may be NULL if that is 0 */
  void* workspace;
  int64_t workspace_bytes;
  const int32_t* leftpad_k;         /* fa_varlen_fwd / fa_fwd_kvcache, optional (B): the keys of entry b start at row leftpad_k[b];
                                       lengths (seqused_k / cu_seqlens_k) count from row 0 (reference block_info.h:17-36);
                                       not together with block_table */
  const int32_t* seqused_q;         /* ABI v6, fa_fwd / fa_varlen_fwd, optional (B): only the first seqused_q[b] query rows of entry b exist
                                       (the rest of its rows -- cu_seqlens_q[b+1] - cu_seqlens_q[b], or seqlen_q -- are padding: never
                                       read, their o / softmax_lse rows never written).  With cu_seqlens_q[b] = b*S + first valid row and
                                       seqused_q / seqused_k = the valid lengths, a PADDED (B,S,H,D) batch runs in place: the gather /
                                       scatter passes of the reference's unpad_input / pad_input (flash_attn/bert_padding.py:98-128,
                                       204-218) disappear */
} FaFwdParams;

/* Append step of the KV-cache path: copy knew/vnew (B, S_new, Hk, D) into the cache at rows
 * seqlens_k[b] .. seqlens_k[b]+S_new-1 of cache row cache_batch_idx[b] (or page-table addressed). */
typedef struct FaKvAppendParams {
  const void* knew;
  const void* vnew;
  void* kcache;
  void* vcache;
  int64_t knew_batch_stride, knew_row_stride, knew_head_stride;
  int64_t vnew_batch_stride, vnew_row_stride, vnew_head_stride;
  int64_t kcache_batch_stride, kcache_row_stride, kcache_head_stride;
  int64_t vcache_batch_stride, vcache_row_stride, vcache_head_stride;
  const int32_t* seqlens_k;         /* (B) current lengths; NULL => append at row 0                  */
  const int32_t* cache_batch_idx;   /* optional                                                       */
  const int32_t* block_table;       /* optional (paged cache)                                         */
  int64_t block_table_batch_stride;
  int32_t page_block_size;
  int32_t b, seqlen_new, h_k, d;
  int32_t dtype;
  int32_t reserved[2];
} FaKvAppendParams;

/* Rotary position embedding of x (B, S, H, D) into y (same shape, may alias x): the first rotary_dim channels of
 * every row are rotated by the angle row of position seqlen_offsets[b] + (per_token ? s : 0)
 * (reference flash_fwd_kernel.h:640-720 Append_KV rotary, csrc/flash_attn/src/rotary.h; Python
 * flash_attn/layers/rotary.py apply_rotary_emb).  cos/sin: (seqlen_ro, rotary_dim/2), same dtype as x. */
typedef struct FaRotaryParams {
  const void* x;
  void* y;
  const void* cos;
  const void* sin;
  const int32_t* seqlen_offsets;   /* (B) or NULL (= 0) */
  int64_t x_batch_stride, x_row_stride, x_head_stride;
  int64_t y_batch_stride, y_row_stride, y_head_stride;
  int64_t cos_row_stride;          /* elements between consecutive positions of cos / sin */
  int32_t b, s, h, d;
  int32_t rotary_dim;              /* multiple of 16, <= d */
  int32_t seqlen_ro;               /* rows of cos / sin */
  int32_t interleaved;             /* 1: pairs (2t, 2t+1) (GPT-J); 0: pairs (t, t + rotary_dim/2) (GPT-NeoX) */
  int32_t per_token;               /* 1: row s sits at position offset + s; 0: every row at position offset */
  int32_t dtype;
  int32_t reserved[3];
} FaRotaryParams;

typedef struct FaBwdParams {
  const void* dout;
  const void* q;
  const void* k;
  const void* v;
  const void* o;
  const float* softmax_lse;     /* as written by fa_fwd                */
  void* dq;
  void* dk;
  void* dv;
  float* softmax_d;             /* out: rowsum(dO*O), fp32, same shape/indexing as softmax_lse */
  void* workspace;              /* fa_bwd_workspace_bytes() bytes, 256-B aligned, may be NULL if that is 0 */
  int64_t workspace_bytes;
  int64_t do_batch_stride, do_row_stride, do_head_stride;
  int64_t q_batch_stride, q_row_stride, q_head_stride;
  int64_t k_batch_stride, k_row_stride, k_head_stride;
  int64_t v_batch_stride, v_row_stride, v_head_stride;
  int64_t o_batch_stride, o_row_stride, o_head_stride;
  int64_t dq_batch_stride, dq_row_stride, dq_head_stride;
  int64_t dk_batch_stride, dk_row_stride, dk_head_stride;
  int64_t dv_batch_stride, dv_row_stride, dv_head_stride;
  const int32_t* cu_seqlens_q;
  const int32_t* cu_seqlens_k;
  const float* alibi_slopes;
  int64_t alibi_batch_stride;
  int32_t b, h, h_k, d;
  int32_t seqlen_q, seqlen_k;
  int32_t total_q, total_k;
  int32_t dtype;
  int32_t is_causal;
  int32_t window_left, window_right;
  float softmax_scale;
  float softcap;
  int32_t deterministic;        /* accepted; this implementation is always deterministic */
  float p_dropout;              /* as in the forward call                                  */
  int32_t reserved[3];
  const uint64_t* rng_state;    /* device {seed, offset} the forward used (p_dropout > 0) */
  const int32_t* seqused_q;     /* ABI v6, optional (B): as FaFwdParams::seqused_q; dq rows past it are not written */
  const int32_t* seqused_k;     /* ABI v6, optional (B): keys of entry b in use.  In the backward it can only SHORTEN an entry: the length is
                                   min(seqused_k[b], the cu_seqlens_k length) (fixed-length: min(seqused_k[b], seqlen_k)) -- the key-block work
                                   list is sized from cu_seqlens_k, so a longer value could not be honoured by every kernel (round 5).  Round 6: the
                                   forward applies the same rule inside a packed batch (cu_seqlens_k without a block table): min(seqused_k[b] +
                                   seqused_k_add, the cu_seqlens_k length); only against a KV cache (no cu_seqlens_k, or a paged one) does
                                   seqused_k REPLACE the length.  A forward / backward pair therefore always sees the same keys; dk / dv rows
                                   past it are not written */
} FaBwdParams;

/* ABI version of the loaded library (== FA_ABI_VERSION of the header it was built from). */
int fa_abi_version(void);
/* sizeof() of the two parameter blocks as the library sees them (binder self-check). */
int fa_sizeof_fwd_params(void);
int fa_sizeof_bwd_params(void);
int fa_sizeof_kvappend_params(void);
int fa_sizeof_rotary_params(void);
/* Last error message of the calling thread ("" if none). */
const char* fa_last_error(void);

/* Run-time knobs (FA_FWD_NW, FA_RESCALE_THR, FA_VARLEN_LIST, FA_IL_SCHED, FA_BWD_DQ_NW; INTEGRATION.md "Run-time knobs")
 * are read from the environment once per process; this re-reads them (tests / A-B tools after changing the environment). */
void fa_knobs_reload(void);
/* Which kernels the calling thread's last fa_fwd* / fa_bwd* call enqueued (for tests and the benchmark's labels; the
 * reference exposes nothing comparable -- its dispatch is compile-time, flash_fwd_launch_template.h).  Fills up to n of
 * FA_SCHEDULE_FIELDS int32: {forward kernel id (0 none, 1 lock-step fa_fwd_kernel, 2 pipelined fa_fwd_il_kernel,
 * 3 64-rows-per-wave fa_fwd_w64_kernel), waves per workgroup (16 = 8-wave ping-pong), feature variant, key splits,
 * varlen work list used, head dim, bf16, dQ-kernel waves, backward work lists used, backward spilled dS (5 contractions),
 * query heads packed into the rows of a block (fa_fwd_kvcache, 1 = none), dK/dV schedule (8 waves x 32 keys, 4 at head dim 256,
 * or 64 = 4 waves x 64 keys)}; returns FA_SCHEDULE_FIELDS (fields are only ever appended). */
#define FA_SCHEDULE_FIELDS 12
int fa_last_schedule(int32_t* out, int n);
/* Name of the forward kernel instantiation of that call, e.g. "fa::fa_fwd_il_kernel<bf16,128,4,3>" ("" if none). */
const char* fa_last_kernel_name(void);
/* Host logic only (no launch, no pointer is dereferenced -- pointers count as flags: alibi_slopes, block_table, cu_seqlens_q, seqused_q): the forward
 * schedule fa_fwd (varlen = 0) / fa_varlen_fwd (varlen = 1) would run for these parameters, as an FA_FWD_NW code -- 64 = 64-rows-per-wave kernel,
 * 34 / 38 = software-pipelined kernel with 4 / 8 waves, 4 / 8 / 16 = lock-step kernel -- after the feature / head-dim / head-packing fallbacks; and the dQ
 * schedule of the backward (4 / 8 waves x 32 rows, 64 = 4 waves x 64 rows).  Negative = FA_ERR_*.  For tests of the dispatch on a box without a GPU. */
int fa_fwd_schedule_query(const FaFwdParams* params, int varlen);
int fa_bwd_dq_schedule_query(const FaBwdParams* params);
/* Which backward a fixed-length call runs and how its workspace is laid out (round 6; host logic only, for tests of the dispatch on a box without a GPU):
 * out[0] = 0 the recomputing pair (7 contractions, no dS workspace), 3 = the fused launch (dK/dV + dQ = dS.K, FA_BWD_MODE=3 or the default table; out[1] = launches
 * = chunks of whole batch entries, out[2] = batch entries per chunk, out[7] = MiB of workspace),
 * 5 = the chunked 5-contraction backward (FA_BWD_MODE=5); for 5: out[1] = chunks, out[2] = XCD rounds (8 units) per chunk, out[3] = dS sub-tiles (2 KB) per
 * head with packed rows, out[4] = 64-key pairs per row, out[5] / out[6] = the row packing's a / jb (csrc/fa_device.h ds_row_start), out[7] = MiB per slot.
 * For 0 (the pair): out[3] = virtual kv heads per GQA group when the dK/dV kernels split the groups to fill the chip (0 = unsplit), out[7] = MiB of workspace for the partials.
 * Returns the number of fields (8) or a negative FA_ERR_*. */
int fa_bwd_plan_query(const FaBwdParams* params, int32_t* out, int n);

/* Forward, fixed-length batch.  cu_seqlens_* must be NULL.  `stream` is a hipStream_t. */
int fa_fwd(const FaFwdParams* params, void* stream);
/* Forward, packed variable-length batch.  cu_seqlens_q/k must be non-NULL device int32 (b+1). */
int fa_varlen_fwd(const FaFwdParams* params, void* stream);
/* Inference forward against a KV cache: fa_fwd plus seqused_k (cache_seqlens), cache_batch_idx and/or a
 * paged cache (block_table).  k/v point at the cache.  No backward. */
int fa_fwd_kvcache(const FaFwdParams* params, void* stream);
/* Writes the new keys/values into the cache (call before fa_fwd_kvcache with seqused_k_add = seqlen_new). */
int fa_kvcache_append(const FaKvAppendParams* params, void* stream);
/* Writes {seed, offset} to a device rng_state (2 x u64) in stream order (binder helper for p_dropout > 0). */
int fa_set_rng_state(uint64_t seed, uint64_t offset, uint64_t* rng_state, void* stream);
/* Bytes of split-KV scratch fa_fwd_kvcache needs for this problem with params->num_splits (0 is possible). */
int64_t fa_fwd_workspace_bytes(const FaFwdParams* params);
/* Rotary embedding of q / new keys ahead of fa_kvcache_append + fa_fwd_kvcache (y may alias x). */
int fa_rotary(const FaRotaryParams* params, void* stream);
/* Bytes of scratch the backward can use for this problem (0 is possible).  For fa_bwd (fixed-length batches) the scratch is the dS area of the 5-contraction
 * launches (up to 1.25 GiB by default; FA_BWD_MODE / FA_BWD_DS_CAP_MB / FA_BWD_C5_CAP_MB): a speed-up, not a requirement -- called with workspace = NULL or fewer
 * bytes, fa_bwd runs the recomputing pair, which needs none (both binders do exactly that when their allocator is out of memory).  For fa_varlen_bwd it holds the
 * work lists of an uneven batch (a few KB). */
int64_t fa_bwd_workspace_bytes(const FaBwdParams* params);
/* Backward, fixed-length batch: writes dq, dk, dv (caller-allocated) and softmax_d. */
int fa_bwd(const FaBwdParams* params, void* stream);
/* Backward, packed variable-length batch. */
int fa_varlen_bwd(const FaBwdParams* params, void* stream);
/* Opt-in fused backward (FA_BWD_MODE=3) only: after fa_bwd with the SAME params, waits for `stream` and reads the launch's error flag from the workspace.
   0 = the launch completed its hand-offs (or the call did not take the fused path), FA_ERR_LAUNCH = a dQ hand-off timed out: that 256-row block of dq was
   not written, the gradients of this call must be discarded.  Both binders call it after every fused backward (the mode is experimental; the default
   backward has no cross-workgroup hand-off and nothing to check). */
int fa_bwd_fused_status(const FaBwdParams* params, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FA_GFX950_H_ */
