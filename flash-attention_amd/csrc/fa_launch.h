// Host-side launch entry points of the kernel translation units (internal to libfa_gfx950.so).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include "fa_kernel_params.h"

namespace fa {

// Run-time knobs (environment, read ONCE per process; fa_knobs_reload() re-reads them -- tests and the A/B tools
// call it after changing the environment).  Defined in fa_api.cpp.
struct Knobs {
  int fwd_nw;          // FA_FWD_NW: 0 = heuristic; 4 / 8 / 16 lock-step, 34 / 38 pipelined, 64 = 64-rows-per-wave kernel
  float rescale_thr;   // FA_RESCALE_THR: deferred-rescale threshold in log2 units (default 8; 0 = the reference's rule)
  int varlen_list;     // FA_VARLEN_LIST: 0 = always the dense varlen grid
  int il_sched;        // FA_IL_SCHED: 0 = compiler-ordered pipelined step, else hand-placed slots
  int bwd_dq_nw;       // FA_BWD_DQ_NW: 0 = heuristic; recomputing dQ kernel of 4 or 8 waves x 32 rows, 64 = 4 waves x 64 rows (fa_bwd_w64.hip)
  int bwd_dkdv;        // FA_BWD_DKDV: 0 = heuristic; 8 = eight waves x 32 keys (fa_bwd.hip), 64 = four waves x 64 keys, software-pipelined (fa_bwd_dkdv_w64.hip)
  int bwd_mode;        // FA_BWD_MODE: 0 = the measured table (fa_api.cpp: the recomputing pair -- 7 contractions, no scratch -- except head dim 128 under a causal mask from 1k to 2k
                       // rows, where the fused launch below runs on <= 1 GiB of workspace); 1 / -1 = the recomputing pair everywhere; 3 = the dK/dV part hands dS over and dQ = dS.K
                       // (5 contractions) in ONE persistent launch (fa_bwd.hip fa_bwd_fused_kernel): plain attention, head dim 64 / 128, fixed-length batches, no
                       // left window, whose dS workspace fits FA_BWD_DS_CAP_MB.  (2 = round 2's two-launch dS spill: retired, superseded by 5)
                       // 5 = the 5-contraction backward in chunked mixed launches (fa_bwd_dkdv_w64.hip fa_bwd_c5_kernel; workspace bounded by FA_BWD_C5_CAP_MB);
                       // -1 = never (the recomputing pair everywhere); 0 = the measured table (fa_api.cpp bwd_c5_plan)
  int bwd_c5_cap_mb;   // FA_BWD_C5_CAP_MB: workspace bound of the 5-contraction backward, both slots together (default 1024)
  int bwd_gsplit;      // FA_BWD_GSPLIT: GQA group split of the dK/dV kernels (fa_api.cpp bwd_gsplit_plan): 1 (default) = when the grid does not fill the chip, 0 = never, n > 1 = always, up to n virtual heads per group (tests)
  int fz_line;         // FA_FZ_LINE: fused backward, int32 words between two arrival counters of the sync area (default 32 = one 128-byte line each)
  int bwd_ds_cap_mb;   // FA_BWD_DS_CAP_MB: largest dS workspace FA_BWD_MODE=3 asks for (default 8192)
  int bwd_fused_check; // FA_BWD_FUSED_CHECK=-1: never read the flag, not even for FA_BWD_MODE=3 (A/B timings of the launch itself); =1: fa_bwd_fused_status reads the fused launch's error flag (a stream sync) also where the fused backward ran by default
  int w64_persist;     // FA_W64_PERSIST: 0 = one workgroup per block (no persistent walk) in the 64-rows-per-wave forward
  int pack_gqa;        // FA_PACK_GQA: 0 = never pack the query heads of a KV group into the rows of a block on the KV-cache path (A/B, tests)
  int dkdv_prescale;   // FA_DKDV_PRESCALE=1: the plain dK/dV kernel pre-scales K by softmax_scale*log2e (rounded to the input dtype; ~3 % faster, fa_bwd.hip: PRE);
                       // default 0 = every score scaled in fp32 (FEAT_EXACT)
  int bwd_fuse_delta;  // FA_BWD_FUSE_DELTA=0: always run the delta pre-pass (default 1: the 64-rows-per-wave dQ kernel computes softmax_d of its rows itself and runs first)
  int strict;          // FA_STRICT=1: the reference's numerics contract -- rescale on any growth of a row maximum (threshold 0) and
                       // softmax_scale applied in fp32 to every score (never the bf16 pre-scaled Q of the 64-rows-per-wave kernel)
};
const Knobs& knobs();

// What the last fa_fwd* / fa_bwd* call of this thread launched (fa_last_schedule in the C ABI).
struct LastSchedule {
  int fwd_kernel;   // 0 none, 1 fa_fwd_kernel (lock-step), 2 fa_fwd_il_kernel (pipelined), 3 fa_fwd_w64_kernel
  int fwd_nw;       // waves per workgroup (16 = 8-wave ping-pong)
  int fwd_feat;     // FEAT_* variant of the lock-step kernel
  int fwd_splits;   // split-KV factor
  int fwd_list;     // 1 = varlen work list
  int d, bf16;
  int bwd_dq_nw, bwd_list, bwd_spill;
  int fwd_pack;     // query heads packed into the rows (FwdK::pack_g)
  int bwd_dkdv_nw;  // dK/dV schedule: 8 = eight waves x 32 keys (4 at head dim 256), 64 = four waves x 64 keys (fa_bwd_dkdv_w64.hip)
  char name[96];
};
LastSchedule& last_schedule();

// CUs of the current device (cached per device id)
inline int device_cu_count() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  int v = dev < 64 ? cache[dev].load(std::memory_order_relaxed) : 0;
  if (v == 0) {
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (dev < 64) cache[dev].store(v, std::memory_order_relaxed);
  }
  return v;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is per device: `mask` (one per kernel instantiation) records the devices
// it has been set on.  Concurrent first calls may both set it (idempotent).  Also checks that the kernel has no static
// LDS in front of the dynamic segment when the kernel addresses LDS by absolute byte offset.
inline int ensure_dyn_lds(std::atomic<unsigned long long>& mask, const void* kern, int smem, bool need_base0 = false) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  const unsigned long long bit = 1ull << (dev & 63);
  if (dev < 64 && (mask.load(std::memory_order_acquire) & bit)) return 0;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return -1;
  if (need_base0) {
    hipFuncAttributes fattr;
    if (hipFuncGetAttributes(&fattr, kern) != hipSuccess || fattr.sharedSizeBytes != 0) return -1;
  }
  if (dev < 64) mask.fetch_or(bit, std::memory_order_release);
  return 0;
}

// Choose the XCD mapping (see xcd_interleave): whole KV groups split by head range when the KV head count is a
// multiple of 8; otherwise round-robin over KV groups, falling back to heads and then to single blocks whenever
// the coarser unit count would leave XCDs idle (fewer than 16 units, or a ragged last round above 12 %).
inline void choose_units(int batch, int kv_heads, int heads_per_group, int blocks_per_head, int& n_units, int& unit_size, int& hpx) {
  const int n_groups = batch * kv_heads;
  hpx = 0;
  if (kv_heads % 8 == 0) { n_units = n_groups; unit_size = heads_per_group * blocks_per_head; hpx = kv_heads / 8; return; }
  auto ok = [](int n) { return n >= 16 && ((n + 7) / 8 * 8 - n) * 8 <= n; };
  if (ok(n_groups)) { n_units = n_groups; unit_size = heads_per_group * blocks_per_head; return; }
  const int n_heads = n_groups * heads_per_group;
  if (ok(n_heads)) { n_units = n_heads; unit_size = blocks_per_head; return; }
  n_units = n_heads * blocks_per_head; unit_size = 1;
}
inline long long units_grid(int n_units, int unit_size) { return (long long)((n_units + 7) / 8) * 8 * unit_size; }

// Forward.  `nw` = waves per workgroup (4 or 8); query block = 32*nw rows.  Returns 0, -1 (launch
// failure) or -2 (no kernel built for this dtype/head-dim/nw).
int launch_fwd(const FwdK& p, int dtype_bf16, int d, int nw, hipStream_t stream);
int fwd_block_m(int nw);

// KV-cache append: 16-byte copies of the new rows into the (contiguous, batch-indexed or paged) cache.
struct KvAppendK {
  const void* knew; const void* vnew; void* kcache; void* vcache;
  int64_t kn_bs, kn_rs, kn_hs, vn_bs, vn_rs, vn_hs, kc_bs, kc_rs, kc_hs, vc_bs, vc_rs, vc_hs;
  const int32_t* seqlens_k; const int32_t* kv_batch_idx; const int32_t* block_table; int64_t block_table_bs;
  int32_t page_size, b, s_new, h_k, d;
};
int launch_kv_append(const KvAppendK& p, hipStream_t stream);
// Software-pipelined forward (fa_fwd_il.hip); nw = 4 or 8 waves per workgroup.  No softcap / ALiBi variant.
// varlen work-list pre-pass: blocks of `blk` rows of side a (cu_a), work estimated against side o (cu_o)
struct SchedK {
  const int32_t* cu_a; const int32_t* cu_o; const int32_t* seqused_o;
  int2* list; int32_t nb, blk, bound, wl, wr, keys_blocked;
  int32_t work_shift;   // log2 of the work-bucket width in rows of side o: the longest sequence of side o spans the 1024 buckets
};
inline int sched_work_shift(int max_len_o) { int s = 6; while (((long)max_len_o >> s) > 1023) ++s; return s; }
int launch_varlen_schedule(const SchedK& p, hipStream_t stream);
struct RotaryK {
  const void* x; void* y; const void* cos; const void* sin; const int32_t* offsets;
  int64_t x_bs, x_rs, x_hs, y_bs, y_rs, y_hs, cos_rs;
  int32_t b, s, h, d, rotary_dim, seqlen_ro, interleaved, per_token;
};
int launch_rotary(const RotaryK& p, int dtype_bf16, hipStream_t stream);
int launch_splitkv_combine(const FwdK& p, int dtype_bf16, int d, hipStream_t stream);
int launch_set_rng(uint64_t seed, uint64_t offset, uint64_t* dst, hipStream_t stream);
int launch_fwd_il(const FwdK& p, int dtype_bf16, int d, int nw, hipStream_t stream);
// 64-rows-per-wave forward (fa_fwd_w64.hip): 4 waves, 256 query rows per workgroup, one workgroup per CU.
int launch_fwd_w64(const FwdK& p, int dtype_bf16, int d, hipStream_t stream);

// Backward: delta = rowsum(dO*O) pre-pass, dK/dV kernel (loops over query blocks),
// dQ kernel (loops over key blocks).  Same return convention.
int launch_bwd_delta(const BwdK& p, int dtype_bf16, int d, hipStream_t stream);
int launch_bwd_gsum(const void* src, void* dst, int dtype_bf16, int b, int sk, int h_k, int gs, int d, int64_t dst_bs, int64_t dst_rs, int64_t dst_hs, hipStream_t stream);   // fa_bwd.hip: sums a split GQA group's partial dK / dV
int launch_bwd_dkdv(const BwdK& p, int dtype_bf16, int d, hipStream_t stream);
int launch_bwd_fused(const BwdK& p, int dtype_bf16, int d, hipStream_t stream);   // fa_bwd.hip (FA_BWD_PART=3): dK/dV + dQ = dS.K in one launch; -2 = does not apply
int launch_bwd_dq(const BwdK& p, int dtype_bf16, int d, hipStream_t stream);
int launch_bwd_dq_w64(const BwdK& p, int dtype_bf16, int d, hipStream_t stream);
int launch_bwd_dkdv_w64(const BwdK& p, int dtype_bf16, int d, hipStream_t stream);
int launch_bwd_c5(const BwdK& p, int dtype_bf16, int d, hipStream_t stream);   // fa_bwd_dkdv_w64.hip (FA_DKDV64_PART=2): one mixed launch of the 5-contraction backward (dK/dV items of a chunk + dQ = dS.K items of the chunk before); -2 = not covered   // fa_bwd_dkdv_w64.hip: 64 keys per wave, software-pipelined; -2 = not covered
int bwd_block_m(int dq_nw);   // query rows per dQ workgroup of schedule dq_nw (BwdK::dq_nw)
int bwd_block_n(int d);   // key rows per dK/dV workgroup (256; 128 for head dim 256)

}  // namespace fa
