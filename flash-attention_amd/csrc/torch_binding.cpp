// PyTorch-ROCm extension module `flash_attn_2_cuda`: the backend module the reference's Python layer
// imports (flash_attn/flash_attn_interface.py:13-23) -- fwd / varlen_fwd / bwd / varlen_bwd /
// fwd_kvcache with the positional signatures of csrc/flash_attn/flash_api.cpp:368-382, :538-561,
// :800-820, :1010-1035, :1243-1264 (pybind block :1535-1542).
//
// This file is glue only: tensor validation (TORCH_CHECK messages follow the reference where its
// tests match on them), output allocation from the caching allocator, current-stream lookup, and one
// call into the C ABI of libfa_gfx950.so (include/fa_gfx950.h).  No kernels, no math.
#include <ATen/ATen.h>
#include <ATen/Context.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/utils/pybind.h>
#include <torch/extension.h>

#include <mutex>
#include <optional>
#include <vector>

#include "fa_gfx950.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<at::Tensor>;
// The retired `generator` slot of fwd / varlen_fwd / bwd / varlen_bwd (flash_api.cpp:382-386: kept for argument positions, must be None).  The CUDA
// extension types it std::optional<at::Tensor>, the ROCm one std::optional<at::Generator> (csrc/flash_attn_ck/mha_fwd.cpp): any Python object is
// taken here and everything but None is refused with the reference's message.
using GenSlot = py::object;

#define CHECK_DEVICE(x) TORCH_CHECK((x).is_cuda(), #x " must be on CUDA")
#define CHECK_LAST_CONTIG(x) TORCH_CHECK((x).stride(-1) == 1, #x " must have contiguous last dimension")
#define CHECK_SHAPE(x, ...) TORCH_CHECK((x).sizes() == at::IntArrayRef({__VA_ARGS__}), #x " must have shape (" #__VA_ARGS__ ")")

void fa_check(int rc) { TORCH_CHECK(rc == FA_OK, fa_last_error()); }

void* cur_stream(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.get_device()).stream(); }

int dtype_code(const Tensor& q) {
  TORCH_CHECK(q.dtype() == at::kHalf || q.dtype() == at::kBFloat16, "FlashAttention only support fp16 and bf16 data type");
  return q.dtype() == at::kBFloat16 ? FA_DTYPE_BF16 : FA_DTYPE_FP16;
}

// Head dim the tensors go to the kernels with: their own.  Every multiple of 8 up to 256 is taken as it is -- the six built sizes directly, the sizes
// in between through the kernels' run-time column bound (FaFwdParams / FaBwdParams::d, fa_api.cpp: head_dim_kernel): no padded copies anywhere.
void check_head_dim(int64_t d) { TORCH_CHECK(d <= 256, "FlashAttention only supports head dimension at most 256"); }

void common_checks(const Tensor& q, const Tensor& k, const Tensor& v, double p_dropout, const OptTensor& alibi,
                   const GenSlot& gen) {
  TORCH_CHECK(gen.is_none(), "flash-attn: the RNG `generator` argument is no longer supported and must be None; dropout (when enabled) uses the default generator of the device.");
  TORCH_CHECK(p_dropout >= 0.0 && p_dropout < 1.0, "p_dropout must be in [0, 1)");
  CHECK_DEVICE(q); CHECK_DEVICE(k); CHECK_DEVICE(v);
  TORCH_CHECK(k.dtype() == q.dtype() && v.dtype() == q.dtype(), "query, key and value must have the same dtype");
  TORCH_CHECK(q.stride(-1) == 1 && k.stride(-1) == 1 && v.stride(-1) == 1, "Input tensor must have contiguous last dimension");
  if (alibi.has_value()) {
    TORCH_CHECK(alibi->dtype() == at::kFloat, "ALiBi slopes must have dtype fp32");
    CHECK_DEVICE(*alibi);
    TORCH_CHECK(alibi->stride(-1) == 1, "ALiBi slopes tensor must have contiguous last dimension");
  }
}

// rng_state = {seed, offset} of the device's default generator (flash_api.cpp:496-515); the Philox offset advances as
// in csrc/flash_attn_ck/mha_fwd.cpp:283-295 so that consecutive calls draw different masks.
Tensor make_rng_state(const Tensor& q, double p_dropout, int64_t B, int64_t H) {
  Tensor rng_state = at::empty({2}, q.options().dtype(at::kLong));
  if (p_dropout > 0.0) {
    at::Generator gen = at::globalContext().defaultGenerator(q.device());
    uint64_t seed, off;
    {
      std::lock_guard<std::mutex> lock(gen.mutex());
      seed = gen.current_seed();
      off = gen.get_offset();
      gen.set_offset(off + (uint64_t)((B * H * 64 + 3) / 4) * 4);
    }
    fa_check(fa_set_rng_state(seed, off, reinterpret_cast<uint64_t*>(rng_state.data_ptr<int64_t>()), cur_stream(q)));
  }
  return rng_state;
}

const uint64_t* bwd_rng(double p_dropout, const OptTensor& rng_state) {
  if (!(p_dropout > 0.0)) return nullptr;
  TORCH_CHECK(rng_state.has_value(), "p_dropout > 0 in the backward needs the forward's rng_state");
  TORCH_CHECK(rng_state->is_cuda() && rng_state->dtype() == at::kLong && rng_state->numel() == 2, "rng_state must be a CUDA int64 tensor of shape (2,)");
  return reinterpret_cast<const uint64_t*>(rng_state->data_ptr<int64_t>());
}

void set_alibi(const OptTensor& alibi, int64_t B, int64_t H, const float*& ptr, int64_t& bs) {
  ptr = nullptr;
  bs = 0;
  if (!alibi.has_value()) return;
  TORCH_CHECK(alibi->sizes() == at::IntArrayRef({H}) || alibi->sizes() == at::IntArrayRef({B, H}),
              "ALiBi slopes must have shape (nheads,) or (batch_size, nheads)");
  ptr = alibi->data_ptr<float>();
  bs = alibi->dim() == 2 ? alibi->stride(0) : 0;
}

std::vector<Tensor> mha_fwd(Tensor& q, const Tensor& k, const Tensor& v, OptTensor& out_, OptTensor& alibi_slopes_,
                            const double p_dropout, const double softmax_scale, bool is_causal, int64_t window_size_left,
                            int64_t window_size_right, const double softcap, const bool return_softmax,
                            GenSlot gen_) {
  common_checks(q, k, v, p_dropout, alibi_slopes_, gen_);
  TORCH_CHECK(!return_softmax || p_dropout > 0.0, "return_softmax is only supported when p_dropout > 0.0");
  TORCH_CHECK(q.dim() == 4 && k.dim() == 4 && v.dim() == 4, "q, k, v must be 4-D (batch, seqlen, nheads, headdim)");
  const int64_t B = q.size(0), Sq = q.size(1), H = q.size(2), D = q.size(3), Sk = k.size(1), Hk = k.size(2);
  TORCH_CHECK(B > 0, "batch size must be positive");
  TORCH_CHECK(D <= 256, "FlashAttention forward only supports head dimension at most 256");
  TORCH_CHECK(D % 8 == 0, "query, key, value, and out_ must have a head_size that is a multiple of 8");
  TORCH_CHECK(H % Hk == 0, "Number of heads in key/value must divide number of heads in query");
  CHECK_SHAPE(k, B, Sk, Hk, D);
  CHECK_SHAPE(v, B, Sk, Hk, D);
  c10::DeviceGuard guard(q.device());
  if (out_.has_value()) {
    TORCH_CHECK(out_->dtype() == q.dtype(), "Output must have the same dtype as inputs");
    CHECK_DEVICE(*out_); CHECK_LAST_CONTIG(*out_); CHECK_SHAPE(*out_, B, Sq, H, D);
  }
  // One query row and grouped heads: the query heads of a KV group become the rows of one block, K/V are streamed once
  // per KV head (seqlenq_ngroups_swapped, flash_api.cpp:429-437 and :531-535)
  if (Sq == 1 && H > Hk && window_size_left < 0 && window_size_right < 0 && p_dropout == 0.0 && !alibi_slopes_.has_value() && Sk > 0) {
    const int64_t ng = H / Hk;
    Tensor q2 = q.reshape({B, Hk, ng, D}).transpose(1, 2);
    OptTensor none;
    std::vector<Tensor> r = mha_fwd(q2, k, v, none, none, 0.0, softmax_scale, false, -1, -1, softcap, false, GenSlot(py::none()));
    Tensor o = r[0].transpose(1, 2).reshape({B, 1, H, D});
    if (out_.has_value()) { out_->copy_(o); o = *out_; }
    return {o, r[1].reshape({B, H, 1}), r[2], r[3]};
  }
  check_head_dim(D);
  const Tensor &qp = q, &kp = k, &vp = v;
  Tensor out;
  out = out_.has_value() ? *out_ : at::empty({B, Sq, H, D}, q.options());
  Tensor lse = at::empty({B, H, Sq}, q.options().dtype(at::kFloat));
  Tensor rng_state = make_rng_state(q, p_dropout, B, H);
  // return_softmax: the random byte of every (query, key) pair, the ROCm backend's payload (csrc/flash_attn_ck/mha_fwd.cpp:275-279)
  Tensor p = return_softmax ? at::zeros({B, H, Sq, Sk}, q.options().dtype(at::kByte)) : at::empty({0}, q.options());
  if (Sk == 0) {  // flash_api.cpp:524-528
    out.zero_();
    lse.fill_(std::numeric_limits<float>::infinity());
  } else if (Sq > 0) {
    FaFwdParams a{};
    a.q = qp.data_ptr(); a.k = kp.data_ptr(); a.v = vp.data_ptr(); a.o = out.data_ptr(); a.softmax_lse = lse.data_ptr<float>();
    a.q_batch_stride = qp.stride(0); a.q_row_stride = qp.stride(1); a.q_head_stride = qp.stride(2);
    a.k_batch_stride = kp.stride(0); a.k_row_stride = kp.stride(1); a.k_head_stride = kp.stride(2);
    a.v_batch_stride = vp.stride(0); a.v_row_stride = vp.stride(1); a.v_head_stride = vp.stride(2);
    a.o_batch_stride = out.stride(0); a.o_row_stride = out.stride(1); a.o_head_stride = out.stride(2);
    set_alibi(alibi_slopes_, B, H, a.alibi_slopes, a.alibi_batch_stride);
    a.b = B; a.h = H; a.h_k = Hk; a.d = (int)D; a.seqlen_q = Sq; a.seqlen_k = Sk; a.total_q = B * Sq;
    a.dtype = dtype_code(q);
    a.is_causal = is_causal; a.window_left = (int)window_size_left; a.window_right = (int)window_size_right;
    a.softmax_scale = (float)softmax_scale; a.softcap = (float)softcap;
    if (p_dropout > 0.0) {
      a.p_dropout = (float)p_dropout;
      a.rng_state = reinterpret_cast<const uint64_t*>(rng_state.data_ptr<int64_t>());
      if (return_softmax) {
        a.randval = p.data_ptr<uint8_t>();
        a.randval_batch_stride = p.stride(0); a.randval_head_stride = p.stride(1); a.randval_row_stride = p.stride(2);
      }
    }
    fa_check(fa_fwd(&a, cur_stream(q)));
  }
  return {out, lse, p, rng_state};
}

std::vector<Tensor> mha_varlen_fwd(Tensor& q, const Tensor& k, const Tensor& v, OptTensor& out_, const Tensor& cu_seqlens_q,
                                   const Tensor& cu_seqlens_k, OptTensor& seqused_k, OptTensor& leftpad_k_,
                                   OptTensor& block_table_, OptTensor& alibi_slopes_, int64_t max_seqlen_q,
                                   const int64_t max_seqlen_k, const double p_dropout, const double softmax_scale,
                                   const bool zero_tensors, bool is_causal, int64_t window_size_left, int64_t window_size_right,
                                   const double softcap, const bool return_softmax, GenSlot gen_,
                                   const int64_t num_splits) {
  common_checks(q, k, v, p_dropout, alibi_slopes_, gen_);
  TORCH_CHECK(!return_softmax || p_dropout > 0.0, "return_softmax is only supported when p_dropout > 0.0");
  const bool paged = block_table_.has_value();  // k / v are (num_blocks, page, Hk, D), addressed through block_table (flash_api.cpp:586-649)
  if (paged) {
    CHECK_DEVICE(*block_table_);
    TORCH_CHECK(block_table_->dtype() == at::kInt, "block_table must have dtype torch.int32");
    TORCH_CHECK(block_table_->stride(-1) == 1, "block_table must have contiguous last dimension");
    TORCH_CHECK(k.dim() == 4 && v.dim() == 4, "With block_table, k and v must be 4-D (num_blocks, page_block_size, nheads_k, headdim)");
    TORCH_CHECK(k.size(1) % 256 == 0, "Paged KV cache block size must be divisible by 256");
    TORCH_CHECK(!leftpad_k_.has_value(), "We don't support Paged KV and leftpad_k running at the same time yet");
  }
  if (leftpad_k_.has_value()) {
    TORCH_CHECK(leftpad_k_->dtype() == at::kInt, "leftpad_k must have dtype int32");
    CHECK_DEVICE(*leftpad_k_); TORCH_CHECK(leftpad_k_->is_contiguous(), "leftpad_k must be contiguous");
  }
  TORCH_CHECK(num_splits <= 1, "num_splits > 1 is not supported");
  TORCH_CHECK(cu_seqlens_q.dtype() == at::kInt, "cu_seqlens_q must have dtype int32");
  TORCH_CHECK(cu_seqlens_k.dtype() == at::kInt, "cu_seqlens_k must have dtype int32");
  CHECK_DEVICE(cu_seqlens_q); CHECK_DEVICE(cu_seqlens_k);
  TORCH_CHECK(cu_seqlens_q.is_contiguous() && cu_seqlens_k.is_contiguous(), "cu_seqlens_q/k must be contiguous");
  TORCH_CHECK(q.dim() == 3 && (paged || (k.dim() == 3 && v.dim() == 3)), "q, k, v must be 3-D (total, nheads, headdim)");
  const int64_t total_q = q.size(0), H = q.size(1), D = q.size(2);
  const int64_t total_k = paged ? k.size(0) * k.size(1) : k.size(0), Hk = paged ? k.size(2) : k.size(1);
  const int64_t B = cu_seqlens_q.numel() - 1;
  TORCH_CHECK(B > 0, "batch size must be positive");
  CHECK_SHAPE(cu_seqlens_q, B + 1);
  CHECK_SHAPE(cu_seqlens_k, B + 1);
  TORCH_CHECK(D <= 256 && D % 8 == 0, "head_size must be a multiple of 8 and at most 256");
  TORCH_CHECK(H % Hk == 0, "Number of heads in key/value must divide number of heads in query");
  if (paged) {
    CHECK_SHAPE(k, k.size(0), k.size(1), Hk, D);
    TORCH_CHECK(v.sizes() == k.sizes(), "paged k / v shape mismatch");
    CHECK_SHAPE(*block_table_, B, block_table_->size(1));
  } else {
    CHECK_SHAPE(k, total_k, Hk, D);
    CHECK_SHAPE(v, total_k, Hk, D);
  }
  if (leftpad_k_.has_value()) CHECK_SHAPE(*leftpad_k_, B);
  if (seqused_k.has_value()) {
    TORCH_CHECK(seqused_k->dtype() == at::kInt, "seqused_k must have dtype int32");
    CHECK_DEVICE(*seqused_k);
    TORCH_CHECK(seqused_k->is_contiguous(), "seqused_k must be contiguous");
    CHECK_SHAPE(*seqused_k, B);
  }
  c10::DeviceGuard guard(q.device());
  // One query row per sequence and grouped heads (decode over a packed batch): the query heads of a KV group become the
  // rows of one block (seqlenq_ngroups_swapped, flash_api.cpp:620-629 and :776-782); q is then ngroups rows per sequence
  if (max_seqlen_q == 1 && total_q == B && H > Hk && window_size_left < 0 && window_size_right < 0 && p_dropout == 0.0 &&
      !alibi_slopes_.has_value() && max_seqlen_k > 0 && total_k > 0) {
    const int64_t ng = H / Hk;
    Tensor q2 = q.reshape({B, Hk, ng, D}).transpose(1, 2).reshape({B * ng, Hk, D});
    Tensor cu_q2 = at::arange(0, (B + 1) * ng, ng, cu_seqlens_q.options());
    OptTensor none;
    std::vector<Tensor> r = mha_varlen_fwd(q2, k, v, none, cu_q2, cu_seqlens_k, seqused_k, leftpad_k_, block_table_, none, ng, max_seqlen_k, 0.0,
                                           softmax_scale, zero_tensors, false, -1, -1, softcap, false, GenSlot(py::none()), num_splits);
    Tensor o = r[0].reshape({B, ng, Hk, D}).transpose(1, 2).reshape({B, H, D});
    if (out_.has_value()) {
      TORCH_CHECK(out_->dtype() == q.dtype(), "Output must have the same dtype as inputs");
      CHECK_DEVICE(*out_); CHECK_LAST_CONTIG(*out_); CHECK_SHAPE(*out_, total_q, H, D);
      out_->copy_(o); o = *out_;
    }
    // lse (Hk, B*ng) -> (H, B): head hk*ng + g of sequence b sits at [hk][b*ng + g]
    return {o, r[1].reshape({Hk, B, ng}).permute({0, 2, 1}).reshape({H, B}).contiguous(), r[2], r[3]};   // (.contiguous(): with ONE KV head the reshape is a strided view)
  }
  check_head_dim(D);
  const Tensor &qp = q, &kp = k, &vp = v;
  if (out_.has_value()) {
    TORCH_CHECK(out_->dtype() == q.dtype(), "Output must have the same dtype as inputs");
    CHECK_DEVICE(*out_); CHECK_LAST_CONTIG(*out_); CHECK_SHAPE(*out_, total_q, H, D);
  }
  Tensor out = out_.has_value() ? *out_ : at::empty({total_q, H, D}, q.options());
  Tensor lse = at::empty({H, total_q}, q.options().dtype(at::kFloat));
  Tensor rng_state = make_rng_state(q, p_dropout, B, H);
  // varlen payload layout of the ROCm backend: (nheads, total_q, max_seqlen_k)
  Tensor p = return_softmax ? at::zeros({H, total_q, max_seqlen_k}, q.options().dtype(at::kByte)) : at::empty({0}, q.options());
  if (zero_tensors) {  // flash_api.cpp:693-697
    out.zero_();
    lse.fill_(-std::numeric_limits<float>::infinity());
  }
  if (max_seqlen_k == 0 || total_k == 0) {
    out.zero_();
    lse.fill_(std::numeric_limits<float>::infinity());
  } else if (total_q > 0 && max_seqlen_q > 0) {
    FaFwdParams a{};
    a.q = qp.data_ptr(); a.k = kp.data_ptr(); a.v = vp.data_ptr(); a.o = out.data_ptr(); a.softmax_lse = lse.data_ptr<float>();
    a.q_row_stride = qp.stride(0); a.q_head_stride = qp.stride(1);
    if (paged) {
      a.k_batch_stride = kp.stride(0); a.k_row_stride = kp.stride(1); a.k_head_stride = kp.stride(2);
      a.v_batch_stride = vp.stride(0); a.v_row_stride = vp.stride(1); a.v_head_stride = vp.stride(2);
      a.block_table = block_table_->data_ptr<int>(); a.block_table_batch_stride = block_table_->stride(0);
      a.page_block_size = (int)kp.size(1);
    } else {
      a.k_row_stride = kp.stride(0); a.k_head_stride = kp.stride(1);
      a.v_row_stride = vp.stride(0); a.v_head_stride = vp.stride(1);
    }
    a.leftpad_k = leftpad_k_.has_value() ? leftpad_k_->data_ptr<int>() : nullptr;
    a.o_row_stride = out.stride(0); a.o_head_stride = out.stride(1);
    a.cu_seqlens_q = cu_seqlens_q.data_ptr<int>(); a.cu_seqlens_k = cu_seqlens_k.data_ptr<int>();
    a.seqused_k = seqused_k.has_value() ? seqused_k->data_ptr<int>() : nullptr;
    set_alibi(alibi_slopes_, B, H, a.alibi_slopes, a.alibi_batch_stride);
    a.b = B; a.h = H; a.h_k = Hk; a.d = (int)D; a.seqlen_q = (int)max_seqlen_q; a.seqlen_k = (int)max_seqlen_k; a.total_q = total_q;
    a.dtype = dtype_code(q);
    a.is_causal = is_causal; a.window_left = (int)window_size_left; a.window_right = (int)window_size_right;
    a.softmax_scale = (float)softmax_scale; a.softcap = (float)softcap;
    if (p_dropout > 0.0) {
      a.p_dropout = (float)p_dropout;
      a.rng_state = reinterpret_cast<const uint64_t*>(rng_state.data_ptr<int64_t>());
      if (return_softmax) {
        a.randval = p.data_ptr<uint8_t>();
        a.randval_batch_stride = 0; a.randval_head_stride = p.stride(0); a.randval_row_stride = p.stride(1);
      }
    }
    Tensor ws;  // work list of an uneven packed batch (0 bytes = dense grid)
    const int64_t ws_bytes = fa_fwd_workspace_bytes(&a);
    if (ws_bytes > 0) {
      ws = at::empty({ws_bytes}, q.options().dtype(at::kByte));
      a.workspace = ws.data_ptr(); a.workspace_bytes = ws_bytes;
    }
    fa_check(fa_varlen_fwd(&a, cur_stream(q)));
  }
  return {out, lse, p, rng_state};
}

Tensor grad_buffer(const OptTensor& g_, const Tensor& like, const char* name) {
  if (!g_.has_value()) return at::empty_like(like);
  TORCH_CHECK(g_->dtype() == like.dtype(), name, " must have the same dtype as q");
  TORCH_CHECK(g_->is_cuda(), name, " must be on CUDA");
  TORCH_CHECK(g_->stride(-1) == 1, name, " must have contiguous last dimension");
  TORCH_CHECK(g_->sizes() == like.sizes(), name, " has the wrong shape");
  return *g_;
}

struct BwdBufs {
  Tensor dout, q, k, v, out, dq, dk, dv;  // tensors handed to the kernels (padded copies when D is not native)
};

void run_bwd(FaBwdParams& a, const Tensor& q, bool varlen) {
  const int64_t ws = fa_bwd_workspace_bytes(&a);
  Tensor wsbuf;
  if (ws > 0) {
    // A fixed-length call's workspace is the dS area of the 5-contraction launches (up to 1 GiB by default): a speed-up, not a requirement.  When the allocator
    // cannot supply it the call proceeds without one and the library runs the recomputing pair (fa_api.cpp do_bwd checks workspace_bytes).
    try {
      wsbuf = at::empty({ws}, q.options().dtype(at::kByte));
    } catch (const c10::OutOfMemoryError&) {
      if (varlen) throw;   // (work lists: a few KB -- if that fails nothing will succeed)
    }
    if (wsbuf.defined()) {
      a.workspace = wsbuf.data_ptr();
      a.workspace_bytes = ws;
    }
  }
  fa_check(varlen ? fa_varlen_bwd(&a, cur_stream(q)) : fa_bwd(&a, cur_stream(q)));
  if (wsbuf.defined() && !varlen) fa_check(fa_bwd_fused_status(&a, cur_stream(q)));   // (FA_OK at once unless the opt-in fused backward ran: fa_gfx950.h)
}

void fill_bwd_ptrs(FaBwdParams& a, const BwdBufs& t, const Tensor& lse, Tensor& delta) {
  a.dout = t.dout.data_ptr(); a.q = t.q.data_ptr(); a.k = t.k.data_ptr(); a.v = t.v.data_ptr(); a.o = t.out.data_ptr();
  a.softmax_lse = lse.data_ptr<float>();
  a.dq = t.dq.data_ptr(); a.dk = t.dk.data_ptr(); a.dv = t.dv.data_ptr(); a.softmax_d = delta.data_ptr<float>();
}

#define SET3(nm, T_) a.nm##_batch_stride = (T_).stride(0); a.nm##_row_stride = (T_).stride(1); a.nm##_head_stride = (T_).stride(2);
#define SET2(nm, T_) a.nm##_row_stride = (T_).stride(0); a.nm##_head_stride = (T_).stride(1);

std::vector<Tensor> mha_bwd(const Tensor& dout, const Tensor& q, const Tensor& k, const Tensor& v, const Tensor& out,
                            const Tensor& softmax_lse, OptTensor& dq_, OptTensor& dk_, OptTensor& dv_, OptTensor& alibi_slopes_,
                            const double p_dropout, const double softmax_scale, const bool is_causal, int64_t window_size_left,
                            int64_t window_size_right, const double softcap, const bool deterministic,
                            GenSlot gen_, OptTensor& rng_state) {
  common_checks(q, k, v, p_dropout, alibi_slopes_, gen_);
  CHECK_DEVICE(dout); CHECK_DEVICE(out); CHECK_DEVICE(softmax_lse);
  TORCH_CHECK(dout.dtype() == q.dtype() && out.dtype() == q.dtype(), "query and dout/out must have the same dtype");
  TORCH_CHECK(out.stride(-1) == 1, "out tensor must have contiguous last dimension");
  TORCH_CHECK(dout.stride(-1) == 1, "dout tensor must have contiguous last dimension");
  TORCH_CHECK(softmax_lse.dtype() == at::kFloat && softmax_lse.is_contiguous(), "softmax_lse must be contiguous fp32");
  const int64_t B = q.size(0), Sq = q.size(1), H = q.size(2), D = q.size(3), Sk = k.size(1), Hk = k.size(2);
  TORCH_CHECK(B > 0, "batch size must be positive");
  TORCH_CHECK(D % 8 == 0, "head_size should be a multiple of 8");
  TORCH_CHECK(D <= 256, "FlashAttention backward only supports head dimension at most 256");
  TORCH_CHECK(H % Hk == 0, "Number of heads in key/value must divide number of heads in query");
  CHECK_SHAPE(k, B, Sk, Hk, D); CHECK_SHAPE(v, B, Sk, Hk, D); CHECK_SHAPE(out, B, Sq, H, D); CHECK_SHAPE(dout, B, Sq, H, D);
  CHECK_SHAPE(softmax_lse, B, H, Sq);
  c10::DeviceGuard guard(q.device());
  Tensor dq = grad_buffer(dq_, q, "dq"), dk = grad_buffer(dk_, k, "dk"), dv = grad_buffer(dv_, v, "dv");
  Tensor delta = at::empty({B, H, Sq}, q.options().dtype(at::kFloat));  // ROCm shape convention (mha_bwd.cpp:340)
  if (Sq == 0 || Sk == 0) {  // flash_api.cpp:992-999
    dq.zero_(); dk.zero_(); dv.zero_(); delta.zero_();
    return {dq, dk, dv, delta};
  }
  check_head_dim(D);
  BwdBufs t{dout, q, k, v, out, dq, dk, dv};
  FaBwdParams a{};
  fill_bwd_ptrs(a, t, softmax_lse, delta);
  SET3(do, t.dout) SET3(q, t.q) SET3(k, t.k) SET3(v, t.v) SET3(o, t.out) SET3(dq, t.dq) SET3(dk, t.dk) SET3(dv, t.dv)
  set_alibi(alibi_slopes_, B, H, a.alibi_slopes, a.alibi_batch_stride);
  a.b = B; a.h = H; a.h_k = Hk; a.d = (int)D; a.seqlen_q = Sq; a.seqlen_k = Sk; a.total_q = B * Sq; a.total_k = B * Sk;
  a.dtype = dtype_code(q);
  a.is_causal = is_causal; a.window_left = (int)window_size_left; a.window_right = (int)window_size_right;
  a.softmax_scale = (float)softmax_scale; a.softcap = (float)softcap; a.deterministic = deterministic;
  a.p_dropout = (float)p_dropout; a.rng_state = bwd_rng(p_dropout, rng_state);
  run_bwd(a, q, false);
  return {dq, dk, dv, delta};
}

std::vector<Tensor> mha_varlen_bwd(const Tensor& dout, const Tensor& q, const Tensor& k, const Tensor& v, const Tensor& out,
                                   const Tensor& softmax_lse, OptTensor& dq_, OptTensor& dk_, OptTensor& dv_,
                                   const Tensor& cu_seqlens_q, const Tensor& cu_seqlens_k, OptTensor& alibi_slopes_,
                                   const int64_t max_seqlen_q, const int64_t max_seqlen_k, const double p_dropout,
                                   const double softmax_scale, const bool zero_tensors, const bool is_causal,
                                   int64_t window_size_left, int64_t window_size_right, const double softcap,
                                   const bool deterministic, GenSlot gen_, OptTensor& rng_state) {
  common_checks(q, k, v, p_dropout, alibi_slopes_, gen_);
  CHECK_DEVICE(dout); CHECK_DEVICE(out); CHECK_DEVICE(softmax_lse); CHECK_DEVICE(cu_seqlens_q); CHECK_DEVICE(cu_seqlens_k);
  TORCH_CHECK(dout.dtype() == q.dtype() && out.dtype() == q.dtype(), "query and dout/out must have the same dtype");
  TORCH_CHECK(cu_seqlens_q.dtype() == at::kInt && cu_seqlens_k.dtype() == at::kInt, "cu_seqlens_q/k must have dtype int32");
  TORCH_CHECK(cu_seqlens_q.is_contiguous() && cu_seqlens_k.is_contiguous(), "cu_seqlens_q/k must be contiguous");
  TORCH_CHECK(out.stride(-1) == 1 && dout.stride(-1) == 1, "out/dout tensor must have contiguous last dimension");
  TORCH_CHECK(softmax_lse.dtype() == at::kFloat && softmax_lse.is_contiguous(), "softmax_lse must be contiguous fp32");
  const int64_t total_q = q.size(0), H = q.size(1), D = q.size(2), total_k = k.size(0), Hk = k.size(1);
  const int64_t B = cu_seqlens_q.numel() - 1;
  TORCH_CHECK(B > 0, "batch size must be positive");
  TORCH_CHECK(D % 8 == 0 && D <= 256, "head_size should be a multiple of 8 and at most 256");
  TORCH_CHECK(H % Hk == 0, "Number of heads in key/value must divide number of heads in query");
  CHECK_SHAPE(k, total_k, Hk, D); CHECK_SHAPE(v, total_k, Hk, D); CHECK_SHAPE(out, total_q, H, D); CHECK_SHAPE(dout, total_q, H, D);
  CHECK_SHAPE(cu_seqlens_q, B + 1); CHECK_SHAPE(cu_seqlens_k, B + 1);
  CHECK_SHAPE(softmax_lse, H, total_q);
  c10::DeviceGuard guard(q.device());
  Tensor dq = grad_buffer(dq_, q, "dq"), dk = grad_buffer(dk_, k, "dk"), dv = grad_buffer(dv_, v, "dv");
  Tensor delta = at::empty({H, total_q}, q.options().dtype(at::kFloat));
  if (zero_tensors) { dq.zero_(); dk.zero_(); dv.zero_(); delta.zero_(); }  // flash_api.cpp:1171-1176
  if (max_seqlen_q == 0 || total_q == 0 || total_k == 0) {
    dq.zero_(); dk.zero_(); dv.zero_(); delta.zero_();
    return {dq, dk, dv, delta};
  }
  check_head_dim(D);
  BwdBufs t{dout, q, k, v, out, dq, dk, dv};
  FaBwdParams a{};
  fill_bwd_ptrs(a, t, softmax_lse, delta);
  SET2(do, t.dout) SET2(q, t.q) SET2(k, t.k) SET2(v, t.v) SET2(o, t.out) SET2(dq, t.dq) SET2(dk, t.dk) SET2(dv, t.dv)
  a.cu_seqlens_q = cu_seqlens_q.data_ptr<int>(); a.cu_seqlens_k = cu_seqlens_k.data_ptr<int>();
  set_alibi(alibi_slopes_, B, H, a.alibi_slopes, a.alibi_batch_stride);
  a.b = B; a.h = H; a.h_k = Hk; a.d = (int)D; a.seqlen_q = (int)max_seqlen_q; a.seqlen_k = (int)max_seqlen_k;
  a.total_q = total_q; a.total_k = total_k;
  a.dtype = dtype_code(q);
  a.is_causal = is_causal; a.window_left = (int)window_size_left; a.window_right = (int)window_size_right;
  a.softmax_scale = (float)softmax_scale; a.softcap = (float)softcap; a.deterministic = deterministic;
  a.p_dropout = (float)p_dropout; a.rng_state = bwd_rng(p_dropout, rng_state);
  run_bwd(a, q, true);
  return {dq, dk, dv, delta};
}

// mha_fwd_kvcache (flash_api.cpp:1243-1532): inference forward against a (contiguous, batch-indexed or paged) KV
// cache, optionally appending new keys/values first (rotary embedding of the new keys / queries and leftpad_k included).
std::vector<Tensor> mha_fwd_kvcache(Tensor& q, const Tensor& kcache, const Tensor& vcache, OptTensor& k_, OptTensor& v_,
                                    OptTensor& seqlens_k_, OptTensor& rotary_cos_, OptTensor& rotary_sin_,
                                    OptTensor& cache_batch_idx_, OptTensor& leftpad_k_, OptTensor& block_table_,
                                    OptTensor& alibi_slopes_, OptTensor& out_, const double softmax_scale, bool is_causal,
                                    int64_t window_size_left, int64_t window_size_right, const double softcap,
                                    bool is_rotary_interleaved, int64_t num_splits) {
  TORCH_CHECK(q.dtype() == at::kHalf || q.dtype() == at::kBFloat16, "FlashAttention only support fp16 and bf16 data type");
  TORCH_CHECK(kcache.dtype() == q.dtype(), "query and key must have the same dtype");
  TORCH_CHECK(vcache.dtype() == q.dtype(), "query and value must have the same dtype");
  CHECK_DEVICE(q); CHECK_DEVICE(kcache); CHECK_DEVICE(vcache);
  TORCH_CHECK(q.stride(-1) == 1 && kcache.stride(-1) == 1 && vcache.stride(-1) == 1, "Input tensor must have contiguous last dimension");
  const bool paged = block_table_.has_value();
  if (leftpad_k_.has_value()) {  // flash_api.cpp:1377-1386
    TORCH_CHECK(!paged, "We don't support Paged KV and leftpad_k running at the same time yet");
    TORCH_CHECK(leftpad_k_->dtype() == at::kInt, "leftpad_k must have dtype int32");
    CHECK_DEVICE(*leftpad_k_); TORCH_CHECK(leftpad_k_->is_contiguous(), "leftpad_k must be contiguous");
    CHECK_SHAPE(*leftpad_k_, q.size(0));
    TORCH_CHECK(seqlens_k_.has_value(), "leftpad_k needs seqlens_k (cache_seqlens)");
  }
  if (rotary_cos_.has_value()) {  // flash_api.cpp:1455-1477
    TORCH_CHECK(k_.has_value(), "If rotary cos/sin are provided, new key / value to be appended to KV cache must also be provided");
    TORCH_CHECK(rotary_sin_.has_value(), "If rotary cos is provided, rotary sin must also be provided");
    const Tensor &rc = *rotary_cos_, &rs = *rotary_sin_;
    CHECK_DEVICE(rc); CHECK_DEVICE(rs);
    TORCH_CHECK(rc.dim() == 2 && rs.sizes() == rc.sizes(), "rotary_cos and rotary_sin must both be (seqlen_ro, rotary_dim / 2)");
    TORCH_CHECK(2 * rc.size(1) <= q.size(3), "rotary_dim must be <= headdim");
    TORCH_CHECK((2 * rc.size(1)) % 16 == 0, "Only rotary dimensions divisible by 16 are currently supported");
    TORCH_CHECK(rc.dtype() == q.dtype() && rs.dtype() == q.dtype(), "rotary_cos/sin must have the same dtype as query");
    TORCH_CHECK(rc.stride(-1) == 1 && rs.stride(-1) == 1 && rc.stride(0) == rs.stride(0), "rotary_cos/sin must have contiguous last dimension and equal row strides");
  }
  if (paged) {
    TORCH_CHECK(!cache_batch_idx_.has_value(), "Paged KVcache does not support cache_batch_idx");
    CHECK_DEVICE(*block_table_);
    TORCH_CHECK(block_table_->dtype() == at::kInt, "block_table must have dtype torch.int32");
    TORCH_CHECK(block_table_->stride(-1) == 1, "block_table must have contiguous last dimension");
  }
  const int64_t B = q.size(0), Sq = q.size(1), H = q.size(2), D = q.size(3);
  const int64_t Hk = kcache.size(2);
  const int64_t page = paged ? kcache.size(1) : 0;
  const int64_t Sk = paged ? block_table_->size(1) * page : kcache.size(1);
  TORCH_CHECK(B > 0, "batch size must be positive");
  TORCH_CHECK(D <= 256, "FlashAttention forward only supports head dimension at most 256");
  if (D % 8 != 0) {
    // flash_api.cpp:1340-1350, 1517-1527: q and BOTH caches are zero-padded to the next multiple of 8 (copies of the whole cache -- "we don't expect
    // to get this case in practice", the reference says of it), the call runs on the copies, and appended keys / values are copied back.
    // (Head dims that ARE multiples of 8 but have no kernel of their own never copy: they run behind the kernels' run-time column bound.)
    const int64_t pad = 8 - D % 8;
    auto padded = [pad](const Tensor& t) { return at::constant_pad_nd(t, {0, pad}, 0); };
    Tensor q_p = padded(q), kc_p = padded(kcache), vc_p = padded(vcache);
    OptTensor k_p = k_.has_value() ? OptTensor(padded(*k_)) : OptTensor();
    OptTensor v_p = v_.has_value() ? OptTensor(padded(*v_)) : OptTensor();
    OptTensor no_out;
    std::vector<Tensor> r = mha_fwd_kvcache(q_p, kc_p, vc_p, k_p, v_p, seqlens_k_, rotary_cos_, rotary_sin_, cache_batch_idx_, leftpad_k_, block_table_,
                                            alibi_slopes_, no_out, softmax_scale, is_causal, window_size_left, window_size_right, softcap,
                                            is_rotary_interleaved, num_splits);
    Tensor out = r[0].slice(-1, 0, D);
    if (out_.has_value()) { out_->copy_(out); out = *out_; }
    if (k_.has_value()) {
      const_cast<Tensor&>(kcache).copy_(kc_p.slice(-1, 0, D));
      const_cast<Tensor&>(vcache).copy_(vc_p.slice(-1, 0, D));
    }
    return {out, r[1]};
  }
  TORCH_CHECK(H % Hk == 0, "Number of heads in key/value must divide number of heads in query");
  TORCH_CHECK(kcache.size(3) == D && vcache.sizes() == kcache.sizes(), "kcache / vcache shape mismatch");
  if (paged) {
    TORCH_CHECK(page % 256 == 0, "Paged KV cache block size must be divisible by 256");
  } else if (!cache_batch_idx_.has_value()) {
    TORCH_CHECK(kcache.size(0) == B, "kcache batch size must match q (or pass cache_batch_idx)");
  }
  if (seqlens_k_.has_value()) {
    TORCH_CHECK(seqlens_k_->dtype() == at::kInt, "seqlens_k must have dtype int32");
    CHECK_DEVICE(*seqlens_k_); TORCH_CHECK(seqlens_k_->is_contiguous(), "seqlens_k must be contiguous"); CHECK_SHAPE(*seqlens_k_, B);
  }
  if (cache_batch_idx_.has_value()) {
    TORCH_CHECK(cache_batch_idx_->dtype() == at::kInt, "cache_batch_idx must have dtype int32");
    CHECK_DEVICE(*cache_batch_idx_); TORCH_CHECK(cache_batch_idx_->is_contiguous(), "cache_batch_idx must be contiguous"); CHECK_SHAPE(*cache_batch_idx_, B);
  }
  c10::DeviceGuard guard(q.device());
  if (Sq == 1 && !alibi_slopes_.has_value()) is_causal = false;  // flash_api.cpp:1340
  if (is_causal) window_size_right = 0;

  int64_t s_new = k_.has_value() ? k_->size(1) : 0;
  TORCH_CHECK(s_new <= Sk, "If key is supplied, it must have seqlen <= the seqlen of the KV cache");                                  // flash_api.cpp:1397
  TORCH_CHECK(!rotary_cos_.has_value() || rotary_cos_->size(0) >= Sk, "cos/sin seqlen must be at least the seqlen of KV cache");   // :1470
  if (paged && seqlens_k_.has_value()) {  // the reference's guard (flash_api.cpp:1433-1447); costs a device->host sync
    const int64_t need = seqlens_k_->max().item<int>() + s_new;
    TORCH_CHECK(need <= Sk, "Paged KV cache: max(seqlens_k)", s_new > 0 ? " + seqlen_knew" : "", " (= ", need,
                ") exceeds the capacity addressable by block_table (max_num_blocks_per_seq * page_block_size = ", Sk, ")");
  }
  // rotary: new keys at positions cache_seqlens + i; queries too if causal / local, else all at cache_seqlens
  // (flash_attn_interface.py:1530-1541)
  Tensor q_in = q, k_rot;
  auto rotate = [&](const Tensor& x, bool per_token) {
    Tensor y = at::empty_like(x);
    FaRotaryParams r{};
    r.x = x.data_ptr(); r.y = y.data_ptr(); r.cos = rotary_cos_->data_ptr(); r.sin = rotary_sin_->data_ptr();
    r.seqlen_offsets = seqlens_k_.has_value() ? seqlens_k_->data_ptr<int>() : nullptr;
    r.x_batch_stride = x.stride(0); r.x_row_stride = x.stride(1); r.x_head_stride = x.stride(2);
    r.y_batch_stride = y.stride(0); r.y_row_stride = y.stride(1); r.y_head_stride = y.stride(2);
    r.cos_row_stride = rotary_cos_->stride(0);
    r.b = (int)x.size(0); r.s = (int)x.size(1); r.h = (int)x.size(2); r.d = (int)x.size(3);
    r.rotary_dim = (int)(2 * rotary_cos_->size(1)); r.seqlen_ro = (int)rotary_cos_->size(0);
    r.interleaved = is_rotary_interleaved; r.per_token = per_token; r.dtype = dtype_code(q);
    fa_check(fa_rotary(&r, cur_stream(q)));
    return y;
  };
  if (rotary_cos_.has_value()) {
    TORCH_CHECK(k_->dim() == 4 && k_->stride(-1) == 1, "key must be 4-D with contiguous last dimension");
    k_rot = rotate(*k_, true);
    q_in = rotate(q, is_causal || window_size_left >= 0 || window_size_right >= 0);
  }
  if (Sq == 1) window_size_right = -1;  // a right bound cannot hide a key from the single, bottom-right aligned query row
  if (k_.has_value()) {  // append first (flash_fwd_kernel.h Append_KV branch)
    TORCH_CHECK(v_.has_value(), "If key is supplied, value must also be passed in");
    TORCH_CHECK(seqlens_k_.has_value(), "If key is supplied, seqlens_k must also be passed in");
    const Tensor &kn = rotary_cos_.has_value() ? k_rot : *k_, &vn = *v_;
    TORCH_CHECK(kn.dtype() == q.dtype() && vn.dtype() == q.dtype(), "Key and value must have the same dtype as query");
    CHECK_DEVICE(kn); CHECK_DEVICE(vn); CHECK_LAST_CONTIG(kn); CHECK_LAST_CONTIG(vn);
    s_new = kn.size(1);
    CHECK_SHAPE(kn, B, s_new, Hk, D); CHECK_SHAPE(vn, B, s_new, Hk, D);
    FaKvAppendParams ap{};
    ap.knew = kn.data_ptr(); ap.vnew = vn.data_ptr(); ap.kcache = kcache.data_ptr(); ap.vcache = vcache.data_ptr();
    ap.knew_batch_stride = kn.stride(0); ap.knew_row_stride = kn.stride(1); ap.knew_head_stride = kn.stride(2);
    ap.vnew_batch_stride = vn.stride(0); ap.vnew_row_stride = vn.stride(1); ap.vnew_head_stride = vn.stride(2);
    ap.kcache_batch_stride = kcache.stride(0); ap.kcache_row_stride = kcache.stride(1); ap.kcache_head_stride = kcache.stride(2);
    ap.vcache_batch_stride = vcache.stride(0); ap.vcache_row_stride = vcache.stride(1); ap.vcache_head_stride = vcache.stride(2);
    ap.seqlens_k = seqlens_k_->data_ptr<int>();
    ap.cache_batch_idx = cache_batch_idx_.has_value() ? cache_batch_idx_->data_ptr<int>() : nullptr;
    ap.block_table = paged ? block_table_->data_ptr<int>() : nullptr;
    ap.block_table_batch_stride = paged ? block_table_->stride(0) : 0;
    ap.page_block_size = (int)page;
    ap.b = B; ap.seqlen_new = (int)s_new; ap.h_k = Hk; ap.d = D; ap.dtype = dtype_code(q);
    fa_check(fa_kvcache_append(&ap, cur_stream(q)));
  }

  // Decode trick of the reference (flash_api.cpp:1346-1353): with one query row the query heads of a KV group
  // become the rows of the score matrix, so K/V are streamed once per KV head.
  const bool swap = Sq == 1 && H > Hk && window_size_left < 0 && !alibi_slopes_.has_value();
  const int64_t ratio = H / Hk;
  Tensor qk = swap ? q_in.reshape({B, Hk, ratio, D}).transpose(1, 2) : q_in;  // (B, rows, heads, D)
  const int64_t rows = swap ? ratio : Sq, heads = swap ? Hk : H;
  Tensor out;
  if (out_.has_value() && !swap) {
    TORCH_CHECK(out_->dtype() == q.dtype(), "Output must have the same dtype as inputs");
    CHECK_DEVICE(*out_); CHECK_LAST_CONTIG(*out_); CHECK_SHAPE(*out_, B, Sq, H, D);
    out = *out_;
  } else {
    out = at::empty({B, rows, heads, D}, q.options());
  }
  Tensor lse = at::empty({B, heads, rows}, q.options().dtype(at::kFloat));
  FaFwdParams a{};
  a.q = qk.data_ptr(); a.k = kcache.data_ptr(); a.v = vcache.data_ptr(); a.o = out.data_ptr(); a.softmax_lse = lse.data_ptr<float>();
  a.q_batch_stride = qk.stride(0); a.q_row_stride = qk.stride(1); a.q_head_stride = qk.stride(2);
  a.k_batch_stride = kcache.stride(0); a.k_row_stride = kcache.stride(1); a.k_head_stride = kcache.stride(2);
  a.v_batch_stride = vcache.stride(0); a.v_row_stride = vcache.stride(1); a.v_head_stride = vcache.stride(2);
  a.o_batch_stride = out.stride(0); a.o_row_stride = out.stride(1); a.o_head_stride = out.stride(2);
  a.seqused_k = seqlens_k_.has_value() ? seqlens_k_->data_ptr<int>() : nullptr;
  a.seqused_k_add = (int)s_new;
  a.cache_batch_idx = cache_batch_idx_.has_value() ? cache_batch_idx_->data_ptr<int>() : nullptr;
  a.block_table = paged ? block_table_->data_ptr<int>() : nullptr;
  a.block_table_batch_stride = paged ? block_table_->stride(0) : 0;
  a.page_block_size = (int)page;
  a.leftpad_k = leftpad_k_.has_value() ? leftpad_k_->data_ptr<int>() : nullptr;
  set_alibi(alibi_slopes_, B, H, a.alibi_slopes, a.alibi_batch_stride);
  a.b = B; a.h = heads; a.h_k = Hk; a.d = D; a.seqlen_q = (int)rows; a.seqlen_k = (int)Sk; a.total_q = B * rows;
  a.dtype = dtype_code(q);
  a.is_causal = is_causal; a.window_left = (int)window_size_left; a.window_right = (int)window_size_right;
  a.softmax_scale = (float)softmax_scale; a.softcap = (float)softcap;
  a.num_splits = (int)num_splits;
  Tensor ws;  // split-KV partials (reference softmax_lse_accum / out_accum, flash_api.cpp:320-345), freed in stream order
  const int64_t ws_bytes = fa_fwd_workspace_bytes(&a);
  if (ws_bytes > 0) {
    ws = at::empty({ws_bytes}, q.options().dtype(at::kByte));
    a.workspace = ws.data_ptr(); a.workspace_bytes = ws_bytes;
  }
  fa_check(fa_fwd_kvcache(&a, cur_stream(q)));
  if (swap) {
    Tensor o2 = out.transpose(1, 2).reshape({B, 1, H, D});
    if (out_.has_value()) { out_->copy_(o2); o2 = *out_; }
    out = o2;
    lse = lse.reshape({B, H, 1});
  }
  return {out, lse};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "MI355X-native FlashAttention backend (gfx950 HIP kernels behind the flash_attn_2_cuda module API)";
  m.def("fwd", &mha_fwd, "Forward pass");
  m.def("varlen_fwd", &mha_varlen_fwd, "Forward pass (variable length)", pybind11::arg("q"), pybind11::arg("k"), pybind11::arg("v"),
        pybind11::arg("out_"), pybind11::arg("cu_seqlens_q"), pybind11::arg("cu_seqlens_k"), pybind11::arg("seqused_k"),
        pybind11::arg("leftpad_k_"), pybind11::arg("block_table_"), pybind11::arg("alibi_slopes_"), pybind11::arg("max_seqlen_q"),
        pybind11::arg("max_seqlen_k"), pybind11::arg("p_dropout"), pybind11::arg("softmax_scale"), pybind11::arg("zero_tensors"),
        pybind11::arg("is_causal"), pybind11::arg("window_size_left"), pybind11::arg("window_size_right"), pybind11::arg("softcap"),
        pybind11::arg("return_softmax"), pybind11::arg("gen_"), pybind11::arg("num_splits") = 0);
  m.def("bwd", &mha_bwd, "Backward pass");
  m.def("varlen_bwd", &mha_varlen_bwd, "Backward pass (variable length)");
  m.def("fwd_kvcache", &mha_fwd_kvcache, "Forward pass, with KV-cache");
}
