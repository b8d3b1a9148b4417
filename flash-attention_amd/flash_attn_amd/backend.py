"""Backend-module API of the reference (``flash_attn_2_cuda``) on top of the C ABI, via ctypes.

Exports ``fwd``, ``varlen_fwd``, ``bwd``, ``varlen_bwd`` (and a raising ``fwd_kvcache``) with the
exact positional signatures the reference custom ops use
(flash_attn/flash_attn_interface.py:99,177,280,380; C++ originals csrc/flash_attn/flash_api.cpp
:368-382, :538-561, :800-820, :1010-1035).  Validation messages follow the reference's
``TORCH_CHECK`` texts where tests match on them.  torch is used for device memory and the
current stream only; all arithmetic happens in libfa_gfx950.so.  No fallback path exists.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import torch

from . import _cabi

_NATIVE_HEAD_DIMS = (32, 64, 96, 128, 192, 256)  # head dims with their own kernels (csrc/fa_api.cpp head_dim_native)


def reload_knobs() -> None:
    """Re-read the FA_* environment knobs (the library reads them once per process)."""
    _cabi.load().fa_knobs_reload()


_SCHED_FIELDS = ("fwd_kernel", "fwd_nw", "fwd_feat", "fwd_splits", "fwd_list", "d", "bf16", "bwd_dq_nw", "bwd_list", "bwd_spill", "fwd_pack", "bwd_dkdv_nw")
FWD_KERNEL_NAMES = {0: "none", 1: "fa_fwd_kernel", 2: "fa_fwd_il_kernel", 3: "fa_fwd_w64_kernel"}


def last_schedule() -> dict:
    """Kernels enqueued by this thread's last forward / backward call (C ABI fa_last_schedule)."""
    import ctypes as C
    lib = _cabi.load()
    buf = (C.c_int32 * len(_SCHED_FIELDS))()
    lib.fa_last_schedule(buf, len(_SCHED_FIELDS))
    d = dict(zip(_SCHED_FIELDS, [int(x) for x in buf]))
    d["name"] = lib.fa_last_kernel_name().decode()
    return d


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return _cabi.FA_DTYPE_BF16
    if t.dtype == torch.float16:
        return _cabi.FA_DTYPE_FP16
    raise RuntimeError("FlashAttention only support fp16 and bf16 data type")


def _check_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("Input tensor must be on CUDA device")


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _check_d(d: int) -> None:
    """The C ABI takes every head dim that is a multiple of 8 up to 256 as it is (the six built sizes directly, the sizes in between through the
    kernels' run-time column bound, fa_gfx950.h FaFwdParams::d): no padded copies on this side either."""
    if d > 256:
        raise RuntimeError(f"FlashAttention only supports head dimension at most 256 (got {d})")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _common_checks(q, k, v, p_dropout, alibi_slopes, gen_):
    if gen_ is not None:
        raise RuntimeError("Passing a `generator` argument is no longer supported; seed the default generator instead")
    if not (0.0 <= p_dropout < 1.0):
        raise RuntimeError("p_dropout must be in [0, 1)")
    _check_dev(q, k, v)
    if not (q.dtype == k.dtype == v.dtype):
        raise RuntimeError("query, key and value must have the same dtype")
    for t in (q, k, v):
        if t.stride(-1) != 1:
            raise RuntimeError("Input tensor must have contiguous last dimension")
    if alibi_slopes is not None:
        if alibi_slopes.dtype != torch.float32:
            raise RuntimeError("ALiBi slopes must have dtype fp32")
        if alibi_slopes.stride(-1) != 1:
            raise RuntimeError("ALiBi slopes tensor must have contiguous last dimension")


def _new_rng_state(device, p_dropout, batch, nheads):
    """(seed, offset) of the default generator of ``device`` as the device int64[2] the reference returns
    (flash_api.cpp:496-515); the generator's Philox offset advances as in csrc/flash_attn_ck/mha_fwd.cpp:283-295."""
    rng_state = torch.empty((2,), dtype=torch.int64, device=device)
    if p_dropout > 0.0:
        gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
        seed, off = gen.initial_seed(), gen.get_offset()
        gen.set_offset(off + ((batch * nheads * 64 + 3) // 4) * 4)
        with torch.cuda.device(device):
            _cabi.check(_cabi.load().fa_set_rng_state(C.c_uint64(seed & (2 ** 64 - 1)), C.c_uint64(off), _ptr(rng_state),
                                                      C.c_void_p(_stream_ptr(device))))
    return rng_state


def _alibi_args(alibi_slopes, batch, nheads):
    if alibi_slopes is None:
        return None, 0
    if tuple(alibi_slopes.shape) not in ((nheads,), (batch, nheads)):
        raise RuntimeError("ALiBi slopes must have shape (nheads,) or (batch_size, nheads)")
    return alibi_slopes, (alibi_slopes.stride(0) if alibi_slopes.dim() == 2 else 0)


def fwd(q, k, v, out_, alibi_slopes_, p_dropout, softmax_scale, is_causal, window_size_left,
        window_size_right, softcap, return_softmax, gen_) -> List[torch.Tensor]:
    """mha_fwd (flash_api.cpp:368-536): q (B,Sq,H,D), k/v (B,Sk,Hk,D) -> [out, softmax_lse, p, rng_state]."""
    _common_checks(q, k, v, p_dropout, alibi_slopes_, gen_)
    if return_softmax and not p_dropout > 0.0:
        raise RuntimeError("return_softmax is only supported when p_dropout > 0.0")
    B, Sq, H, D = q.shape
    Sk, Hk = k.shape[1], k.shape[2]
    if B <= 0:
        raise RuntimeError("batch size must be positive")
    if D > 256:
        raise RuntimeError("FlashAttention forward only supports head dimension at most 256")
    if D % 8 != 0:
        raise RuntimeError("query, key, value, and out_ must have a head_size that is a multiple of 8")
    if H % Hk != 0:
        raise RuntimeError("Number of heads in key/value must divide number of heads in query")
    if tuple(k.shape) != (B, Sk, Hk, D) or tuple(v.shape) != (B, Sk, Hk, D):
        raise RuntimeError("key/value shape mismatch")
    if out_ is not None:
        if out_.dtype != q.dtype or tuple(out_.shape) != (B, Sq, H, D) or out_.stride(-1) != 1:
            raise RuntimeError("out_ must have the same dtype/shape as q and a contiguous last dimension")
    # One query row and grouped heads: the query heads of a KV group become the rows of one block, K/V are streamed once
    # per KV head (seqlenq_ngroups_swapped, flash_api.cpp:429-437 and :531-535)
    if (Sq == 1 and H > Hk and window_size_left < 0 and window_size_right < 0 and p_dropout == 0.0 and alibi_slopes_ is None
            and Sk > 0):
        ng = H // Hk
        o2, l2, p2, r2 = fwd(q.reshape(B, Hk, ng, D).transpose(1, 2), k, v, None, None, 0.0, softmax_scale, False, -1, -1,
                             softcap, False, None)
        out = o2.transpose(1, 2).reshape(B, 1, H, D)
        if out_ is not None:
            out_.copy_(out)
            out = out_
        return [out, l2.reshape(B, H, 1), p2, r2]
    _check_d(D)
    qp, kp, vp = q, k, v
    out = out_ if out_ is not None else torch.empty((B, Sq, H, D), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)
    rng_state = _new_rng_state(q.device, p_dropout, B, H)
    # return_softmax: the random byte of every (query, key) pair, the ROCm backend's payload (mha_fwd.cpp:275-279)
    p_out = (torch.zeros((B, H, Sq, Sk), dtype=torch.uint8, device=q.device) if return_softmax
             else torch.empty((0,), dtype=q.dtype, device=q.device))
    if Sk == 0:  # flash_api.cpp:524-528
        out.zero_()
        lse.fill_(float("inf"))
    elif Sq > 0:
        alibi, alibi_bs = _alibi_args(alibi_slopes_, B, H)
        a = _cabi.FaFwdParams()
        a.q, a.k, a.v, a.o, a.softmax_lse = _ptr(qp), _ptr(kp), _ptr(vp), _ptr(out), _ptr(lse)
        a.q_batch_stride, a.q_row_stride, a.q_head_stride = qp.stride(0), qp.stride(1), qp.stride(2)
        a.k_batch_stride, a.k_row_stride, a.k_head_stride = kp.stride(0), kp.stride(1), kp.stride(2)
        a.v_batch_stride, a.v_row_stride, a.v_head_stride = vp.stride(0), vp.stride(1), vp.stride(2)
        a.o_batch_stride, a.o_row_stride, a.o_head_stride = out.stride(0), out.stride(1), out.stride(2)
        a.alibi_slopes, a.alibi_batch_stride = _ptr(alibi), alibi_bs
        a.b, a.h, a.h_k, a.d = B, H, Hk, D
        a.seqlen_q, a.seqlen_k, a.total_q = Sq, Sk, B * Sq
        a.dtype = _dtype_code(q)
        a.is_causal, a.window_left, a.window_right = int(bool(is_causal)), int(window_size_left), int(window_size_right)
        a.softmax_scale, a.softcap = float(softmax_scale), float(softcap)
        if p_dropout > 0.0:
            a.p_dropout, a.rng_state = float(p_dropout), _ptr(rng_state)
            if return_softmax:
                a.randval = _ptr(p_out)
                a.randval_batch_stride, a.randval_head_stride, a.randval_row_stride = p_out.stride(0), p_out.stride(1), p_out.stride(2)
        with torch.cuda.device(q.device):
            _cabi.check(_cabi.load().fa_fwd(C.byref(a), C.c_void_p(_stream_ptr(q.device))))
    return [out, lse, p_out, rng_state]


def varlen_fwd(q, k, v, out_, cu_seqlens_q, cu_seqlens_k, seqused_k, leftpad_k_, block_table_,
               alibi_slopes_, max_seqlen_q, max_seqlen_k, p_dropout, softmax_scale, zero_tensors,
               is_causal, window_size_left, window_size_right, softcap, return_softmax, gen_,
               num_splits: int = 0, seqused_q: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """mha_varlen_fwd (flash_api.cpp:538-788): packed q (total_q,H,D), k/v (total_k,Hk,D), int32 cu_seqlens (B+1).
    seqused_q (extension, not in the reference's signature): only the first seqused_q[b] rows of entry b are queries -- with seqused_k the
    in-place form of a padded batch (flash_attn_interface.flash_attn_padded_func)."""
    _common_checks(q, k, v, p_dropout, alibi_slopes_, gen_)
    if return_softmax and not p_dropout > 0.0:
        raise RuntimeError("return_softmax is only supported when p_dropout > 0.0")
    paged = block_table_ is not None  # k / v are (num_blocks, page, Hk, D), addressed through block_table (B, max_blocks)
    if paged:
        if block_table_.dtype != torch.int32 or block_table_.stride(-1) != 1:
            raise RuntimeError("block_table must have dtype torch.int32 and a contiguous last dimension")
        if k.dim() != 4 or v.dim() != 4:
            raise RuntimeError("With block_table, k and v must be 4-D (num_blocks, page_block_size, nheads_k, headdim)")
        if k.shape[1] % 256 != 0:
            raise RuntimeError("Paged KV cache block size must be divisible by 256")
        if leftpad_k_ is not None:
            raise RuntimeError("We don't support Paged KV and leftpad_k running at the same time yet")
    if leftpad_k_ is not None and (leftpad_k_.dtype != torch.int32 or not leftpad_k_.is_contiguous()):
        raise RuntimeError("leftpad_k must be a contiguous int32 tensor of shape (batch_size)")
    if num_splits > 1:
        raise RuntimeError("num_splits > 1 is not supported")
    for cu in (cu_seqlens_q, cu_seqlens_k):
        if cu.dtype != torch.int32:
            raise RuntimeError("cu_seqlens_q/k must have dtype int32")
        if not cu.is_contiguous():
            raise RuntimeError("cu_seqlens_q/k must be contiguous")
    _check_dev(cu_seqlens_q, cu_seqlens_k, seqused_k, block_table_, leftpad_k_)
    total_q, H, D = q.shape
    total_k, Hk = (k.shape[0] * k.shape[1], k.shape[2]) if paged else (k.shape[0], k.shape[1])
    B = cu_seqlens_q.numel() - 1
    if paged and tuple(block_table_.shape[:1]) != (B,):
        raise RuntimeError("block_table must have shape (batch_size, max_num_blocks_per_seq)")
    if B <= 0:
        raise RuntimeError("batch size must be positive")
    if cu_seqlens_k.numel() != B + 1:
        raise RuntimeError("cu_seqlens_k must have shape (batch_size + 1)")
    if D > 256 or D % 8 != 0:
        raise RuntimeError("head_size must be a multiple of 8 and at most 256")
    if H % Hk != 0:
        raise RuntimeError("Number of heads in key/value must divide number of heads in query")
    # One query row per sequence and grouped heads (decode over a packed batch): the query heads of a KV group become the
    # rows of one block (seqlenq_ngroups_swapped, flash_api.cpp:620-629 and :776-782); q is then ngroups rows per sequence
    if (max_seqlen_q == 1 and total_q == B and H > Hk and window_size_left < 0 and window_size_right < 0 and p_dropout == 0.0
            and alibi_slopes_ is None and max_seqlen_k > 0 and total_k > 0):
        ng = H // Hk
        q2 = q.reshape(B, Hk, ng, D).transpose(1, 2).reshape(B * ng, Hk, D)
        cu_q2 = torch.arange(0, (B + 1) * ng, ng, dtype=torch.int32, device=q.device)
        o2, l2, p2, r2 = varlen_fwd(q2, k, v, None, cu_q2, cu_seqlens_k, seqused_k, leftpad_k_, block_table_, None, ng,
                                    max_seqlen_k, 0.0, softmax_scale, zero_tensors, False, -1, -1, softcap, False, None, num_splits)
        out = o2.reshape(B, ng, Hk, D).transpose(1, 2).reshape(B, H, D)
        if out_ is not None:
            out_.copy_(out)
            out = out_
        # lse (Hk, B*ng) -> (H, B): head hk*ng + g of sequence b sits at [hk][b*ng + g]
        return [out, l2.reshape(Hk, B, ng).permute(0, 2, 1).reshape(H, B).contiguous(), p2, r2]
    if seqused_k is not None and (seqused_k.dtype != torch.int32 or seqused_k.numel() != B or not seqused_k.is_contiguous()):
        raise RuntimeError("seqused_k must be a contiguous int32 tensor of shape (batch_size)")
    if seqused_q is not None and (seqused_q.dtype != torch.int32 or seqused_q.numel() != B or not seqused_q.is_contiguous()):
        raise RuntimeError("seqused_q must be a contiguous int32 tensor of shape (batch_size)")
    _check_dev(seqused_q)
    _check_d(D)
    qp, kp, vp = q, k, v
    out = out_ if out_ is not None else torch.empty((total_q, H, D), dtype=q.dtype, device=q.device)
    lse = torch.empty((H, total_q), dtype=torch.float32, device=q.device)
    rng_state = _new_rng_state(q.device, p_dropout, B, H)
    # varlen payload layout of the ROCm backend: (nheads, total_q, max_seqlen_k) (mha_varlen_fwd.cpp)
    p_out = (torch.zeros((H, total_q, int(max_seqlen_k)), dtype=torch.uint8, device=q.device) if return_softmax
             else torch.empty((0,), dtype=q.dtype, device=q.device))
    if zero_tensors:  # flash_api.cpp:693-697
        out.zero_()
        lse.fill_(float("-inf"))
    if max_seqlen_k == 0 or total_k == 0:
        out.zero_()
        lse.fill_(float("inf"))
    elif total_q > 0 and max_seqlen_q > 0:
        alibi, alibi_bs = _alibi_args(alibi_slopes_, B, H)
        a = _cabi.FaFwdParams()
        a.q, a.k, a.v, a.o, a.softmax_lse = _ptr(qp), _ptr(kp), _ptr(vp), _ptr(out), _ptr(lse)
        a.q_row_stride, a.q_head_stride = qp.stride(0), qp.stride(1)
        if paged:
            a.k_batch_stride, a.k_row_stride, a.k_head_stride = kp.stride(0), kp.stride(1), kp.stride(2)
            a.v_batch_stride, a.v_row_stride, a.v_head_stride = vp.stride(0), vp.stride(1), vp.stride(2)
            a.block_table, a.block_table_batch_stride, a.page_block_size = _ptr(block_table_), block_table_.stride(0), kp.shape[1]
        else:
            a.k_row_stride, a.k_head_stride = kp.stride(0), kp.stride(1)
            a.v_row_stride, a.v_head_stride = vp.stride(0), vp.stride(1)
        a.leftpad_k = _ptr(leftpad_k_)
        a.o_row_stride, a.o_head_stride = out.stride(0), out.stride(1)
        a.cu_seqlens_q, a.cu_seqlens_k, a.seqused_k = _ptr(cu_seqlens_q), _ptr(cu_seqlens_k), _ptr(seqused_k)
        a.seqused_q = _ptr(seqused_q)
        a.alibi_slopes, a.alibi_batch_stride = _ptr(alibi), alibi_bs
        a.b, a.h, a.h_k, a.d = B, H, Hk, D
        a.seqlen_q, a.seqlen_k, a.total_q = int(max_seqlen_q), int(max_seqlen_k), total_q
        a.dtype = _dtype_code(q)
        a.is_causal, a.window_left, a.window_right = int(bool(is_causal)), int(window_size_left), int(window_size_right)
        a.softmax_scale, a.softcap = float(softmax_scale), float(softcap)
        if p_dropout > 0.0:
            a.p_dropout, a.rng_state = float(p_dropout), _ptr(rng_state)
            if return_softmax:
                a.randval = _ptr(p_out)
                a.randval_batch_stride, a.randval_head_stride, a.randval_row_stride = 0, p_out.stride(0), p_out.stride(1)
        with torch.cuda.device(q.device):
            lib = _cabi.load()
            ws_bytes = lib.fa_fwd_workspace_bytes(C.byref(a))  # work list of an uneven packed batch (0 = dense grid)
            if ws_bytes > 0:
                ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=q.device)
                a.workspace, a.workspace_bytes = _ptr(ws), ws_bytes
            _cabi.check(lib.fa_varlen_fwd(C.byref(a), C.c_void_p(_stream_ptr(q.device))))
    return [out, lse, p_out, rng_state]


def _bwd_out(buf, like, name):
    if buf is None:
        return torch.empty_like(like)
    if buf.dtype != like.dtype or tuple(buf.shape) != tuple(like.shape) or buf.stride(-1) != 1:
        raise RuntimeError(f"{name} must have the same dtype/shape as its input and a contiguous last dimension")
    return buf


def _bwd_rng(p_dropout, rng_state, device):
    if not p_dropout > 0.0:
        return None
    if rng_state is None:
        raise RuntimeError("p_dropout > 0 in the backward needs the forward's rng_state")
    if rng_state.dtype != torch.int64 or rng_state.numel() != 2 or not rng_state.is_cuda:
        raise RuntimeError("rng_state must be a CUDA int64 tensor of shape (2,)")
    return rng_state


def _fill_bwd_common(a, dout, q, k, v, out, lse, dq, dk, dv, delta, alibi, alibi_bs, softmax_scale,
                     is_causal, wl, wr, softcap, deterministic):
    a.dout, a.q, a.k, a.v, a.o, a.softmax_lse = _ptr(dout), _ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse)
    a.dq, a.dk, a.dv, a.softmax_d = _ptr(dq), _ptr(dk), _ptr(dv), _ptr(delta)
    a.alibi_slopes, a.alibi_batch_stride = _ptr(alibi), alibi_bs
    a.dtype = _dtype_code(q)
    a.is_causal, a.window_left, a.window_right = int(bool(is_causal)), int(wl), int(wr)
    a.softmax_scale, a.softcap, a.deterministic = float(softmax_scale), float(softcap), int(bool(deterministic))


def _alloc_workspace(nbytes, device):   # (a seam for tests/test_bwd_schedules_gpu.py: the out-of-memory fallback)
    return torch.empty((nbytes,), dtype=torch.uint8, device=device)


def _run_bwd(a, device, varlen):
    lib = _cabi.load()
    ws_bytes = lib.fa_bwd_workspace_bytes(C.byref(a))
    ws = None
    if ws_bytes > 0:
        # A fixed-length call's workspace is the dS area of the 5-contraction launches (up to 1 GiB by default): a speed-up, not a requirement.  When the allocator
        # cannot supply it the call proceeds without one and the library runs the recomputing pair (fa_api.cpp do_bwd checks workspace_bytes).
        try:
            ws = _alloc_workspace(ws_bytes, device)
        except torch.OutOfMemoryError:
            if varlen:   # (work lists: a few KB -- if that fails nothing will succeed)
                raise
            ws = None
    if ws is not None:
        if os.environ.get("FA_DEBUG_POISON_WS"):   # tests: every 16-bit word of the workspace a NaN, so that a read of something nobody wrote shows
            ws.fill_(0xFF)
        a.workspace, a.workspace_bytes = _ptr(ws), ws_bytes
    with torch.cuda.device(device):
        fn = lib.fa_varlen_bwd if varlen else lib.fa_bwd
        _cabi.check(fn(C.byref(a), C.c_void_p(_stream_ptr(device))))
        if ws is not None and not varlen and last_schedule().get("bwd_spill") == 3:   # the opt-in fused backward: its hand-offs can time out (fa_gfx950.h)
            _cabi.check(lib.fa_bwd_fused_status(C.byref(a), C.c_void_p(_stream_ptr(device))))
    return ws


def bwd(dout, q, k, v, out, softmax_lse, dq_, dk_, dv_, alibi_slopes_, p_dropout, softmax_scale,
        is_causal, window_size_left, window_size_right, softcap, deterministic, gen_, rng_state
        ) -> List[torch.Tensor]:
    """mha_bwd (flash_api.cpp:800-1008) -> [dq, dk, dv, softmax_d]."""
    _common_checks(q, k, v, p_dropout, alibi_slopes_, gen_)
    _check_dev(dout, out, softmax_lse)
    if dout.dtype != q.dtype or out.dtype != q.dtype:
        raise RuntimeError("query and dout/out must have the same dtype")
    if dout.stride(-1) != 1 or out.stride(-1) != 1:
        raise RuntimeError("out/dout tensor must have contiguous last dimension")
    B, Sq, H, D = q.shape
    Sk, Hk = k.shape[1], k.shape[2]
    if D % 8 != 0 or D > 256:
        raise RuntimeError("head_size should be a multiple of 8 and at most 256")
    if H % Hk != 0:
        raise RuntimeError("Number of heads in key/value must divide number of heads in query")
    dq, dk, dv = _bwd_out(dq_, q, "dq"), _bwd_out(dk_, k, "dk"), _bwd_out(dv_, v, "dv")
    delta = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)
    if Sq == 0 or Sk == 0:  # flash_api.cpp:992-999
        dq.zero_(); dk.zero_(); dv.zero_(); delta.zero_()
        return [dq, dk, dv, delta]
    _check_d(D)
    dop, qp, kp, vp, op, dqp, dkp, dvp = dout, q, k, v, out, dq, dk, dv
    alibi, alibi_bs = _alibi_args(alibi_slopes_, B, H)
    a = _cabi.FaBwdParams()
    _fill_bwd_common(a, dop, qp, kp, vp, op, softmax_lse, dqp, dkp, dvp, delta, alibi, alibi_bs,
                     softmax_scale, is_causal, window_size_left, window_size_right, softcap, deterministic)
    for nm, t in (("do", dop), ("q", qp), ("k", kp), ("v", vp), ("o", op), ("dq", dqp), ("dk", dkp), ("dv", dvp)):
        setattr(a, nm + "_batch_stride", t.stride(0))
        setattr(a, nm + "_row_stride", t.stride(1))
        setattr(a, nm + "_head_stride", t.stride(2))
    a.b, a.h, a.h_k, a.d = B, H, Hk, D
    a.seqlen_q, a.seqlen_k, a.total_q, a.total_k = Sq, Sk, B * Sq, B * Sk
    rng = _bwd_rng(p_dropout, rng_state, q.device)
    if rng is not None:
        a.p_dropout, a.rng_state = float(p_dropout), _ptr(rng)
    _run_bwd(a, q.device, False)
    return [dq, dk, dv, delta]


def varlen_bwd(dout, q, k, v, out, softmax_lse, dq_, dk_, dv_, cu_seqlens_q, cu_seqlens_k,
               alibi_slopes_, max_seqlen_q, max_seqlen_k, p_dropout, softmax_scale, zero_tensors,
               is_causal, window_size_left, window_size_right, softcap, deterministic, gen_, rng_state,
               seqused_q: Optional[torch.Tensor] = None, seqused_k: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """mha_varlen_bwd (flash_api.cpp:1010-1241) -> [dq, dk, dv, softmax_d].  seqused_q / seqused_k (extension): as in varlen_fwd; gradient
    rows past them are not written."""
    _common_checks(q, k, v, p_dropout, alibi_slopes_, gen_)
    _check_dev(dout, out, softmax_lse, cu_seqlens_q, cu_seqlens_k)
    for cu in (cu_seqlens_q, cu_seqlens_k):
        if cu.dtype != torch.int32 or not cu.is_contiguous():
            raise RuntimeError("cu_seqlens_q/k must be contiguous int32")
    total_q, H, D = q.shape
    total_k, Hk = k.shape[0], k.shape[1]
    B = cu_seqlens_q.numel() - 1
    if D % 8 != 0 or D > 256:
        raise RuntimeError("head_size should be a multiple of 8 and at most 256")
    dq, dk, dv = _bwd_out(dq_, q, "dq"), _bwd_out(dk_, k, "dk"), _bwd_out(dv_, v, "dv")
    delta = torch.empty((H, total_q), dtype=torch.float32, device=q.device)
    if zero_tensors:  # flash_api.cpp:1171-1176
        dq.zero_(); dk.zero_(); dv.zero_(); delta.zero_()
    if max_seqlen_q == 0 or total_q == 0 or total_k == 0:
        dk.zero_(); dv.zero_(); delta.zero_(); dq.zero_()
        return [dq, dk, dv, delta]
    _check_d(D)
    dop, qp, kp, vp, op, dqp, dkp, dvp = dout, q, k, v, out, dq, dk, dv
    alibi, alibi_bs = _alibi_args(alibi_slopes_, B, H)
    a = _cabi.FaBwdParams()
    _fill_bwd_common(a, dop, qp, kp, vp, op, softmax_lse, dqp, dkp, dvp, delta, alibi, alibi_bs,
                     softmax_scale, is_causal, window_size_left, window_size_right, softcap, deterministic)
    for nm, t in (("do", dop), ("q", qp), ("k", kp), ("v", vp), ("o", op), ("dq", dqp), ("dk", dkp), ("dv", dvp)):
        setattr(a, nm + "_row_stride", t.stride(0))
        setattr(a, nm + "_head_stride", t.stride(1))
    a.cu_seqlens_q, a.cu_seqlens_k = _ptr(cu_seqlens_q), _ptr(cu_seqlens_k)
    for nm, t in (("seqused_q", seqused_q), ("seqused_k", seqused_k)):
        if t is not None and (t.dtype != torch.int32 or t.numel() != B or not t.is_contiguous()):
            raise RuntimeError(nm + " must be a contiguous int32 tensor of shape (batch_size)")
    _check_dev(seqused_q, seqused_k)
    a.seqused_q, a.seqused_k = _ptr(seqused_q), _ptr(seqused_k)
    a.b, a.h, a.h_k, a.d = B, H, Hk, D
    a.seqlen_q, a.seqlen_k, a.total_q, a.total_k = int(max_seqlen_q), int(max_seqlen_k), total_q, total_k
    rng = _bwd_rng(p_dropout, rng_state, q.device)
    if rng is not None:
        a.p_dropout, a.rng_state = float(p_dropout), _ptr(rng)
    _run_bwd(a, q.device, True)
    return [dq, dk, dv, delta]


def _rotary(lib, x, cos, sin, offsets, interleaved, per_token):
    """x (B,S,H,D) -> rotated copy (fa_rotary; reference layers/rotary.py apply_rotary_emb semantics)."""
    y = torch.empty_like(x)
    r = _cabi.FaRotaryParams()
    r.x, r.y, r.cos, r.sin, r.seqlen_offsets = _ptr(x), _ptr(y), _ptr(cos), _ptr(sin), _ptr(offsets)
    r.x_batch_stride, r.x_row_stride, r.x_head_stride = x.stride(0), x.stride(1), x.stride(2)
    r.y_batch_stride, r.y_row_stride, r.y_head_stride = y.stride(0), y.stride(1), y.stride(2)
    r.cos_row_stride = cos.stride(0)
    r.b, r.s, r.h, r.d = x.shape
    r.rotary_dim, r.seqlen_ro = 2 * cos.shape[1], cos.shape[0]
    r.interleaved, r.per_token, r.dtype = int(bool(interleaved)), int(bool(per_token)), _dtype_code(x)
    _cabi.check(lib.fa_rotary(C.byref(r), C.c_void_p(_stream_ptr(x.device))))
    return y


def fwd_kvcache(q, kcache, vcache, k_, v_, seqlens_k_, rotary_cos_, rotary_sin_, cache_batch_idx_, leftpad_k_, block_table_,
                alibi_slopes_, out_, softmax_scale, is_causal, window_size_left, window_size_right, softcap,
                is_rotary_interleaved, num_splits) -> List[torch.Tensor]:
    """mha_fwd_kvcache (flash_api.cpp:1243-1532) -> [out, softmax_lse]."""
    _check_dev(q, kcache, vcache, k_, v_, seqlens_k_, cache_batch_idx_, block_table_, rotary_cos_, rotary_sin_, leftpad_k_)
    if not (q.dtype == kcache.dtype == vcache.dtype):
        raise RuntimeError("query and key must have the same dtype")
    paged = block_table_ is not None
    if paged and cache_batch_idx_ is not None:
        raise RuntimeError("Paged KVcache does not support cache_batch_idx")
    if leftpad_k_ is not None:
        if paged:
            raise RuntimeError("We don't support Paged KV and leftpad_k running at the same time yet")
        if leftpad_k_.dtype != torch.int32 or not leftpad_k_.is_contiguous() or leftpad_k_.numel() != q.shape[0]:
            raise RuntimeError("leftpad_k must be a contiguous int32 tensor of shape (batch_size)")
        if seqlens_k_ is None:
            raise RuntimeError("leftpad_k needs seqlens_k (cache_seqlens)")
    if rotary_cos_ is not None:
        if k_ is None:
            raise RuntimeError("If rotary cos/sin are provided, new key / value to be appended to KV cache must also be provided")
        if rotary_sin_ is None or rotary_cos_.shape != rotary_sin_.shape or rotary_cos_.dim() != 2:
            raise RuntimeError("rotary_cos and rotary_sin must both be (seqlen_ro, rotary_dim / 2)")
        if 2 * rotary_cos_.shape[1] > q.shape[-1]:
            raise RuntimeError("rotary_dim must be <= headdim")
        if (2 * rotary_cos_.shape[1]) % 16 != 0:
            raise RuntimeError("Only rotary dimensions divisible by 16 are currently supported")
        if rotary_cos_.dtype != q.dtype or rotary_sin_.dtype != q.dtype:
            raise RuntimeError("rotary_cos/sin must have the same dtype as query")
        if rotary_cos_.stride(-1) != 1 or rotary_sin_.stride(-1) != 1 or rotary_cos_.stride(0) != rotary_sin_.stride(0):
            raise RuntimeError("rotary_cos/sin must have contiguous last dimension and equal row strides")
    B, Sq, H, D = q.shape
    Hk = kcache.shape[2]
    page = kcache.shape[1] if paged else 0
    Sk = block_table_.shape[1] * page if paged else kcache.shape[1]
    if D > 256:
        raise RuntimeError("FlashAttention forward only supports head dimension at most 256")
    if D % 8 != 0:
        # flash_api.cpp:1340-1350, 1517-1527: q and both caches zero-padded to the next multiple of 8 (whole-cache copies, as in the reference), the call
        # runs on the copies, appended keys / values are copied back
        pad = 8 - D % 8
        P = lambda t: None if t is None else torch.nn.functional.pad(t, (0, pad))
        kc_p, vc_p = P(kcache), P(vcache)
        o, lse = fwd_kvcache(P(q), kc_p, vc_p, P(k_), P(v_), seqlens_k_, rotary_cos_, rotary_sin_, cache_batch_idx_, leftpad_k_, block_table_,
                             alibi_slopes_, None, softmax_scale, is_causal, window_size_left, window_size_right, softcap,
                             is_rotary_interleaved, num_splits)
        o = o[..., :D]
        if out_ is not None:
            out_.copy_(o)
            o = out_
        if k_ is not None:
            kcache.copy_(kc_p[..., :D])
            vcache.copy_(vc_p[..., :D])
        return [o, lse]
    if H % Hk != 0:
        raise RuntimeError("Number of heads in key/value must divide number of heads in query")
    if paged and page % 256 != 0:
        raise RuntimeError("Paged KV cache block size must be divisible by 256")
    if Sq == 1 and alibi_slopes_ is None:
        is_causal = False
    lib = _cabi.load()
    s_new = 0 if k_ is None else k_.shape[1]
    if s_new > Sk:      # flash_api.cpp:1397
        raise RuntimeError("If key is supplied, it must have seqlen <= the seqlen of the KV cache")
    if rotary_cos_ is not None and rotary_cos_.shape[0] < Sk:   # flash_api.cpp:1470
        raise RuntimeError("cos/sin seqlen must be at least the seqlen of KV cache")
    if paged and seqlens_k_ is not None:  # the reference's guard (flash_api.cpp:1433-1447); costs a device->host sync
        need = int(seqlens_k_.max().item()) + s_new
        if need > Sk:
            raise RuntimeError(f"Paged KV cache: max(seqlens_k){' + seqlen_knew' if s_new else ''} (= {need}) exceeds the capacity "
                               f"addressable by block_table (max_num_blocks_per_seq * page_block_size = {Sk})")
    with torch.cuda.device(q.device):
        if rotary_cos_ is not None:  # keys at positions cache_seqlens + i; queries too if causal/local, else all at cache_seqlens
            local = is_causal or window_size_left >= 0 or window_size_right >= 0
            k_ = _rotary(lib, k_, rotary_cos_, rotary_sin_, seqlens_k_, is_rotary_interleaved, True)
            q = _rotary(lib, q, rotary_cos_, rotary_sin_, seqlens_k_, is_rotary_interleaved, local)
        if k_ is not None:
            if v_ is None or seqlens_k_ is None:
                raise RuntimeError("If key is supplied, value and seqlens_k must also be passed in")
            s_new = k_.shape[1]
            ap = _cabi.FaKvAppendParams()
            ap.knew, ap.vnew, ap.kcache, ap.vcache = _ptr(k_), _ptr(v_), _ptr(kcache), _ptr(vcache)
            for nm, t in (("knew", k_), ("vnew", v_), ("kcache", kcache), ("vcache", vcache)):
                setattr(ap, nm + "_batch_stride", t.stride(0)); setattr(ap, nm + "_row_stride", t.stride(1)); setattr(ap, nm + "_head_stride", t.stride(2))
            ap.seqlens_k, ap.cache_batch_idx = _ptr(seqlens_k_), _ptr(cache_batch_idx_)
            ap.block_table, ap.block_table_batch_stride = _ptr(block_table_), (block_table_.stride(0) if paged else 0)
            ap.page_block_size = page
            ap.b, ap.seqlen_new, ap.h_k, ap.d, ap.dtype = B, s_new, Hk, D, _dtype_code(q)
            _cabi.check(lib.fa_kvcache_append(C.byref(ap), C.c_void_p(_stream_ptr(q.device))))
        if Sq == 1:
            window_size_right = -1  # a right bound cannot hide a key from the single, bottom-right aligned query row
        swap = Sq == 1 and H > Hk and window_size_left < 0 and alibi_slopes_ is None
        ratio = H // Hk
        qk = q.reshape(B, Hk, ratio, D).transpose(1, 2) if swap else q
        rows, heads = (ratio, Hk) if swap else (Sq, H)
        out = out_ if (out_ is not None and not swap) else torch.empty((B, rows, heads, D), dtype=q.dtype, device=q.device)
        lse = torch.empty((B, heads, rows), dtype=torch.float32, device=q.device)
        alibi, alibi_bs = _alibi_args(alibi_slopes_, B, H)
        a = _cabi.FaFwdParams()
        a.q, a.k, a.v, a.o, a.softmax_lse = _ptr(qk), _ptr(kcache), _ptr(vcache), _ptr(out), _ptr(lse)
        a.q_batch_stride, a.q_row_stride, a.q_head_stride = qk.stride(0), qk.stride(1), qk.stride(2)
        a.k_batch_stride, a.k_row_stride, a.k_head_stride = kcache.stride(0), kcache.stride(1), kcache.stride(2)
        a.v_batch_stride, a.v_row_stride, a.v_head_stride = vcache.stride(0), vcache.stride(1), vcache.stride(2)
        a.o_batch_stride, a.o_row_stride, a.o_head_stride = out.stride(0), out.stride(1), out.stride(2)
        a.seqused_k, a.seqused_k_add = _ptr(seqlens_k_), s_new
        a.cache_batch_idx, a.block_table, a.leftpad_k = _ptr(cache_batch_idx_), _ptr(block_table_), _ptr(leftpad_k_)
        a.block_table_batch_stride, a.page_block_size = (block_table_.stride(0) if paged else 0), page
        a.alibi_slopes, a.alibi_batch_stride = _ptr(alibi), alibi_bs
        a.b, a.h, a.h_k, a.d = B, heads, Hk, D
        a.seqlen_q, a.seqlen_k, a.total_q = rows, Sk, B * rows
        a.dtype = _dtype_code(q)
        a.is_causal, a.window_left, a.window_right = int(bool(is_causal)), int(window_size_left), int(window_size_right)
        a.softmax_scale, a.softcap = float(softmax_scale), float(softcap)
        a.num_splits = int(num_splits)
        ws_bytes = lib.fa_fwd_workspace_bytes(C.byref(a))  # split-KV partials (0 when the keys are not split)
        if ws_bytes > 0:
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=q.device)
            a.workspace, a.workspace_bytes = _ptr(ws), ws_bytes
        _cabi.check(lib.fa_fwd_kvcache(C.byref(a), C.c_void_p(_stream_ptr(q.device))))
    if swap:
        o2 = out.transpose(1, 2).reshape(B, 1, H, D)
        if out_ is not None:
            out_.copy_(o2)
            o2 = out_
        out, lse = o2, lse.reshape(B, H, 1)
    return [out, lse]
