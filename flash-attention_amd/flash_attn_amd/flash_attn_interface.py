"""Host-side mirror of the reference's public attention API for the MI355X backend.

Same names, argument meaning and error behaviour as ``flash_attn/flash_attn_interface.py``
(reference :1019-1482): ``flash_attn_func``, ``flash_attn_varlen_func``, the qkv-/kv-packed
variants, and the raw ``_flash_attn_forward`` / ``_flash_attn_backward`` (+ varlen) wrappers.
All compute goes to the gfx950 library through the backend module (``flash_attn_2_cuda`` torch
extension when built, else the ctypes binder) -- there is no PyTorch fallback.

A user of the reference can also keep importing ``flash_attn`` itself: with
``flash-attention_amd/`` on ``sys.path`` the reference package picks up our ``flash_attn_2_cuda``
unmodified.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

try:  # the C++ torch extension is the primary binder
    import flash_attn_2_cuda as flash_attn_gpu  # type: ignore
except ImportError:  # same C ABI through ctypes (raises if libfa_gfx950.so is missing)
    from . import backend as flash_attn_gpu  # type: ignore

__all__ = [
    "flash_attn_func", "flash_attn_varlen_func", "flash_attn_qkvpacked_func", "flash_attn_kvpacked_func",
    "flash_attn_varlen_qkvpacked_func", "flash_attn_varlen_kvpacked_func", "flash_attn_with_kvcache", "flash_attn_padded_func",
    "_flash_attn_forward", "_flash_attn_backward", "_flash_attn_varlen_forward", "_flash_attn_varlen_backward",
]


def _unit_stride_last(x):
    return x.contiguous() if x is not None and x.stride(-1) != 1 else x


def _pad_head_dim(*ts):
    """Head dims that are not a multiple of 8 are zero-padded (reference :851-854)."""
    d = ts[0].shape[-1]
    if d % 8 == 0:
        return ts
    pad = 8 - d % 8
    return tuple(torch.nn.functional.pad(t, [0, pad]) for t in ts)


# The four raw wrappers are registered as custom ops (namespace ``flash_attn_amd``) with fake implementations, like the
# reference's ``flash_attn::_flash_attn_forward`` etc. (reference :84-144, :153-243, :252-338, :347-452), so that
# torch.compile / export trace through them without graph breaks; the public functions below call the registered ops.
def _fwd_impl(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, dropout_p: float, softmax_scale: float, causal: bool,
              window_size_left: int, window_size_right: int, softcap: float, alibi_slopes: Optional[torch.Tensor],
              return_softmax: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    q, k, v = (_unit_stride_last(t) for t in (q, k, v))
    out, lse, s_dmask, rng_state = flash_attn_gpu.fwd(
        q, k, v, None, alibi_slopes, dropout_p, softmax_scale, causal, window_size_left, window_size_right,
        softcap, return_softmax, None)
    return out, lse, s_dmask, rng_state


def _fwd_fake(q, k, v, dropout_p, softmax_scale, causal, window_size_left, window_size_right, softcap, alibi_slopes,
              return_softmax):
    B, Sq, H, _ = q.shape
    out = torch.empty_like(q)
    lse = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)
    p = (torch.empty((B, H, Sq, k.shape[1]), dtype=torch.uint8, device=q.device) if return_softmax
         else torch.empty((0,), dtype=q.dtype, device=q.device))
    return out, lse, p, torch.empty((2,), dtype=torch.int64, device=q.device)


def _varlen_fwd_impl(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, cu_seqlens_q: torch.Tensor, cu_seqlens_k: torch.Tensor,
                     max_seqlen_q: int, max_seqlen_k: int, dropout_p: float, softmax_scale: float, causal: bool,
                     window_size_left: int = -1, window_size_right: int = -1, softcap: float = 0.0,
                     alibi_slopes: Optional[torch.Tensor] = None, return_softmax: bool = False,
                     block_table: Optional[torch.Tensor] = None, leftpad_k: Optional[torch.Tensor] = None,
                     seqused_k: Optional[torch.Tensor] = None, zero_tensors: bool = False
                     ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    q, k, v = (_unit_stride_last(t) for t in (q, k, v))
    out, lse, s_dmask, rng_state = flash_attn_gpu.varlen_fwd(
        q, k, v, None, cu_seqlens_q, cu_seqlens_k, seqused_k, leftpad_k, block_table, alibi_slopes, max_seqlen_q,
        max_seqlen_k, dropout_p, softmax_scale, zero_tensors, causal, window_size_left, window_size_right, softcap,
        return_softmax, None)
    return out, lse, s_dmask, rng_state


def _varlen_fwd_fake(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p, softmax_scale, causal,
                     window_size_left=-1, window_size_right=-1, softcap=0.0, alibi_slopes=None, return_softmax=False,
                     block_table=None, leftpad_k=None, seqused_k=None, zero_tensors=False):
    total_q, H, _ = q.shape
    out = torch.empty_like(q)
    lse = torch.empty((H, total_q), dtype=torch.float32, device=q.device)
    p = (torch.empty((H, total_q, max_seqlen_k), dtype=torch.uint8, device=q.device) if return_softmax
         else torch.empty((0,), dtype=q.dtype, device=q.device))
    return out, lse, p, torch.empty((2,), dtype=torch.int64, device=q.device)


def _bwd_impl(dout: torch.Tensor, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor,
              softmax_lse: torch.Tensor, dq: Optional[torch.Tensor], dk: Optional[torch.Tensor], dv: Optional[torch.Tensor],
              dropout_p: float, softmax_scale: float, causal: bool, window_size_left: int, window_size_right: int,
              softcap: float, alibi_slopes: Optional[torch.Tensor], deterministic: bool,
              rng_state: Optional[torch.Tensor] = None) -> torch.Tensor:
    dout, q, k, v, out = (_unit_stride_last(t) for t in (dout, q, k, v, out))
    dq, dk, dv, softmax_d = flash_attn_gpu.bwd(
        dout, q, k, v, out, softmax_lse, dq, dk, dv, alibi_slopes, dropout_p, softmax_scale, causal,
        window_size_left, window_size_right, softcap, deterministic, None, rng_state)
    return softmax_d


def _bwd_fake(dout, q, k, v, out, softmax_lse, dq, dk, dv, dropout_p, softmax_scale, causal, window_size_left,
              window_size_right, softcap, alibi_slopes, deterministic, rng_state=None):
    B, Sq, H, _ = q.shape
    return torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)  # the ROCm shape (reference :333-336)


def _varlen_bwd_impl(dout: torch.Tensor, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor,
                     softmax_lse: torch.Tensor, dq: Optional[torch.Tensor], dk: Optional[torch.Tensor],
                     dv: Optional[torch.Tensor], cu_seqlens_q: torch.Tensor, cu_seqlens_k: torch.Tensor, max_seqlen_q: int,
                     max_seqlen_k: int, dropout_p: float, softmax_scale: float, causal: bool, window_size_left: int,
                     window_size_right: int, softcap: float, alibi_slopes: Optional[torch.Tensor], deterministic: bool,
                     rng_state: Optional[torch.Tensor] = None, zero_tensors: bool = False) -> torch.Tensor:
    dout, q, k, v, out = (_unit_stride_last(t) for t in (dout, q, k, v, out))
    dq, dk, dv, softmax_d = flash_attn_gpu.varlen_bwd(
        dout, q, k, v, out, softmax_lse, dq, dk, dv, cu_seqlens_q, cu_seqlens_k, alibi_slopes, max_seqlen_q,
        max_seqlen_k, dropout_p, softmax_scale, zero_tensors, causal, window_size_left, window_size_right, softcap,
        deterministic, None, rng_state)
    return softmax_d


def _varlen_bwd_fake(dout, q, k, v, out, softmax_lse, dq, dk, dv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                     dropout_p, softmax_scale, causal, window_size_left, window_size_right, softcap, alibi_slopes,
                     deterministic, rng_state=None, zero_tensors=False):
    total_q, H, _ = q.shape
    return torch.empty((H, total_q), dtype=torch.float32, device=q.device)


# The fused unpad -> attention -> pad path (flash_attn_padded_func) as custom ops too: the varlen kernels with C ABI v6's seqused_q / seqused_k next to
# cu_seqlens.  The extension module keeps the reference's positional signatures, so these two go through the ctypes binder of the same C ABI.
def _padded_fwd_impl(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, cu_seqlens_q: torch.Tensor, seqused_q: torch.Tensor,
                     cu_seqlens_k: torch.Tensor, seqused_k: torch.Tensor, max_seqlen_q: int, max_seqlen_k: int, dropout_p: float,
                     softmax_scale: float, causal: bool, window_size_left: int, window_size_right: int, softcap: float,
                     alibi_slopes: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    from . import backend as be
    q, k, v = (_unit_stride_last(t) for t in (q, k, v))
    out, lse, _, rng_state = be.varlen_fwd(q, k, v, None, cu_seqlens_q, cu_seqlens_k, seqused_k, None, None, alibi_slopes, max_seqlen_q,
                                           max_seqlen_k, dropout_p, softmax_scale, True, causal, window_size_left, window_size_right,
                                           softcap, False, None, 0, seqused_q=seqused_q)
    return out, lse, rng_state


def _padded_fwd_fake(q, k, v, cu_seqlens_q, seqused_q, cu_seqlens_k, seqused_k, max_seqlen_q, max_seqlen_k, dropout_p, softmax_scale,
                     causal, window_size_left, window_size_right, softcap, alibi_slopes):
    total_q, H, _ = q.shape
    # (contiguous like the real op's freshly allocated output, whatever the strides of q: q may be a view of a packed qkv)
    return (torch.empty(q.shape, dtype=q.dtype, device=q.device), torch.empty((H, total_q), dtype=torch.float32, device=q.device),
            torch.empty((2,), dtype=torch.int64, device=q.device))


def _padded_bwd_impl(dout: torch.Tensor, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, softmax_lse: torch.Tensor,
                     dq: torch.Tensor, dk: torch.Tensor, dv: torch.Tensor, cu_seqlens_q: torch.Tensor, seqused_q: torch.Tensor,
                     cu_seqlens_k: torch.Tensor, seqused_k: torch.Tensor, max_seqlen_q: int, max_seqlen_k: int, dropout_p: float,
                     softmax_scale: float, causal: bool, window_size_left: int, window_size_right: int, softcap: float,
                     alibi_slopes: Optional[torch.Tensor], deterministic: bool, rng_state: Optional[torch.Tensor] = None) -> torch.Tensor:
    from . import backend as be
    dout, q, k, v, out = (_unit_stride_last(t) for t in (dout, q, k, v, out))
    _, _, _, softmax_d = be.varlen_bwd(dout, q, k, v, out, softmax_lse, dq, dk, dv, cu_seqlens_q, cu_seqlens_k, alibi_slopes, max_seqlen_q,
                                       max_seqlen_k, dropout_p, softmax_scale, True, causal, window_size_left, window_size_right, softcap,
                                       deterministic, None, rng_state, seqused_q=seqused_q, seqused_k=seqused_k)
    return softmax_d


def _padded_bwd_fake(dout, q, k, v, out, softmax_lse, dq, dk, dv, cu_seqlens_q, seqused_q, cu_seqlens_k, seqused_k, max_seqlen_q,
                     max_seqlen_k, dropout_p, softmax_scale, causal, window_size_left, window_size_right, softcap, alibi_slopes,
                     deterministic, rng_state=None):
    total_q, H, _ = q.shape
    return torch.empty((H, total_q), dtype=torch.float32, device=q.device)


def _register(name, impl, fake, mutates=()):
    op = torch.library.custom_op(f"flash_attn_amd::{name}", impl, mutates_args=mutates, device_types="cuda")
    op.register_fake(fake)
    return getattr(torch.ops.flash_attn_amd, name)


_flash_attn_forward = _register("_flash_attn_forward", _fwd_impl, _fwd_fake)
_flash_attn_varlen_forward = _register("_flash_attn_varlen_forward", _varlen_fwd_impl, _varlen_fwd_fake)
_flash_attn_backward = _register("_flash_attn_backward", _bwd_impl, _bwd_fake, ("dq", "dk", "dv"))
_flash_attn_padded_forward = _register("_flash_attn_padded_forward", _padded_fwd_impl, _padded_fwd_fake)
_flash_attn_padded_backward = _register("_flash_attn_padded_backward", _padded_bwd_impl, _padded_bwd_fake, ("dq", "dk", "dv"))
_flash_attn_varlen_backward = _register("_flash_attn_varlen_backward", _varlen_bwd_impl, _varlen_bwd_fake, ("dq", "dk", "dv"))


# An exported program (torch.export) holds the raw ops, not the autograd.Function wrappers below; with an autograd formula on
# the two forward ops such a graph can still be differentiated (the reference registers one for its FA3 ops, the precedent
# hopper/test_torch_compile_and_export.py exercises; its FA2 ops have none).  Inside the Function wrappers the ops run
# without grad mode, so this never double-counts.
def _fwd_setup(ctx, inputs, output):
    q, k, v, dropout_p, softmax_scale, causal, wl, wr, softcap, alibi_slopes, _ = inputs
    out, lse, _, rng_state = output
    ctx.save_for_backward(q, k, v, out, lse, rng_state, *([alibi_slopes] if alibi_slopes is not None else []))
    ctx.args = (dropout_p, softmax_scale, causal, wl, wr, softcap, alibi_slopes is not None)


def _fwd_backward(ctx, dout, dlse, dp, drng):
    q, k, v, out, lse, rng_state, *rest = ctx.saved_tensors
    dropout_p, softmax_scale, causal, wl, wr, softcap, has_alibi = ctx.args
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    _flash_attn_backward(dout.contiguous(), q, k, v, out, lse, dq, dk, dv, dropout_p, softmax_scale, causal, wl, wr, softcap,
                         rest[0] if has_alibi else None, False, rng_state)
    return dq, dk, dv, None, None, None, None, None, None, None, None


def _varlen_fwd_setup(ctx, inputs, output):
    (q, k, v, cu_q, cu_k, max_q, max_k, dropout_p, softmax_scale, causal, wl, wr, softcap, alibi_slopes, _, block_table,
     leftpad_k, seqused_k, zero_tensors) = inputs
    out, lse, _, rng_state = output
    # paged / left-padded / partially used K has no backward: say so when (and only if) a gradient is actually asked for -- the
    # forward itself (e.g. paged-KV inference without torch.no_grad()) must keep working
    ctx.no_backward = block_table is not None or leftpad_k is not None or seqused_k is not None
    if ctx.no_backward:
        return
    ctx.save_for_backward(q, k, v, out, lse, rng_state, cu_q, cu_k, *([alibi_slopes] if alibi_slopes is not None else []))
    ctx.args = (max_q, max_k, dropout_p, softmax_scale, causal, wl, wr, softcap, alibi_slopes is not None)


def _varlen_fwd_backward(ctx, dout, dlse, dp, drng):
    if ctx.no_backward:
        raise RuntimeError("flash_attn_amd::_flash_attn_varlen_forward: no backward with block_table / leftpad_k / seqused_k")
    q, k, v, out, lse, rng_state, cu_q, cu_k, *rest = ctx.saved_tensors
    max_q, max_k, dropout_p, softmax_scale, causal, wl, wr, softcap, has_alibi = ctx.args
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    _flash_attn_varlen_backward(dout.contiguous(), q, k, v, out, lse, dq, dk, dv, cu_q, cu_k, max_q, max_k, dropout_p, softmax_scale,
                                causal, wl, wr, softcap, rest[0] if has_alibi else None, False, rng_state)
    return (dq, dk, dv) + (None,) * 16


torch.library.register_autograd("flash_attn_amd::_flash_attn_forward", _fwd_backward, setup_context=_fwd_setup)
torch.library.register_autograd("flash_attn_amd::_flash_attn_varlen_forward", _varlen_fwd_backward, setup_context=_varlen_fwd_setup)


class _AttnFn(torch.autograd.Function):
    """Fixed-length attention with separate q, k, v (reference FlashAttnFunc :828-911)."""

    @staticmethod
    def forward(ctx, q, k, v, dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes, deterministic,
                return_softmax, is_grad_enabled):
        needs_grad = is_grad_enabled and any(t.requires_grad for t in (q, k, v))
        if softmax_scale is None:
            softmax_scale = q.shape[-1] ** (-0.5)
        d_orig = q.shape[-1]
        q, k, v = _pad_head_dim(q, k, v)
        out_p, lse, s_dmask, rng_state = _flash_attn_forward(
            q, k, v, dropout_p, softmax_scale, causal, window_size[0], window_size[1], softcap, alibi_slopes,
            return_softmax and dropout_p > 0)
        if needs_grad:
            ctx.save_for_backward(q, k, v, out_p, lse, rng_state)
            ctx.cfg = (dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes, deterministic, d_orig)
        out = out_p[..., :d_orig]
        return out if not return_softmax else (out, lse, s_dmask)

    @staticmethod
    def backward(ctx, dout, *unused):
        q, k, v, out, lse, rng_state = ctx.saved_tensors
        dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes, deterministic, d_orig = ctx.cfg
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        (dout_p,) = _pad_head_dim(dout) if dout.shape[-1] % 8 else (dout,)
        _flash_attn_backward(dout_p, q, k, v, out, lse, dq, dk, dv, dropout_p, softmax_scale, causal, window_size[0],
                             window_size[1], softcap, alibi_slopes, deterministic, rng_state)
        return (dq[..., :d_orig], dk[..., :d_orig], dv[..., :d_orig]) + (None,) * 9


class _VarlenAttnFn(torch.autograd.Function):
    """Packed variable-length attention (reference FlashAttnVarlenFunc :914-1016)."""

    @staticmethod
    def forward(ctx, q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p, softmax_scale, causal,
                window_size, softcap, alibi_slopes, deterministic, return_softmax, block_table, is_grad_enabled):
        needs_grad = is_grad_enabled and any(t.requires_grad for t in (q, k, v))
        if softmax_scale is None:
            softmax_scale = q.shape[-1] ** (-0.5)
        d_orig = q.shape[-1]
        q, k, v = _pad_head_dim(q, k, v)
        out_p, lse, s_dmask, rng_state = _flash_attn_varlen_forward(
            q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p, softmax_scale, causal,
            window_size[0], window_size[1], softcap, alibi_slopes, return_softmax and dropout_p > 0, block_table)
        if needs_grad:
            ctx.save_for_backward(q, k, v, out_p, lse, cu_seqlens_q, cu_seqlens_k, rng_state)
            ctx.cfg = (max_seqlen_q, max_seqlen_k, dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes,
                       deterministic, d_orig)
        out = out_p[..., :d_orig]
        return out if not return_softmax else (out, lse, s_dmask)

    @staticmethod
    def backward(ctx, dout, *unused):
        q, k, v, out, lse, cu_q, cu_k, rng_state = ctx.saved_tensors
        mq, mk, dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes, deterministic, d_orig = ctx.cfg
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        (dout_p,) = _pad_head_dim(dout) if dout.shape[-1] % 8 else (dout,)
        _flash_attn_varlen_backward(dout_p, q, k, v, out, lse, dq, dk, dv, cu_q, cu_k, mq, mk, dropout_p, softmax_scale,
                                    causal, window_size[0], window_size[1], softcap, alibi_slopes, deterministic, rng_state)
        return (dq[..., :d_orig], dk[..., :d_orig], dv[..., :d_orig]) + (None,) * 14


class _PaddedAttnFn(torch.autograd.Function):
    """Attention over a PADDED batch, in place: the reference's unpad_input -> flash_attn_varlen_func -> pad_input chain
    (flash_attn/bert_padding.py:98-128, 204-218; e.g. flash_attn/modules/mha.py) without its gather and scatter passes.  The varlen
    kernels address entry b at row b*S + start[b] of the flattened (B*S, H, D) tensors and stop after seqlens[b] rows (C ABI v6:
    seqused_q / seqused_k next to cu_seqlens); padded rows are never read, their outputs and gradients are zero."""

    @staticmethod
    def forward(ctx, q, k, v, cu_q, len_q, cu_k, len_k, dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes,
                deterministic):
        B, Sq, H, D = q.shape
        Sk = k.shape[1]
        needs_grad = any(t.requires_grad for t in (q, k, v))   # (before the head-dim padding: its outputs carry no grad flag in here)
        if softmax_scale is None:
            softmax_scale = D ** (-0.5)
        q, k, v = _pad_head_dim(q, k, v)
        qf, kf, vf = (_unit_stride_last(t).reshape(-1, t.shape[2], t.shape[3]) for t in (q, k, v))
        out, lse, rng_state = _flash_attn_padded_forward(qf, kf, vf, cu_q, len_q, cu_k, len_k, Sq, Sk, dropout_p, softmax_scale, causal,
                                                         window_size[0], window_size[1], softcap, alibi_slopes)
        if needs_grad:
            ctx.save_for_backward(qf, kf, vf, out, lse, cu_q, len_q, cu_k, len_k, rng_state)
            ctx.cfg = (Sq, Sk, dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes, deterministic, D, q.shape, k.shape)
        return out.reshape(B, Sq, H, -1)[..., :D]

    @staticmethod
    def backward(ctx, dout):
        qf, kf, vf, out, lse, cu_q, len_q, cu_k, len_k, rng_state = ctx.saved_tensors
        Sq, Sk, dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes, deterministic, d_orig, q_shape, k_shape = ctx.cfg
        (dout_p,) = _pad_head_dim(dout) if dout.shape[-1] % 8 else (dout,)
        dof = dout_p.contiguous().reshape(-1, dout_p.shape[2], dout_p.shape[3])
        dq, dk, dv = (torch.empty(t.shape, dtype=t.dtype, device=t.device) for t in (qf, kf, vf))   # contiguous whatever qf's strides (zero_tensors: the launcher clears them first)
        _flash_attn_padded_backward(dof, qf, kf, vf, out, lse, dq, dk, dv, cu_q, len_q, cu_k, len_k, Sq, Sk, dropout_p, softmax_scale, causal,
                                    window_size[0], window_size[1], softcap, alibi_slopes, deterministic, rng_state)
        return (dq.reshape(q_shape)[..., :d_orig], dk.reshape(k_shape)[..., :d_orig], dv.reshape(k_shape)[..., :d_orig]) + (None,) * 11


def flash_attn_padded_func(q, k, v, seqlens_q, seqlens_k=None, starts_q=None, starts_k=None, dropout_p=0.0, softmax_scale=None,
                           causal=False, window_size=(-1, -1), softcap=0.0, alibi_slopes=None, deterministic=False):
    """Fused unpad -> attention -> pad.  q (B,Sq,H,D); k, v (B,Sk,Hk,D), PADDED: the tokens of entry b are rows
    starts[b] .. starts[b] + seqlens[b] - 1 (int32 device tensors of shape (B,); starts default to 0 = right padding, seqlens_k /
    starts_k default to the query's: self-attention).  ``bert_padding.padded_batch_args(attention_mask)`` derives them from a
    left- or right-padded mask without a host synchronisation.  Masks (causal, window) are aligned per entry exactly as
    flash_attn_varlen_func aligns them on the unpadded sequences.  Returns out (B,Sq,H,D) with zeros in the padded rows -- the result
    of pad_input(flash_attn_varlen_func(unpad_input(..))) without the three gather / scatter passes and the unpadded copies."""
    B, Sq = q.shape[0], q.shape[1]
    Sk = k.shape[1]

    if B == 0 or Sq == 0 or Sk == 0:   # nothing to attend: the padded output (and, through autograd, the gradients) are zeros
        if torch.is_grad_enabled() and any(t.requires_grad for t in (q, k, v)):
            # zeros attached to the graph of all three (zero gradients); built from EMPTY slices, so no NaN / Inf of the inputs can leak into them
            return torch.zeros_like(q) + (q[:0].sum() + k[:0].sum() + v[:0].sum())
        return torch.zeros_like(q)

    def _args(S, lens, starts):
        # clamped on the device, no host synchronisation: a run may not leave its entry (starts[b] + seqlens[b] <= S), or the kernels would read the
        # next entry's rows and, for the last entry, read and write past the end of the tensors
        st = torch.zeros((B,), dtype=torch.int32, device=q.device) if starts is None else starts.to(torch.int32).clamp(0, S)
        lens = torch.minimum(lens.to(torch.int32).clamp_min(0), S - st).contiguous()
        cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device=q.device)
        cu[:B] += st
        return cu, lens

    if seqlens_k is None:
        if Sk != Sq:
            raise ValueError("seqlens_k is required when k is padded to a different length than q")
        seqlens_k, starts_k = seqlens_q, (starts_q if starts_k is None else starts_k)
    cu_q, len_q = _args(Sq, seqlens_q, starts_q)
    cu_k, len_k = _args(Sk, seqlens_k, starts_k)
    return _PaddedAttnFn.apply(q, k, v, cu_q, len_q, cu_k, len_k, dropout_p, softmax_scale, causal, tuple(window_size), softcap,
                               alibi_slopes, deterministic)


def flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                    alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """q (B,Sq,H,D); k, v (B,Sk,Hk,D) with H % Hk == 0 (MQA/GQA).  Causal / window masks are aligned to the
    bottom-right corner; ``window_size=(l, r)`` lets query i see keys [i+Sk-Sq-l, i+Sk-Sq+r].
    Returns out (B,Sq,H,D) (and softmax_lse (B,H,Sq), S_dmask when ``return_attn_probs``)."""
    return _AttnFn.apply(q, k, v, dropout_p, softmax_scale, causal, tuple(window_size), softcap, alibi_slopes,
                         deterministic, return_attn_probs, torch.is_grad_enabled())


class _PackedAttnFn(torch.autograd.Function):
    """The four packed layouts -- qkv (B,S,3,H,D) / (total,3,H,D) and q + kv (B,Sk,2,Hk,D) / (total_k,2,Hk,D) -- behind one
    Function (reference FlashAttnQKVPackedFunc :461-540, FlashAttnVarlenQKVPackedFunc :543-634, FlashAttnKVPackedFunc :637-722,
    FlashAttnVarlenKVPackedFunc :724-825).  The packed axis is dim -3 in every layout, so `unbind(-3)` yields the strided q / k / v
    views the kernels read in place, and the backward hands the kernels the same views of ONE freshly allocated packed gradient
    (dq_ / dk_ / dv_ of the backend's bwd): no slice-backward temporaries, no zero fill, no adds."""

    @staticmethod
    def forward(ctx, q, packed, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p, softmax_scale, causal,
                window_size, softcap, alibi_slopes, deterministic, return_softmax, is_grad_enabled):
        needs_grad = is_grad_enabled and (packed.requires_grad or (q is not None and q.requires_grad))
        parts = packed.unbind(dim=-3)
        q_, k_, v_ = parts if q is None else (q, parts[0], parts[1])
        if softmax_scale is None:
            softmax_scale = q_.shape[-1] ** (-0.5)
        d_orig = q_.shape[-1]
        q_, k_, v_ = _pad_head_dim(q_, k_, v_)
        if cu_seqlens_q is None:
            out_p, lse, s_dmask, rng_state = _flash_attn_forward(
                q_, k_, v_, dropout_p, softmax_scale, causal, window_size[0], window_size[1], softcap, alibi_slopes,
                return_softmax and dropout_p > 0)
        else:
            out_p, lse, s_dmask, rng_state = _flash_attn_varlen_forward(
                q_, k_, v_, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p, softmax_scale, causal,
                window_size[0], window_size[1], softcap, alibi_slopes, return_softmax and dropout_p > 0, None)
        if needs_grad:
            ctx.save_for_backward(q_, k_, v_, out_p, lse, rng_state, *([] if cu_seqlens_q is None else [cu_seqlens_q, cu_seqlens_k]))
            ctx.cfg = (q is None, max_seqlen_q, max_seqlen_k, dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes,
                       deterministic, d_orig)
        out = out_p[..., :d_orig]
        return out if not return_softmax else (out, lse, s_dmask)

    @staticmethod
    def backward(ctx, dout, *unused):
        q, k, v, out, lse, rng_state, *cu = ctx.saved_tensors
        qkv_packed, mq, mk, dropout_p, softmax_scale, causal, window_size, softcap, alibi_slopes, deterministic, d_orig = ctx.cfg
        # one packed gradient (head dim as the kernels saw it, i.e. padded to a multiple of 8), its slices are the kernels' outputs
        n = 3 if qkv_packed else 2
        dpacked = torch.empty(k.shape[:-2] + (n,) + k.shape[-2:], dtype=k.dtype, device=k.device)
        dparts = dpacked.unbind(dim=-3)
        dq, dk, dv = dparts if qkv_packed else (torch.empty_like(q), dparts[0], dparts[1])
        (dout_p,) = _pad_head_dim(dout) if dout.shape[-1] % 8 else (dout,)
        if not cu:
            _flash_attn_backward(dout_p, q, k, v, out, lse, dq, dk, dv, dropout_p, softmax_scale, causal, window_size[0],
                                 window_size[1], softcap, alibi_slopes, deterministic, rng_state)
        else:
            _flash_attn_varlen_backward(dout_p, q, k, v, out, lse, dq, dk, dv, cu[0], cu[1], mq, mk, dropout_p, softmax_scale,
                                        causal, window_size[0], window_size[1], softcap, alibi_slopes, deterministic, rng_state)
        return (None if qkv_packed else dq[..., :d_orig], dpacked[..., :d_orig]) + (None,) * 13


def flash_attn_kvpacked_func(q, kv, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                             alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """q (B,Sq,H,D), kv (B,Sk,2,Hk,D): the k / v slices are read in place; the backward writes one packed dkv."""
    return _PackedAttnFn.apply(q, kv, None, None, 0, 0, dropout_p, softmax_scale, causal, tuple(window_size), softcap,
                               alibi_slopes, deterministic, return_attn_probs, torch.is_grad_enabled())


def flash_attn_qkvpacked_func(qkv, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                              alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """qkv (B,S,3,H,D): read in place; the backward writes one packed dqkv (no concatenation of separate gradients)."""
    return _PackedAttnFn.apply(None, qkv, None, None, 0, 0, dropout_p, softmax_scale, causal, tuple(window_size), softcap,
                               alibi_slopes, deterministic, return_attn_probs, torch.is_grad_enabled())


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=0.0,
                           softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0, alibi_slopes=None,
                           deterministic=False, return_attn_probs=False, block_table=None):
    """q (total_q,H,D); k, v (total_k,Hk,D); cu_seqlens_* int32 (B+1) cumulative lengths on the device.
    Returns out (total_q,H,D) (and softmax_lse (H,total_q) when ``return_attn_probs``)."""
    return _VarlenAttnFn.apply(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p, softmax_scale,
                               causal, tuple(window_size), softcap, alibi_slopes, deterministic, return_attn_probs,
                               block_table, torch.is_grad_enabled())


def flash_attn_varlen_kvpacked_func(q, kv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=0.0,
                                    softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                                    alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """q (total_q,H,D), kv (total_k,2,Hk,D)."""
    return _PackedAttnFn.apply(q, kv, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p, softmax_scale, causal,
                               tuple(window_size), softcap, alibi_slopes, deterministic, return_attn_probs, torch.is_grad_enabled())


def flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None, causal=False,
                                     window_size=(-1, -1), softcap=0.0, alibi_slopes=None, deterministic=False,
                                     return_attn_probs=False):
    """qkv (total,3,H,D)."""
    return _PackedAttnFn.apply(None, qkv, cu_seqlens, cu_seqlens, max_seqlen, max_seqlen, dropout_p, softmax_scale, causal,
                               tuple(window_size), softcap, alibi_slopes, deterministic, return_attn_probs, torch.is_grad_enabled())


def flash_attn_with_kvcache(q, k_cache, v_cache, k=None, v=None, rotary_cos=None, rotary_sin=None, cache_seqlens=None,
                            cache_batch_idx=None, cache_leftpad=None, block_table=None, softmax_scale=None, causal=False,
                            window_size=(-1, -1), softcap=0.0, rotary_interleaved=True, alibi_slopes=None, num_splits=0,
                            return_softmax_lse=False):
    """Inference attention against a KV cache (reference :1485-1627).  If k / v are given they are written into the
    cache in place at rows ``cache_seqlens[b] ..`` before attending; ``cache_seqlens`` may be an int or an int32
    tensor (batch,); ``cache_batch_idx`` selects cache rows; ``block_table`` selects pages of a paged cache
    (num_blocks, page, Hk, D); ``rotary_cos/sin`` (seqlen_ro, rotary_dim/2) rotate the new keys at positions
    cache_seqlens + i and the queries likewise (all at cache_seqlens unless causal / local); ``cache_leftpad`` gives the
    first valid cache row of each entry.  No backward."""
    q, k, v = (_unit_stride_last(t) for t in (q, k, v))
    if softmax_scale is None:
        softmax_scale = q.shape[-1] ** (-0.5)
    if cache_seqlens is not None and isinstance(cache_seqlens, int):
        cache_seqlens = torch.full((q.shape[0],), cache_seqlens, dtype=torch.int32, device=k_cache.device)
    out, lse = flash_attn_gpu.fwd_kvcache(
        q, k_cache, v_cache, k, v, cache_seqlens, _unit_stride_last(rotary_cos), _unit_stride_last(rotary_sin),
        None if cache_batch_idx is None else cache_batch_idx.contiguous(), cache_leftpad,
        None if block_table is None else _unit_stride_last(block_table), alibi_slopes, None, softmax_scale, causal,
        window_size[0], window_size[1], softcap, rotary_interleaved, num_splits)
    return (out, lse) if return_softmax_lse else out
