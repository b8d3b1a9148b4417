"""Producer side of the varlen path: padded batch <-> packed tokens + int32 ``cu_seqlens``.

Same names, return tuples and semantics as the reference's ``flash_attn/bert_padding.py`` (:98-128
``unpad_input``, :204-218 ``pad_input``, :8-64 the two index autograd functions), written with plain
``index_select`` / ``index_copy`` (autograd handles both), so the varlen kernels can be fed without the
reference package.  ``cu_seqlens`` is the int32 prefix sum with a leading 0 -- the exact layout
``fa_varlen_fwd`` indexes with (reference block_info.h:17-36).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

__all__ = ["index_first_axis", "index_put_first_axis", "unpad_input", "pad_input"]


def index_first_axis(x: torch.Tensor, indices: torch.Tensor) -> torch.Tensor:
    """rows ``indices`` of ``x`` along dim 0 (gradient scatters back into zeros)."""
    return x.index_select(0, indices)


def index_put_first_axis(values: torch.Tensor, indices: torch.Tensor, first_axis_dim: int) -> torch.Tensor:
    """zeros of ``first_axis_dim`` rows with ``values`` written at ``indices`` (gradient gathers)."""
    out = torch.zeros((first_axis_dim,) + tuple(values.shape[1:]), device=values.device, dtype=values.dtype)
    return out.index_copy(0, indices, values)


def unpad_input(hidden_states: torch.Tensor, attention_mask: torch.Tensor, unused_mask: torch.Tensor | None = None):
    """(batch, seqlen, ...) + (batch, seqlen) mask (1 = valid) ->
    (packed (total, ...), indices (total,), cu_seqlens int32 (batch+1,), max_seqlen_in_batch: int, seqused int32 (batch,)).

    ``unused_mask`` marks slots that are allocated (kept in the packed tensor) but unused (not counted in ``seqused``).
    """
    all_masks = attention_mask if unused_mask is None else (attention_mask + unused_mask)
    seqlens = all_masks.sum(dim=-1, dtype=torch.int32)
    seqused = attention_mask.sum(dim=-1, dtype=torch.int32)
    indices = torch.nonzero(all_masks.flatten(), as_tuple=False).flatten()
    max_seqlen = int(seqlens.max().item()) if seqlens.numel() else 0
    cu_seqlens = F.pad(torch.cumsum(seqlens, dim=0, dtype=torch.int32), (1, 0))
    flat = hidden_states.reshape((-1,) + tuple(hidden_states.shape[2:]))
    return index_first_axis(flat, indices), indices, cu_seqlens, max_seqlen, seqused


def pad_input(hidden_states: torch.Tensor, indices: torch.Tensor, batch: int, seqlen: int) -> torch.Tensor:
    """(total, ...) -> (batch, seqlen, ...) with zeros at the padding positions."""
    out = index_put_first_axis(hidden_states, indices, batch * seqlen)
    return out.reshape((batch, seqlen) + tuple(hidden_states.shape[1:]))
