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

__all__ = ["index_first_axis", "index_put_first_axis", "unpad_input", "unpad_input_for_concatenated_sequences", "pad_input",
           "padded_batch_args", "padded_batch_is_contiguous"]


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


def unpad_input_for_concatenated_sequences(hidden_states: torch.Tensor, attention_mask_in_length: torch.Tensor):
    """Several short samples concatenated inside one row of the padded batch (reference :131-201).

    ``attention_mask_in_length`` (batch, seqlen) int: row b lists the lengths of the samples packed into row b, left-aligned and
    zero-filled -- e.g. ``[2, 3, 0, 0, 0, 0]`` = a 2-token and a 3-token sample occupying the first 5 slots of a 6-slot row.  Every
    listed sample becomes its own sequence of the varlen batch, so attention never crosses a sample boundary.
    Returns (packed (total, ...), indices (total,), cu_seqlens int32 (n_samples+1,), max_seqlen_in_batch: int).
    """
    seqlen = attention_mask_in_length.shape[-1]
    used = attention_mask_in_length.sum(dim=-1)                                         # occupied slots per row
    slot = torch.arange(seqlen, device=used.device, dtype=used.dtype)
    indices = torch.nonzero((slot[None, :] < used[:, None]).flatten(), as_tuple=False).flatten()
    lengths = attention_mask_in_length.flatten()
    lengths = lengths[torch.nonzero(lengths, as_tuple=False).flatten()]                 # the samples, in (row, position) order
    max_seqlen = int(lengths.max().item()) if lengths.numel() else 0
    cu_seqlens = F.pad(torch.cumsum(lengths, dim=0, dtype=torch.int32), (1, 0))
    flat = hidden_states.reshape((-1,) + tuple(hidden_states.shape[2:]))
    return index_first_axis(flat, indices), indices, cu_seqlens, max_seqlen


def pad_input(hidden_states: torch.Tensor, indices: torch.Tensor, batch: int, seqlen: int) -> torch.Tensor:
    """(total, ...) -> (batch, seqlen, ...) with zeros at the padding positions."""
    out = index_put_first_axis(hidden_states, indices, batch * seqlen)
    return out.reshape((batch, seqlen) + tuple(hidden_states.shape[1:]))


def padded_batch_args(attention_mask: torch.Tensor):
    """(batch, seqlen) mask (1 = valid) of a LEFT- or RIGHT-padded batch -> (seqlens int32 (batch,), starts int32 (batch,)): entry b's tokens
    are the rows starts[b] .. starts[b] + seqlens[b] - 1.  The arguments of ``flash_attn_padded_func`` (attention over the padded tensors in
    place, no unpad_input / pad_input passes); computed on the device without a host synchronisation.  The valid tokens of an entry must be
    one contiguous run -- what left or right padding produces; a mask with holes needs ``unpad_input`` (``padded_batch_is_contiguous``
    checks, at the price of a synchronisation)."""
    m = attention_mask.to(torch.bool)
    seqlens = m.sum(dim=-1, dtype=torch.int32)
    starts = m.to(torch.int8).argmax(dim=-1).to(torch.int32)   # index of the first valid token (0 for an empty entry)
    return seqlens, starts


def padded_batch_is_contiguous(attention_mask: torch.Tensor) -> bool:
    """True iff every entry's valid tokens form one contiguous run (synchronises)."""
    m = attention_mask.to(torch.bool)
    seqlens, starts = padded_batch_args(m)
    pos = torch.arange(m.shape[1], device=m.device, dtype=torch.int32)[None, :]
    run = (pos >= starts[:, None]) & (pos < (starts + seqlens)[:, None])
    return bool((run == m).all().item())
