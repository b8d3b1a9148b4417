"""CPU oracle for the fused-attention hot path (TEST INFRASTRUCTURE ONLY).

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The shipped path (``flash-attention_amd/``) never imports anything from
``oracle/`` and fails loudly when the HIP library is missing.

It restates, in float64 numpy, the arithmetic the reference defines for
``flash_attn_func`` / ``flash_attn_varlen_func`` / ``_flash_attn_backward``:

* visibility predicate (bottom-right aligned causal / sliding window):
  reference ``flash_attn/flash_attn_interface.py:1175-1189`` (documentation of
  the alignment), ``csrc/flash_attn/src/mask.h:172-203`` (kernel predicate),
  ``tests/test_util.py:150-182`` (``construct_local_mask``).
* host-side flag normalisation: ``csrc/flash_attn/flash_api.cpp:422-427`` and
  ``:155-162``.
* forward / LSE / fully-masked-row convention (out = 0, LSE = +inf):
  ``csrc/flash_attn/src/softmax.h:169-186``, ``flash_fwd_kernel.h:101-135``.
* GQA head mapping ``h_kv = h // (H / H_k)``: ``flash_fwd_kernel.h:155,161``.
* backward formulas (P from the saved LSE, D = rowsum(dO*O), dS = P*(dP-D),
  softmax_scale applied once at the end): ``flash_bwd_preprocess_kernel.h:40-48``,
  ``flash_bwd_kernel.h:536,584-595,733``.
* varlen addressing through ``cu_seqlens``: ``csrc/flash_attn/src/block_info.h:12-45``.
* softcap ``s -> softcap * tanh(s * scale / softcap)``: ``src/utils.h:395-409``,
  ALiBi bias ``-slope * |i + Sk - Sq - j|``: ``src/alibi.h``.

No ``oracle/_ref`` build: the reference's kernels are CUDA (``csrc/flash_attn`` needs nvcc + the un-vendored
``csrc/cutlass`` submodule) or ROCm-CK (``csrc/flash_attn_ck`` needs the empty ``csrc/composable_kernel`` submodule,
ROCm/composable_kernel ``amd-master``), so nothing on this path compiles from its own few source files here; the
reference's CPU-runnable Python oracle is what pins this file.

Parity pinning: ``tests/golden/*.npz`` were produced by importing the
reference's own ``tests/test_util.py::attention_ref`` (+ torch autograd for the
gradients) in the build container (``tests/golden/make_golden.py``); the
``-m "not gpu"`` suite checks this oracle against every one of them, and
against the documented 2x5 / 5x2 causal-mask pictures of
``flash_attn_interface.py:1176-1185``.  The dropout branch (explicit keep-mask, 1/(1-p) scaling) is pinned the same
way (``dropout_ref_cases.npz``: ``attention_ref(dropout_p, dropout_mask)`` + autograd).
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import numpy as np

__all__ = [
    "normalize_window",
    "visible_mask",
    "attention_fwd",
    "attention_bwd",
    "varlen_fwd",
    "varlen_bwd",
    "visible_keys_per_row",
    "attention_flops",
]


def normalize_window(seqlen_q: int, seqlen_k: int, causal: bool, wl: int, wr: int,
                     has_alibi: bool = False) -> Tuple[bool, int, int]:
    """Flag normalisation the host API performs before launching.

    Follows reference csrc/flash_attn/flash_api.cpp:422-427: a window at least
    as wide as the key sequence is "no window"; a single query row needs no
    causal mask (``Sq == 1`` and no alibi); causal forces ``wr = 0``.
    Returns (is_causal, wl, wr) with -1 meaning unbounded.
    """
    if wl >= seqlen_k:
        wl = -1
    if wr >= seqlen_k:
        wr = -1
    if seqlen_q == 1 and not has_alibi:
        causal = False
    if causal:
        wr = 0
    return causal, wl, wr


def visible_mask(seqlen_q: int, seqlen_k: int, wl: int, wr: int) -> np.ndarray:
    """bool (Sq, Sk): True where query row i may attend key column j.

    ``visible(i,j) = j < Sk and (wr < 0 or j <= i + (Sk-Sq) + wr)
                              and (wl < 0 or j >= i + (Sk-Sq) - wl)``
    (reference mask.h:172-203 / tests/test_util.py:150-182; causal == wr = 0).
    """
    i = np.arange(seqlen_q, dtype=np.int64)[:, None]
    j = np.arange(seqlen_k, dtype=np.int64)[None, :]
    shift = seqlen_k - seqlen_q
    vis = np.ones((seqlen_q, seqlen_k), dtype=bool)
    if wr >= 0:
        vis &= j <= i + shift + wr
    if wl >= 0:
        vis &= j >= i + shift - wl
    return vis


def visible_keys_per_row(seqlen_q: int, seqlen_k: int, causal: bool,
                         window: Tuple[int, int] = (-1, -1)) -> np.ndarray:
    """int64 (Sq,): number of visible keys of each query row (for FLOP counts)."""
    c, wl, wr = normalize_window(seqlen_q, seqlen_k, causal, window[0], window[1])
    i = np.arange(seqlen_q, dtype=np.int64)
    shift = seqlen_k - seqlen_q
    hi = np.full(seqlen_q, seqlen_k - 1, dtype=np.int64) if wr < 0 else np.minimum(seqlen_k - 1, i + shift + wr)
    lo = np.zeros(seqlen_q, dtype=np.int64) if wl < 0 else np.maximum(0, i + shift - wl)
    return np.maximum(0, hi - lo + 1)


def attention_flops(batch: int, nheads: int, seqlen_q: int, seqlen_k: int, headdim: int,
                    causal: bool = False, window: Tuple[int, int] = (-1, -1), mode: str = "fwd") -> float:
    """Algorithmic FLOPs, reference convention.

    ``fwd = 4 * B * H * D * sum_i visible_keys(i)`` (benchmarks/benchmark_flash_attention.py:27-30,
    flash_attn/cute/bench_utils.py:15-47); bwd = 2.5x fwd; fwd_bwd = 3.5x fwd.
    """
    vis = float(visible_keys_per_row(seqlen_q, seqlen_k, causal, window).sum())
    f = 4.0 * batch * nheads * headdim * vis
    return {"fwd": f, "bwd": 2.5 * f, "fwd_bwd": 3.5 * f}[mode]


def _f64(x) -> np.ndarray:
    if hasattr(x, "detach"):  # torch tensor
        x = x.detach().to("cpu").double().numpy()
    return np.asarray(x, dtype=np.float64)


def _scores(qh, kh, scale, softcap, vis, alibi_slope, shift):
    """Scaled, capped, biased, masked scores of one (batch, head): (Sq, Sk) f64."""
    s = (qh @ kh.T) * scale
    if softcap > 0.0:
        s = softcap * np.tanh(s / softcap)
    if alibi_slope is not None:
        i = np.arange(qh.shape[0], dtype=np.float64)[:, None]
        j = np.arange(kh.shape[0], dtype=np.float64)[None, :]
        s = s - alibi_slope * np.abs(i + shift - j)
    return np.where(vis, s, -np.inf)


def attention_fwd(q, k, v, softmax_scale: Optional[float] = None, causal: bool = False,
                  window: Tuple[int, int] = (-1, -1), softcap: float = 0.0,
                  alibi_slopes=None, dropout_p: float = 0.0, dropout_mask=None):
    """Forward oracle.

    Dropout (tests/test_util.py:262-269, flash_fwd_kernel.h:357-368): the normalised probabilities are
    multiplied by ``dropout_mask`` (B,H,Sq,Sk, True = keep) and by 1/(1-dropout_p) before the product
    with V; the softmax statistics (lse) are those of the un-dropped scores.

    q (B,Sq,H,D); k,v (B,Sk,Hk,D).  Returns out (B,Sq,H,D) f64 and
    lse (B,H,Sq) f64 with lse = log(sum_j exp(score_ij)) over visible j;
    rows with no visible key give out = 0, lse = +inf (softmax.h:179-180).
    """
    q, k, v = _f64(q), _f64(k), _f64(v)
    B, Sq, H, D = q.shape
    Sk, Hk = k.shape[1], k.shape[2]
    assert H % Hk == 0
    scale = D ** -0.5 if softmax_scale is None else float(softmax_scale)
    _, wl, wr = normalize_window(Sq, Sk, causal, window[0], window[1], alibi_slopes is not None)
    vis = visible_mask(Sq, Sk, wl, wr)
    out = np.zeros((B, Sq, H, D))
    lse = np.full((B, H, Sq), np.inf)
    if Sk == 0:
        return out, lse
    if alibi_slopes is not None:
        alibi_slopes = _f64(alibi_slopes)
        if alibi_slopes.ndim == 1:
            alibi_slopes = np.broadcast_to(alibi_slopes[None, :], (B, H))
    g = H // Hk
    for b in range(B):
        for h in range(H):
            slope = None if alibi_slopes is None else alibi_slopes[b, h]
            s = _scores(q[b, :, h], k[b, :, h // g], scale, softcap, vis, slope, Sk - Sq)
            m = s.max(axis=1)
            live = np.isfinite(m)
            m_safe = np.where(live, m, 0.0)
            p = np.exp(s - m_safe[:, None])
            l = p.sum(axis=1)
            l_safe = np.where(live, l, 1.0)
            if dropout_mask is not None:
                p = p * np.asarray(dropout_mask[b, h], dtype=np.float64) / (1.0 - dropout_p)
            o = (p @ v[b, :, h // g]) / l_safe[:, None]
            out[b, :, h] = np.where(live[:, None], o, 0.0)
            lse[b, h] = np.where(live, m_safe + np.log(l_safe), np.inf)
    return out, lse


def attention_bwd(dout, q, k, v, out=None, lse=None, softmax_scale: Optional[float] = None,
                  causal: bool = False, window: Tuple[int, int] = (-1, -1), softcap: float = 0.0,
                  alibi_slopes=None, dropout_p: float = 0.0, dropout_mask=None):
    """Backward oracle: returns dq (B,Sq,H,D), dk, dv (B,Sk,Hk,D), delta (B,H,Sq), all f64.

    With dropout (flash_bwd_kernel.h:560-600): Z = mask/(1-p); dV = (P*Z)^T dO, dP = (dO V^T)*Z,
    delta = rowsum(dO*O) with the dropped O.

    Restates flash_bwd_kernel.h:536-733: P_ij = exp(score_ij - LSE_i) (0 when LSE_i = +inf),
    dV = P^T dO, dP = dO V^T, delta_i = sum_d dO_id O_id, dS = P * (dP - delta),
    dQ = scale * dS K, dK = scale * dS^T Q; GQA grads are summed over the query-head group.
    With softcap the chain rule multiplies dS by (1 - tanh^2) (utils.h:395-409 / flash_bwd_kernel.h:588).
    If ``out``/``lse`` are None they are recomputed exactly.
    """
    dout, q, k, v = _f64(dout), _f64(q), _f64(k), _f64(v)
    B, Sq, H, D = q.shape
    Sk, Hk = k.shape[1], k.shape[2]
    scale = D ** -0.5 if softmax_scale is None else float(softmax_scale)
    if out is None or lse is None:
        out, lse = attention_fwd(q, k, v, scale, causal, window, softcap, alibi_slopes, dropout_p, dropout_mask)
    out, lse = _f64(out), _f64(lse)
    _, wl, wr = normalize_window(Sq, Sk, causal, window[0], window[1], alibi_slopes is not None)
    vis = visible_mask(Sq, Sk, wl, wr)
    dq = np.zeros_like(q)
    dk = np.zeros_like(k)
    dv = np.zeros_like(v)
    delta = (dout * out).sum(axis=-1).transpose(0, 2, 1).copy()  # (B,H,Sq)
    if Sk == 0 or Sq == 0:
        return dq, dk, dv, delta
    if alibi_slopes is not None:
        alibi_slopes = _f64(alibi_slopes)
        if alibi_slopes.ndim == 1:
            alibi_slopes = np.broadcast_to(alibi_slopes[None, :], (B, H))
    g = H // Hk
    for b in range(B):
        for h in range(H):
            hk = h // g
            slope = None if alibi_slopes is None else alibi_slopes[b, h]
            s = _scores(q[b, :, h], k[b, :, hk], scale, softcap, vis, slope, Sk - Sq)
            row_lse = lse[b, h]
            live = np.isfinite(row_lse)
            p = np.where(live[:, None], np.exp(s - np.where(live, row_lse, 0.0)[:, None]), 0.0)
            do = dout[b, :, h]
            z = 1.0 if dropout_mask is None else np.asarray(dropout_mask[b, h], dtype=np.float64) / (1.0 - dropout_p)
            dv[b, :, hk] += (p * z).T @ do
            dp = (do @ v[b, :, hk].T) * z
            ds = p * (dp - delta[b, h][:, None])
            if softcap > 0.0:
                raw = (q[b, :, h] @ k[b, :, hk].T) * scale
                ds = ds * (1.0 - np.tanh(raw / softcap) ** 2)
            dq[b, :, h] = scale * (ds @ k[b, :, hk])
            dk[b, :, hk] += scale * (ds.T @ q[b, :, h])
    return dq, dk, dv, delta


def _as_int_list(cu) -> Sequence[int]:
    if hasattr(cu, "detach"):
        cu = cu.detach().to("cpu").numpy()
    return [int(x) for x in np.asarray(cu).reshape(-1)]


def varlen_fwd(q, k, v, cu_seqlens_q, cu_seqlens_k, softmax_scale: Optional[float] = None,
               causal: bool = False, window: Tuple[int, int] = (-1, -1), softcap: float = 0.0,
               alibi_slopes=None):
    """Packed varlen forward: q (total_q,H,D), k,v (total_k,Hk,D), cu_seqlens int32 (B+1).

    Sequence b owns rows cu[b] .. cu[b+1]-1 (block_info.h:12-45); result equals the
    concatenation of per-sequence fixed-length calls.  Returns out (total_q,H,D), lse (H,total_q).
    """
    q, k, v = _f64(q), _f64(k), _f64(v)
    cq, ck = _as_int_list(cu_seqlens_q), _as_int_list(cu_seqlens_k)
    H = q.shape[1]
    D = q.shape[2]
    scale = D ** -0.5 if softmax_scale is None else float(softmax_scale)
    out = np.zeros_like(q)
    lse = np.full((H, q.shape[0]), np.inf)
    slopes = None if alibi_slopes is None else _f64(alibi_slopes)
    for b in range(len(cq) - 1):
        q0, q1, k0, k1 = cq[b], cq[b + 1], ck[b], ck[b + 1]
        if q1 == q0:
            continue
        sl = None
        if slopes is not None:
            sl = slopes if slopes.ndim == 1 else slopes[b:b + 1]
        o, l = attention_fwd(q[None, q0:q1], k[None, k0:k1], v[None, k0:k1], scale, causal, window, softcap, sl)
        out[q0:q1] = o[0]
        lse[:, q0:q1] = l[0]
    return out, lse


def varlen_bwd(dout, q, k, v, cu_seqlens_q, cu_seqlens_k, softmax_scale: Optional[float] = None,
               causal: bool = False, window: Tuple[int, int] = (-1, -1), softcap: float = 0.0,
               alibi_slopes=None, out=None, lse=None):
    """Packed varlen backward (exact out/LSE recomputed per sequence unless given).

    Returns dq (total_q,H,D), dk, dv (total_k,Hk,D), delta (H,total_q)."""
    dout, q, k, v = _f64(dout), _f64(q), _f64(k), _f64(v)
    cq, ck = _as_int_list(cu_seqlens_q), _as_int_list(cu_seqlens_k)
    H, D = q.shape[1], q.shape[2]
    scale = D ** -0.5 if softmax_scale is None else float(softmax_scale)
    dq, dk, dv = np.zeros_like(q), np.zeros_like(k), np.zeros_like(v)
    delta = np.zeros((H, q.shape[0]))
    slopes = None if alibi_slopes is None else _f64(alibi_slopes)
    out64 = None if out is None else _f64(out)
    lse64 = None if lse is None else _f64(lse)
    for b in range(len(cq) - 1):
        q0, q1, k0, k1 = cq[b], cq[b + 1], ck[b], ck[b + 1]
        if q1 == q0:
            continue
        sl = None
        if slopes is not None:
            sl = slopes if slopes.ndim == 1 else slopes[b:b + 1]
        o = None if out64 is None else out64[None, q0:q1]
        l = None if lse64 is None else lse64[None, :, q0:q1]
        a, bb, c, d = attention_bwd(dout[None, q0:q1], q[None, q0:q1], k[None, k0:k1], v[None, k0:k1],
                                    o, l, scale, causal, window, softcap, sl)
        dq[q0:q1] = a[0]
        dk[k0:k1] += bb[0]
        dv[k0:k1] += c[0]
        delta[:, q0:q1] = d[0]
    return dq, dk, dv, delta
