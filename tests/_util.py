"""Shared helpers for the parity tests (test infrastructure only)."""
from __future__ import annotations

import math

import numpy as np
import torch


def bf16_bits_to_f32(bits: np.ndarray) -> np.ndarray:
    return (bits.astype(np.uint32) << 16).view(np.float32)


def golden_inputs(case, device=None, dtype=torch.bfloat16):
    """q, k, v, do tensors of a golden case (exactly bf16-representable values)."""
    out = []
    for nm in ("q", "k", "v", "do"):
        t = torch.from_numpy(bf16_bits_to_f32(case[nm + "_bf16bits"]).copy())
        out.append(t.to(device=device, dtype=dtype) if device is not None else t)
    return out


def case_meta(case):
    B, Sq, Sk, H, Hk, D, causal, wl, wr = [int(x) for x in case["meta"]]
    return dict(B=B, Sq=Sq, Sk=Sk, H=H, Hk=Hk, D=D, causal=bool(causal), window=(wl, wr),
                softcap=float(case["softcap"][0]),
                alibi=case.get("alibi_slopes"))


def local_mask_torch(sq, sk, window, device):
    """True = masked; same predicate as the oracle's visible_mask (bottom-right aligned)."""
    i = torch.arange(sq, device=device)[:, None]
    j = torch.arange(sk, device=device)[None, :]
    wl, wr = window
    masked = torch.zeros(sq, sk, dtype=torch.bool, device=device)
    if wr >= 0:
        masked |= j > i + (sk - sq) + wr
    if wl >= 0:
        masked |= j < i + (sk - sq) - wl
    return masked


def attention_torch(q, k, v, causal=False, window=(-1, -1), softmax_scale=None, upcast=True, reorder=False):
    """Plain PyTorch attention on the tensors' device (large-size reference / low-precision baseline).

    upcast=True: fp32 math (the reference point); upcast=False: math in the input dtype (the
    'PyTorch baseline' whose error calibrates the tolerance, as in the reference's tests).
    Returns out (B,Sq,H,D) in the input dtype and lse (B,H,Sq) fp32.
    """
    dt = q.dtype
    if causal:
        window = (window[0], 0)
    if upcast:
        q, k, v = q.float(), k.float(), v.float()
    B, Sq, H, D = q.shape
    Sk, Hk = k.shape[1], k.shape[2]
    g = H // Hk
    k = k.repeat_interleave(g, dim=2)
    v = v.repeat_interleave(g, dim=2)
    scale = D ** -0.5 if softmax_scale is None else softmax_scale
    if reorder:
        s = torch.einsum("bthd,bshd->bhts", q, k * scale)
    else:
        s = torch.einsum("bthd,bshd->bhts", q * scale, k)
    if window[0] >= 0 or window[1] >= 0:
        wl = -1 if window[0] >= Sk else window[0]
        wr = -1 if window[1] >= Sk else window[1]
        m = local_mask_torch(Sq, Sk, (wl, wr), q.device)
        s = s.masked_fill(m, float("-inf"))
        dead = m.all(dim=-1)
    else:
        dead = None
    lse = torch.logsumexp(s.float(), dim=-1)
    p = torch.softmax(s, dim=-1)
    if dead is not None:
        p = p.masked_fill(dead[None, None, :, None], 0.0)
        lse = lse.masked_fill(dead[None, None, :], float("inf"))
    o = torch.einsum("bhts,bshd->bthd", p.to(v.dtype), v)
    return o.to(dt), lse


def max_abs(a, b):
    return float((a.double() - b.double()).abs().max()) if a.numel() else 0.0
