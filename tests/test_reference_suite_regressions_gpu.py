"""Regressions found by running the REFERENCE's own acceptance suites (tests/test_flash_attn_ck.py, tests/test_flash_attn.py; sampled run recorded in
profiles/r04_reference_suite.txt) on our module, restated here so they run on every GPU box without the reference tree:

  * one visible key (tests/test_flash_attn.py::test_flash_attn_causal[1-239-True-..], ::test_flash_attn_varlen_causal, ::test_flash_attn_splitkv):
    P must be exactly 1, so dV of that key is exactly the sum of the dO rows that see it and dQ = dK = 0.  The dK/dV kernel used to pre-scale K by
    softmax_scale*log2e rounded to the input dtype: its exponent then differed from the forward's LSE by |score| * 2^-9 and P came out as 1 + 2^-8.
  * tests/test_flash_attn*.py::test_flash_attn_bwd_overflow (fp16, q*5, k*3: scores of +-70 log2 units): dV within 5x PyTorch's fp16 error + 1e-3.
  * tests/test_flash_attn_ck.py::test_flash_attn_kvcache with head dim 59: the reference zero-pads q and both caches to the next multiple of 8
    (csrc/flash_attn/flash_api.cpp:1340-1350) and copies appended keys back (:1517-1527).
  * the retired `generator` slot takes any object and refuses everything but None with the reference's message (flash_api.cpp:382-386)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-attention_amd"))
from tests._util import attention_torch, max_abs  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    import flash_attn_amd
    return flash_attn_amd


def _grads(fn, q, k, v, g):
    q, k, v = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
    out = fn(q, k, v)
    return (out,) + torch.autograd.grad(out, (q, k, v), g)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("d", [40, 64, 128, 192, 256])
@pytest.mark.parametrize("sq", [1, 239, 700])
def test_one_visible_key_is_exact(fa, sq, d, dtype):
    """Causal, seqlen_k = 1, bottom-right aligned: only the LAST query sees the key.  The reference's rule there is |err| <= 2 * 0 + 1e-5."""
    torch.manual_seed(sq + d)
    B, H = 8, 9
    q = torch.randn(B, sq, H, d, device="cuda", dtype=dtype)
    k = torch.randn(B, 1, H, d, device="cuda", dtype=dtype)
    v = torch.randn(B, 1, H, d, device="cuda", dtype=dtype)
    g = torch.randn(B, sq, H, d, device="cuda", dtype=dtype)
    out, dq, dk, dv = _grads(lambda a, b, c: fa.flash_attn_func(a, b, c, causal=True), q, k, v, g)
    ref = _grads(lambda a, b, c: attention_torch(a, b, c, causal=True)[0], q, k, v, g)
    for name, got, want in zip(("out", "dq", "dk", "dv"), (out, dq, dk, dv), ref):
        assert max_abs(got, want) <= 1e-5, (name, max_abs(got, want))


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("d", [16, 32, 64])
@pytest.mark.parametrize("seqlen", [1, 2, 5, 17, 128])
def test_bwd_overflow_fp16(fa, seqlen, d, causal):
    """tests/test_flash_attn.py:2247-2292 (fp16 only there, too)."""
    torch.manual_seed(0)
    dtype = torch.float16
    q = torch.randn(2, seqlen, 5, d, device="cuda", dtype=dtype) * 5
    k = torch.randn(2, seqlen, 5, d, device="cuda", dtype=dtype) * 3
    v = torch.randn(2, seqlen, 5, d, device="cuda", dtype=dtype) * 3
    g = torch.randn(2, seqlen, 5, d, device="cuda", dtype=dtype)
    got = _grads(lambda a, b, c: fa.flash_attn_func(a, b, c, causal=causal), q, k, v, g)
    ref = _grads(lambda a, b, c: attention_torch(a, b, c, causal=causal)[0], q, k, v, g)
    pt = _grads(lambda a, b, c: attention_torch(a, b, c, causal=causal, upcast=False, reorder=True)[0], q, k, v, g)
    assert max_abs(got[0], ref[0]) <= 2 * max_abs(pt[0], ref[0])
    for i, name in ((1, "dq"), (2, "dk"), (3, "dv")):
        assert torch.isfinite(got[i]).all(), name
        assert max_abs(got[i], ref[i]) <= 5 * max_abs(pt[i], ref[i]) + 1e-3, (name, max_abs(got[i], ref[i]), max_abs(pt[i], ref[i]))


@pytest.mark.parametrize("new_kv", [False, True])
@pytest.mark.parametrize("d", [59, 111])
def test_kvcache_head_dim_not_a_multiple_of_8(fa, d, new_kv):
    import flash_attn_2_cuda as ext
    torch.manual_seed(d)
    B, S, H, Hk = 3, 800, 6, 2
    dtype = torch.float16
    q = torch.randn(B, 1, H, d, device="cuda", dtype=dtype)
    kc = torch.randn(B, S, Hk, d, device="cuda", dtype=dtype)
    vc = torch.randn_like(kc)
    lens = torch.tensor([5, 400, 799], dtype=torch.int32, device="cuda")
    kn = torch.randn(B, 1, Hk, d, device="cuda", dtype=dtype) if new_kv else None
    vn = torch.randn(B, 1, Hk, d, device="cuda", dtype=dtype) if new_kv else None
    kc0, vc0 = kc.clone(), vc.clone()
    out, _ = ext.fwd_kvcache(q, kc, vc, kn, vn, lens, None, None, None, None, None, None, None, d ** -0.5, True, -1, -1, 0.0, True, 0)
    assert out.shape == q.shape
    for b in range(B):
        n = int(lens[b])
        kk, vv = kc0[b:b + 1, :n], vc0[b:b + 1, :n]
        if new_kv:
            kk, vv = torch.cat([kk, kn[b:b + 1]], 1), torch.cat([vv, vn[b:b + 1]], 1)
            assert torch.equal(kc[b, n], kn[b, 0]) and torch.equal(vc[b, n], vn[b, 0])       # appended in place, at the ORIGINAL head dim
            assert torch.equal(kc[b, :n], kc0[b, :n]) and torch.equal(kc[b, n + 1:], kc0[b, n + 1:])
        o_ref, _ = attention_torch(q[b:b + 1], kk, vv)
        assert max_abs(out[b:b + 1], o_ref) < 4e-3, b


def test_generator_slot_must_be_none():
    import flash_attn_2_cuda as ext
    q = torch.randn(1, 1, 2, 8, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(1, 1, 1, 8, device="cuda", dtype=torch.bfloat16)
    lse = torch.randn(1, 2, 1, device="cuda", dtype=torch.float32)
    bad = torch.empty(1, device="cuda")
    match = r"generator` argument is no longer supported"
    with pytest.raises(RuntimeError, match=match):
        ext.fwd(q, k, k, None, None, 0.0, 0.35, True, -1, -1, 0.0, False, bad)
    with pytest.raises(RuntimeError, match=match):
        ext.fwd(q, k, k, None, None, 0.0, 0.35, True, -1, -1, 0.0, False, torch.Generator())
    with pytest.raises(RuntimeError, match=match):
        ext.bwd(q, q, k, k, q, lse, None, None, None, None, 0.0, 0.35, True, -1, -1, 0.0, False, bad, None)
    cu = torch.tensor([0, 1], dtype=torch.int32, device="cuda")
    qf, kf = q.view(1, 2, 8), k.view(1, 1, 8)
    with pytest.raises(RuntimeError, match=match):
        ext.varlen_fwd(qf, kf, kf, None, cu, cu, None, None, None, None, 1, 1, 0.0, 0.35, False, True, -1, -1, 0.0, False, bad)
    with pytest.raises(RuntimeError, match=match):
        ext.varlen_bwd(qf, qf, kf, kf, qf, lse.view(2, 1), None, None, None, cu, cu, None, 1, 1, 0.0, 0.35, False, True, -1, -1, 0.0, False, bad, None)


@pytest.mark.parametrize("hk", [1, 2])
def test_varlen_one_query_per_sequence_grouped_heads_backward(fa, hk):
    """tests/test_flash_attn*.py::test_flash_attn_varlen_output[..-1-147-..-mqa-..]: one query row per sequence with grouped heads takes the head-packing
    swap of mha_varlen_fwd (flash_api.cpp:620-629); with ONE KV head the un-swapped LSE came back as a strided view and varlen_bwd refused it."""
    torch.manual_seed(3)
    B, H, D, Sk = 4, 6, 64, 147
    q = torch.randn(B, H, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    k = torch.randn(B * Sk, hk, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    v = torch.randn(B * Sk, hk, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    cu_q = torch.arange(0, B + 1, dtype=torch.int32, device="cuda")
    cu_k = torch.arange(0, (B + 1) * Sk, Sk, dtype=torch.int32, device="cuda")
    out = fa.flash_attn_varlen_func(q, k, v, cu_q, cu_k, 1, Sk)
    g = torch.randn_like(out)
    dq, dk, dv = torch.autograd.grad(out, (q, k, v), g)
    qr, kr, vr = q.detach().view(B, 1, H, D), k.detach().view(B, Sk, hk, D), v.detach().view(B, Sk, hk, D)
    ref = _grads(lambda a, b, c: attention_torch(a, b, c)[0], qr, kr, vr, g.view(B, 1, H, D))
    assert max_abs(out.view(B, 1, H, D), ref[0]) < 2e-2
    assert max_abs(dq.view(B, 1, H, D), ref[1]) < 3e-2 and max_abs(dk.view(B, Sk, hk, D), ref[2]) < 3e-2 and max_abs(dv.view(B, Sk, hk, D), ref[3]) < 3e-2
