"""GPU parity tests of the backward path (through the C ABI).

Tolerance rule = the reference's (tests/test_flash_attn.py:1130-1132): each gradient's max error
against the fp32 reference is at most 3x the error of the same-dtype plain-PyTorch gradients
(+ small absolute floor); golden cases compare against reference attention_ref + autograd outputs.
"""
import numpy as np
import pytest
import torch

from tests._util import attention_torch, case_meta, golden_inputs, max_abs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from flash_attn_amd import backend
    return backend


def _fwd_bwd(be, q, k, v, do, causal=False, window=(-1, -1), alibi=None):
    scale = q.shape[-1] ** -0.5
    out, lse, _, _ = be.fwd(q, k, v, None, alibi, 0.0, scale, causal, window[0], window[1], 0.0, False, None)
    dq, dk, dv, delta = be.bwd(do, q, k, v, out, lse, None, None, None, alibi, 0.0, scale, causal, window[0], window[1],
                               0.0, False, None, None)
    return out, lse, dq, dk, dv, delta


def _ref_grads(q, k, v, do, causal, window, upcast):
    q, k, v = (t.detach().clone().requires_grad_() for t in (q, k, v))
    o, _ = attention_torch(q, k, v, causal, window, upcast=upcast, reorder=not upcast)
    g = torch.autograd.grad(o, (q, k, v), do.to(o.dtype))
    return o, g


GOLDEN = ["mha_full_d64", "mha_causal_d128", "gqa_causal_sq_gt_sk", "mqa_local_d128", "gqa_causal_window_d128",
          "local_left_only_d64", "local_right_only_d64", "tiny_sq1", "alibi_d64", "d32_full", "d96_causal", "d256_causal"]


@pytest.mark.parametrize("name", GOLDEN)
def test_backward_matches_reference_golden(be, golden_cases, name):
    case = golden_cases[name]
    m = case_meta(case)
    q, k, v, do = golden_inputs(case, "cuda")
    alibi = None if m["alibi"] is None else torch.from_numpy(np.asarray(m["alibi"], dtype=np.float32)).cuda()
    out, lse, dq, dk, dv, delta = _fwd_bwd(be, q, k, v, do, m["causal"], m["window"], alibi)
    for nm, got in (("dq", dq), ("dk", dk), ("dv", dv)):
        ref = torch.from_numpy(case[nm]).cuda()
        tol = 2e-2 * max(1.0, float(ref.abs().max()))
        assert max_abs(got.float(), ref) < tol, (name, nm, max_abs(got.float(), ref), tol)
    dref = (do.float() * out.float()).sum(-1).transpose(1, 2)
    assert max_abs(delta, dref) < 1e-3 * max(1.0, float(dref.abs().max()))


SEQ = [(113, 203), (128, 217), (108, 256), (256, 512), (512, 256), (1024, 1024), (1023, 1024), (1024, 1023), (2048, 2048)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mha_type", ["mha", "gqa", "mqa"])
@pytest.mark.parametrize("mode", ["full", "causal", "local"])
@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("sq,sk", SEQ)
def test_backward_vs_fp32_reference(be, sq, sk, d, mode, mha_type, dtype):
    torch.manual_seed(0)
    B, H = 2, 6
    Hk = {"mha": 6, "gqa": 2, "mqa": 1}[mha_type]
    q = torch.randn(B, sq, H, d, device="cuda", dtype=dtype)
    k = torch.randn(B, sk, Hk, d, device="cuda", dtype=dtype)
    v = torch.randn(B, sk, Hk, d, device="cuda", dtype=dtype)
    do = torch.randn(B, sq, H, d, device="cuda", dtype=dtype)
    causal = mode == "causal"
    window = (-1, -1)
    if mode == "local":
        g = torch.Generator().manual_seed(sq * 7 + sk)
        window = tuple(int(x) for x in torch.randint(0, sk, (2,), generator=g))
    out, lse, dq, dk, dv, _ = _fwd_bwd(be, q, k, v, do, causal, window)
    _, ref = _ref_grads(q.float(), k.float(), v.float(), do.float(), causal, window, True)
    _, pt = _ref_grads(q, k, v, do, causal, window, False)
    for nm, got, r, p_ in zip(("dq", "dk", "dv"), (dq, dk, dv), ref, pt):
        err, err_pt = max_abs(got.float(), r), max_abs(p_.float(), r)
        assert err <= 3 * err_pt + 1e-4, (nm, err, err_pt)
        assert not torch.isnan(got).any()


def test_backward_noncontiguous_dout(be):
    """dO may be non-contiguous in batch/seq/head (reference tests/test_flash_attn.py:2303-2347)."""
    torch.manual_seed(4)
    q = torch.randn(2, 300, 4, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(2, 300, 4, 128, device="cuda", dtype=torch.bfloat16)
    v = torch.randn_like(k)
    do_t = torch.randn(300, 2, 4, 128, device="cuda", dtype=torch.bfloat16)
    do = do_t.transpose(0, 1)
    a = _fwd_bwd(be, q, k, v, do, True)
    b_ = _fwd_bwd(be, q, k, v, do.contiguous(), True)
    for x, y in zip(a[2:5], b_[2:5]):
        assert torch.equal(x, y)


def _varlen(be, q, k, v, do, cu_q, cu_k, mq, mk, causal):
    d = q.shape[-1]
    scale = d ** -0.5
    out, lse, _, _ = be.varlen_fwd(q, k, v, None, cu_q, cu_k, None, None, None, None, mq, mk, 0.0, scale, False, causal,
                                   -1, -1, 0.0, False, None)
    dq, dk, dv, delta = be.varlen_bwd(do, q, k, v, out, lse, None, None, None, cu_q, cu_k, None, mq, mk, 0.0, scale, False,
                                      causal, -1, -1, 0.0, False, None, None)
    return out, lse, dq, dk, dv


@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("causal", [False, True])
def test_varlen_backward_equals_per_sequence_bit_exact(be, knobs, d, causal):
    knobs.set("FA_FWD_NW", "34")  # pin the forward schedule: out / LSE feed the backward
    torch.manual_seed(5)
    lens_q = [0, 76, 34, 146, 1, 300, 257]
    lens_k = [5, 76, 1, 300, 77, 300, 255]
    H, Hk = 4, 2
    cu_q = torch.tensor([0] + list(np.cumsum(lens_q)), dtype=torch.int32, device="cuda")
    cu_k = torch.tensor([0] + list(np.cumsum(lens_k)), dtype=torch.int32, device="cuda")
    q = torch.randn(int(cu_q[-1]), H, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(int(cu_k[-1]), Hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn_like(k)
    do = torch.randn_like(q)
    out, lse, dq, dk, dv = _varlen(be, q, k, v, do, cu_q, cu_k, max(lens_q), max(lens_k), causal)
    for b in range(len(lens_q)):
        a0, a1, b0, b1 = int(cu_q[b]), int(cu_q[b + 1]), int(cu_k[b]), int(cu_k[b + 1])
        if a1 == a0:
            assert torch.all(dk[b0:b1] == 0) and torch.all(dv[b0:b1] == 0)
            continue
        r = _fwd_bwd(be, q[None, a0:a1], k[None, b0:b1], v[None, b0:b1], do[None, a0:a1], causal)
        assert torch.equal(dq[a0:a1], r[2][0]) and torch.equal(dk[b0:b1], r[3][0]) and torch.equal(dv[b0:b1], r[4][0]), b


def test_reference_known_answer_layouts(be):
    """cu_seqlens fixtures of the reference regression tests: no NaN for (q=[0,76,110,256], k=[0,1,2,3])
    (tests/test_flash_attn.py:2363-2380) and exact-zero dK/dV for empty q sequences
    (tests/test_flash_attn_ck.py:1522-1560)."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "known_answers.npz"))
    torch.manual_seed(6)
    cu_q = torch.from_numpy(z["cu_bwd_varlen_overflow_q"]).cuda()
    cu_k = torch.from_numpy(z["cu_bwd_varlen_overflow_k"]).cuda()
    q = torch.randn(256, 2, 64, device="cuda", dtype=torch.bfloat16) * 5
    k = torch.randn(3, 2, 64, device="cuda", dtype=torch.bfloat16) * 5
    v = torch.randn_like(k) * 5
    do = torch.randn_like(q)
    out, lse, dq, dk, dv = _varlen(be, q, k, v, do, cu_q, cu_k, 256, 3, False)
    for t in (out, dq, dk, dv):
        assert not torch.isnan(t).any() and not torch.isinf(t).any()
    cu_q = torch.from_numpy(z["cu_seqq_zero_q"]).cuda()
    cu_k = torch.from_numpy(z["cu_seqq_zero_k"]).cuda()
    for d in (64, 128):
        for hk in (1, 8):
            q = torch.randn(512, 8, d, device="cuda", dtype=torch.bfloat16)
            k = torch.randn(1536, hk, d, device="cuda", dtype=torch.bfloat16)
            v = torch.randn_like(k)
            do = torch.randn_like(q)
            out, lse, dq, dk, dv = _varlen(be, q, k, v, do, cu_q, cu_k, 256, 768, True)
            assert torch.all(dk[:503] == 0) and torch.all(dv[:503] == 0)
            assert not torch.isnan(dq).any() and not torch.isnan(dk).any() and not torch.isnan(dv).any()


def test_backward_bitwise_deterministic(be):
    """Reference race-condition test pattern (tests/test_flash_attn.py:2199-2237): repeated runs are bit-identical
    (here dq too: this backward has no atomics)."""
    torch.manual_seed(7)
    q = torch.randn(8, 513, 8, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(8, 700, 2, 128, device="cuda", dtype=torch.bfloat16)
    v = torch.randn_like(k)
    do = torch.randn_like(q)
    r0 = _fwd_bwd(be, q, k, v, do, True)
    for _ in range(20):
        r = _fwd_bwd(be, q, k, v, do, True)
        for x, y in zip(r0[:5], r[:5]):
            assert torch.equal(x, y)


@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("alibi", [False, True])
def test_softcap_backward_vs_oracle(be, d, alibi):
    """softcap (and softcap+ALiBi) gradients against the fp64 oracle: dS carries (1 - tanh^2)."""
    from oracle import attention_oracle as orc
    torch.manual_seed(8)
    B, Sq, Sk, H, Hk = 2, 150, 231, 4, 2
    q = torch.randn(B, Sq, H, d, device="cuda", dtype=torch.bfloat16) * 3
    k = torch.randn(B, Sk, Hk, d, device="cuda", dtype=torch.bfloat16) * 3
    v = torch.randn(B, Sk, Hk, d, device="cuda", dtype=torch.bfloat16)
    do = torch.randn(B, Sq, H, d, device="cuda", dtype=torch.bfloat16)
    slopes = (torch.rand(B, H, device="cuda") * 0.3).float() if alibi else None
    scale, cap = d ** -0.5, 5.0
    out, lse, _, _ = be.fwd(q, k, v, None, slopes, 0.0, scale, True, -1, -1, cap, False, None)
    dq, dk, dv, _ = be.bwd(do, q, k, v, out, lse, None, None, None, slopes, 0.0, scale, True, -1, -1, cap, False, None, None)
    ref = orc.attention_bwd(do, q, k, v, None, None, scale, True, (-1, -1), cap, None if slopes is None else slopes.cpu())
    o_ref, _ = orc.attention_fwd(q, k, v, scale, True, (-1, -1), cap, None if slopes is None else slopes.cpu())
    assert max_abs(out.float().cpu(), torch.from_numpy(o_ref)) < 2e-2
    for got, r in zip((dq, dk, dv), ref[:3]):
        r = torch.from_numpy(r)
        assert max_abs(got.float().cpu(), r) < 3e-2 * max(1.0, float(r.abs().max())), float(r.abs().max())


@pytest.mark.parametrize("causal,window", [(True, (-1, -1)), (False, (-1, -1)), (False, (300, 100))])
@pytest.mark.parametrize("hk", [8, 2])
def test_varlen_work_list_equals_dense_grid(be, knobs, hk, causal, window):
    """Uneven packed batch (long-tail lengths, empty sequences included): the scheduled work list (forward, dQ and dK/dV)
    gives bit-for-bit the results of the dense max_seqlen grid, and both match the oracle on sampled sequences."""
    from oracle import attention_oracle as orc
    g = torch.Generator().manual_seed(5)
    lens = [int(x) for x in (torch.rand(90, generator=g) ** 4 * 1500).long()] + [2048, 0, 1, 1900]
    lens_k = [max(0, l + int(d)) for l, d in zip(lens, torch.randint(-40, 40, (len(lens),), generator=g))] if not causal else lens
    H, d = 8, 128
    cu_q = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    cu_k = torch.tensor([0] + list(np.cumsum(lens_k)), dtype=torch.int32, device="cuda")
    torch.manual_seed(0)
    q = torch.randn(sum(lens), H, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(sum(lens_k), hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn_like(k)
    do = torch.randn_like(q)
    scale = d ** -0.5

    def run():
        out, lse, _, _ = be.varlen_fwd(q, k, v, None, cu_q, cu_k, None, None, None, None, max(lens), max(lens_k), 0.0, scale, False, causal,
                                       window[0], window[1], 0.0, False, None)
        dq, dk, dv, _ = be.varlen_bwd(do, q, k, v, out, lse, None, None, None, cu_q, cu_k, None, max(lens), max(lens_k), 0.0, scale, False,
                                      causal, window[0], window[1], 0.0, False, None, None)
        return out, lse, dq, dk, dv

    knobs.set("FA_FWD_NW", "34")
    listed = run()
    knobs.set("FA_VARLEN_LIST", "0")
    dense = run()
    for a, b_ in zip(listed, dense):
        assert torch.equal(a, b_)
    for b in (0, 5, 90, 91, 92, 93):
        qs, ks = slice(int(cu_q[b]), int(cu_q[b + 1])), slice(int(cu_k[b]), int(cu_k[b + 1]))
        if lens[b] == 0:
            assert float(listed[3][ks].abs().max()) == 0.0 if lens_k[b] else True   # no query sees these keys: exact zeros
            continue
        o_ref, _ = orc.attention_fwd(q[qs][None], k[ks][None], v[ks][None], scale, causal, window)
        assert max_abs(listed[0][qs].float().cpu(), torch.from_numpy(o_ref[0])) < 2e-2
        rq, rk, rv, _ = orc.attention_bwd(do[qs][None], q[qs][None], k[ks][None], v[ks][None], None, None, scale, causal, window)
        for got, ref in ((listed[2][qs], rq[0]), (listed[3][ks], rk[0]), (listed[4][ks], rv[0])):
            assert max_abs(got.float().cpu(), torch.from_numpy(ref)) < 4e-2 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("shape", ["one_long_many_short", "thousands_of_short"])
def test_varlen_work_list_extreme_batches(be, knobs, shape):
    """The schedule pre-pass at its limits: a sequence longer than 64k keys next to short ones (work buckets are scaled to the
    longest sequence; sequences of more than 8 blocks are walked by the whole workgroup) and a batch of thousands of
    sequences (one thread each).  Work list == dense grid bit for bit, forward and backward."""
    g = torch.Generator().manual_seed(7)
    if shape == "one_long_many_short":
        lens = [70000, 3, 129, 2500] + [int(x) for x in torch.randint(1, 200, (40,), generator=g)]
        H, hk, d = 2, 1, 64
    else:
        lens = [int(x) for x in torch.randint(0, 70, (3000,), generator=g)] + [1400, 900]
        H, hk, d = 2, 2, 64
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    torch.manual_seed(1)
    q = torch.randn(sum(lens), H, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(sum(lens), hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn_like(k)
    do = torch.randn_like(q)
    scale = d ** -0.5

    def run():
        out, lse, _, _ = be.varlen_fwd(q, k, v, None, cu, cu, None, None, None, None, max(lens), max(lens), 0.0, scale, False, True, -1, -1, 0.0,
                                       False, None)
        sched = be.last_schedule()
        dq, dk, dv, _ = be.varlen_bwd(do, q, k, v, out, lse, None, None, None, cu, cu, None, max(lens), max(lens), 0.0, scale, False, True,
                                      -1, -1, 0.0, False, None, None)
        return (out, lse, dq, dk, dv), sched, be.last_schedule()

    listed, s_f, s_b = run()
    assert s_f["fwd_list"] == 1 and s_b["bwd_list"] == 3, (s_f, s_b)
    knobs.set("FA_VARLEN_LIST", "0")
    dense, s_f0, _ = run()
    assert s_f0["fwd_list"] == 0
    for a, b_ in zip(listed, dense):
        assert torch.equal(a, b_)
    # spot check against the fp32 reference on the tail of the longest sequence's neighbour and a short sequence
    from tests._util import attention_torch
    for b in (1, 2, len(lens) - 1):
        sl = slice(int(cu[b]), int(cu[b + 1]))
        o32, _ = attention_torch(q[sl][None].float(), k[sl][None].float(), v[sl][None].float(), True)
        assert max_abs(listed[0][sl].float(), o32[0]) < 2e-2
