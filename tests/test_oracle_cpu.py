"""CPU suite: the oracle against the reference's golden vectors and known answers."""
import os

import numpy as np
import pytest

from oracle import attention_oracle as orc
from tests._util import bf16_bits_to_f32, case_meta

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs(case):
    return [bf16_bits_to_f32(case[n + "_bf16bits"]).astype(np.float64) for n in ("q", "k", "v", "do")]


def test_golden_names(golden_cases):
    assert len(golden_cases) >= 12


@pytest.mark.parametrize("name", [
    "mha_full_d64", "mha_causal_d128", "gqa_causal_sq_gt_sk", "mqa_local_d128", "gqa_causal_window_d128",
    "local_left_only_d64", "local_right_only_d64", "tiny_sq1", "softcap_d64", "alibi_d64", "d32_full",
    "d96_causal", "d256_causal"])
def test_oracle_matches_reference_golden(golden_cases, name):
    """oracle fwd/bwd == reference attention_ref (+autograd) outputs, fp32-roundoff tolerance."""
    case = golden_cases[name]
    m = case_meta(case)
    q, k, v, do = _inputs(case)
    out, lse = orc.attention_fwd(q, k, v, None, m["causal"], m["window"], m["softcap"], m["alibi"])
    assert np.abs(out - case["out"]).max() < 2e-5
    if "dq" in case:
        dq, dk, dv, _ = orc.attention_bwd(do, q, k, v, None, None, None, m["causal"], m["window"], m["softcap"], m["alibi"])
        for got, ref in ((dq, case["dq"]), (dk, case["dk"]), (dv, case["dv"])):
            assert np.abs(got - ref).max() < 5e-5 * max(1.0, np.abs(ref).max())
    # LSE is never pinned by the reference tests; check it against a direct logsumexp here
    g = m["H"] // m["Hk"]
    scale = m["D"] ** -0.5
    if m["softcap"] == 0.0 and m["alibi"] is None:
        _, wl, wr = orc.normalize_window(m["Sq"], m["Sk"], m["causal"], *m["window"])
        vis = orc.visible_mask(m["Sq"], m["Sk"], wl, wr)
        s = np.einsum("bqhd,bkhd->bhqk", q, np.repeat(k, g, axis=2)) * scale
        s = np.where(vis[None, None], s, -np.inf)
        with np.errstate(divide="ignore"):
            ref = np.log(np.exp(s - s.max(-1, keepdims=True).clip(-1e300)).sum(-1)) + s.max(-1).clip(-1e300)
        live = vis.any(-1)
        assert np.allclose(lse[:, :, live], ref[:, :, live], atol=1e-9)
        assert np.all(np.isposinf(lse[:, :, ~live]))


def test_documented_causal_mask_pictures():
    """flash_attn_interface.py:1176-1185: 2x5 and 5x2 bottom-right aligned causal masks."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "known_answers.npz"))
    for nm in ("mask_2x5", "mask_5x2"):
        pic = z[nm].astype(bool)
        sq, sk = pic.shape
        c, wl, wr = orc.normalize_window(sq, sk, True, -1, -1)
        assert np.array_equal(orc.visible_mask(sq, sk, wl, wr), pic), nm


def test_window_normalisation():
    assert orc.normalize_window(5, 7, False, 7, 9) == (False, -1, -1)
    assert orc.normalize_window(5, 7, True, 3, 5) == (True, 3, 0)
    assert orc.normalize_window(1, 7, True, -1, -1) == (False, -1, -1)
    assert orc.normalize_window(1, 7, True, -1, -1, has_alibi=True) == (True, -1, 0)


def test_fully_masked_rows_and_empty_keys():
    rng = np.random.default_rng(0)
    q = rng.standard_normal((1, 6, 2, 8)); k = rng.standard_normal((1, 3, 2, 8)); v = rng.standard_normal((1, 3, 2, 8))
    out, lse = orc.attention_fwd(q, k, v, causal=True)          # Sq > Sk: first 3 rows see nothing
    assert np.all(out[:, :3] == 0) and np.all(np.isposinf(lse[:, :, :3])) and np.all(np.isfinite(lse[:, :, 3:]))
    out, lse = orc.attention_fwd(q, k[:, :0], v[:, :0])          # Sk == 0 (flash_api.cpp:524-528)
    assert np.all(out == 0) and np.all(np.isposinf(lse))
    dq, dk, dv, _ = orc.attention_bwd(rng.standard_normal(q.shape), q, k, v, causal=True)
    assert np.all(dq[:, :3] == 0)


def test_varlen_equals_per_sequence_and_known_cu_seqlens():
    z = np.load(os.path.join(ROOT, "tests", "golden", "known_answers.npz"))
    rng = np.random.default_rng(1)
    for cq, ck in ((z["cu_bwd_varlen_overflow_q"], z["cu_bwd_varlen_overflow_k"]), (z["cu_seqq_zero_q"][:3] // 8, z["cu_seqq_zero_k"][:3] // 8)):
        tq, tk = int(cq[-1]), int(ck[-1])
        q = rng.standard_normal((tq, 2, 16)); k = rng.standard_normal((tk, 1, 16)); v = rng.standard_normal((tk, 1, 16))
        do = rng.standard_normal(q.shape)
        out, lse = orc.varlen_fwd(q, k, v, cq, ck, causal=True)
        dq, dk, dv, delta = orc.varlen_bwd(do, q, k, v, cq, ck, causal=True)
        assert np.all(np.isfinite(out)) and np.all(np.isfinite(dq)) and np.all(np.isfinite(dk)) and np.all(np.isfinite(dv))
        for b in range(len(cq) - 1):
            a0, a1, b0, b1 = int(cq[b]), int(cq[b + 1]), int(ck[b]), int(ck[b + 1])
            if a1 == a0:
                assert np.all(dk[b0:b1] == 0) and np.all(dv[b0:b1] == 0)  # test_flash_attn_ck.py:1522-1560
                continue
            o1, l1 = orc.attention_fwd(q[None, a0:a1], k[None, b0:b1], v[None, b0:b1], causal=True)
            assert np.array_equal(out[a0:a1], o1[0]) and np.array_equal(lse[:, a0:a1], l1[0])


def test_flop_model_matches_survey_table():
    # SURVEY.md section 8(d) / BASELINE.md section 2
    assert abs(orc.attention_flops(8, 16, 2048, 2048, 64) / 1e12 - 0.1374) < 1e-4
    assert abs(orc.attention_flops(4, 32, 4096, 4096, 128, causal=True) / 1e12 - 0.5498) < 2e-4
    assert abs(orc.attention_flops(2, 32, 8192, 8192, 128, causal=True, window=(1024, 0)) / 1e12 - 0.2579) < 2e-4


@pytest.mark.parametrize("name", ["drop_full_d64", "drop_causal_gqa_d128", "drop_local_d64"])
def test_oracle_dropout_matches_reference_golden(dropout_cases, name):
    """dropout branch of the oracle == reference attention_ref(dropout_p, dropout_mask) + autograd."""
    case = dropout_cases[name]
    B, Sq, Sk, H, Hk, D, causal, wl, wr = [int(x) for x in case["meta"]]
    pd = float(case["p"][0])
    q, k, v, do = _inputs(case)
    out, _ = orc.attention_fwd(q, k, v, None, bool(causal), (wl, wr), 0.0, None, pd, case["keep"])
    assert np.abs(out - case["out"]).max() < 2e-5
    dq, dk, dv, _ = orc.attention_bwd(do, q, k, v, None, None, None, bool(causal), (wl, wr), 0.0, None, pd, case["keep"])
    for got, ref in ((dq, case["dq"]), (dk, case["dk"]), (dv, case["dv"])):
        assert np.abs(got - ref).max() < 5e-5 * max(1.0, np.abs(ref).max())


def test_baseline_config1_oracle_vs_reference_cpu_sdpa():
    """BASELINE config 1 (B=2 H=4 S=128 D=64 fp32 non-causal, CPU): the oracle against the reference CPU SDPA
    (torch F.scaled_dot_product_attention) and a naive softmax(QK^T)V; plus the causal / GQA variants SDPA offers
    (SDPA's is_causal is top-left aligned, identical to flash's bottom-right only when Sq == Sk)."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    B, H, S, D = 2, 4, 128, 64
    q, k, v = (torch.randn(B, S, H, D, generator=g) for _ in range(3))
    out, lse = orc.attention_fwd(q.numpy(), k.numpy(), v.numpy())
    sdpa = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2)
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) * D ** -0.5
    naive = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), v)
    assert np.abs(out - sdpa.numpy()).max() < 1e-5 and np.abs(out - naive.numpy()).max() < 1e-5  # fp32 roundoff of the references
    assert np.abs(lse - torch.logsumexp(s, -1).numpy()).max() < 1e-5
    out_c, _ = orc.attention_fwd(q.numpy(), k.numpy(), v.numpy(), None, True)
    sdpa_c = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), is_causal=True).transpose(1, 2)
    assert np.abs(out_c - sdpa_c.numpy()).max() < 1e-5
    kg, vg = k[:, :, :2], v[:, :, :2]
    out_g, _ = orc.attention_fwd(q.numpy(), kg.numpy(), vg.numpy())
    sdpa_g = F.scaled_dot_product_attention(q.transpose(1, 2), kg.transpose(1, 2), vg.transpose(1, 2), enable_gqa=True).transpose(1, 2)
    assert np.abs(out_g - sdpa_g.numpy()).max() < 1e-5
