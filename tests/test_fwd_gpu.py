"""GPU parity tests of the forward path (through the C ABI): HIP kernels vs oracle / golden vectors.

Tolerance rule is the reference's own (tests/test_flash_attn.py:1121,1556-1560): the fused kernel's
max error against an fp32 reference must be at most twice the error of a same-dtype plain-PyTorch
implementation (+1e-5 absolute floor); LSE (not pinned by the reference) is checked to 2e-3.
"""
import math

import numpy as np
import pytest
import torch

from tests._util import attention_torch, case_meta, golden_inputs, max_abs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from flash_attn_amd import backend
    return backend


def _fwd(be, q, k, v, causal=False, window=(-1, -1), softcap=0.0, alibi=None, scale=None):
    scale = q.shape[-1] ** -0.5 if scale is None else scale
    out, lse, _, _ = be.fwd(q, k, v, None, alibi, 0.0, scale, causal, window[0], window[1], softcap, False, None)
    return out, lse


GOLDEN_NATIVE = ["mha_full_d64", "mha_causal_d128", "gqa_causal_sq_gt_sk", "mqa_local_d128",
                 "gqa_causal_window_d128", "local_left_only_d64", "local_right_only_d64", "tiny_sq1",
                 "softcap_d64", "alibi_d64", "d32_full", "d96_causal", "d256_causal"]


@pytest.mark.parametrize("name", GOLDEN_NATIVE)
def test_forward_matches_reference_golden(be, golden_cases, name):
    case = golden_cases[name]
    m = case_meta(case)
    q, k, v, _ = golden_inputs(case, "cuda")
    alibi = None if m["alibi"] is None else torch.from_numpy(np.asarray(m["alibi"], dtype=np.float32)).cuda()
    out, lse = _fwd(be, q, k, v, m["causal"], m["window"], m["softcap"], alibi)
    ref = torch.from_numpy(case["out"]).cuda()
    # bf16 output quantisation (2^-9 relative) + bf16 P rounding: absolute bound scaled by |out|max
    tol = 1.2e-2 * max(1.0, float(ref.abs().max()))
    assert max_abs(out.float(), ref) < tol, (name, max_abs(out.float(), ref))
    from oracle import attention_oracle as orc
    _, lse_ref = orc.attention_fwd(q.float().cpu(), k.float().cpu(), v.float().cpu(), None, m["causal"], m["window"],
                                   m["softcap"], m["alibi"])
    lse_ref = torch.from_numpy(lse_ref).cuda()
    fin = torch.isfinite(lse_ref)
    assert torch.equal(torch.isposinf(lse), ~fin)
    assert max_abs(lse[fin], lse_ref[fin].float()) < 2e-3


def test_head_dim_above_256_fails_loudly(be):
    q = torch.zeros(1, 16, 2, 264, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="256"):
        _fwd(be, q, q, q, True)


SEQ = [(113, 203), (128, 217), (113, 211), (108, 256), (256, 512), (512, 256), (1024, 1024), (1023, 1024), (1024, 1023), (2048, 2048)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mha_type", ["mha", "gqa", "mqa"])
@pytest.mark.parametrize("mode", ["full", "causal", "local"])
@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("sq,sk", SEQ)
def test_forward_vs_fp32_reference(be, sq, sk, d, mode, mha_type, dtype):
    torch.manual_seed(0)
    B, H = 2, 6
    Hk = {"mha": 6, "gqa": 2, "mqa": 1}[mha_type]
    q = torch.randn(B, sq, H, d, device="cuda", dtype=dtype)
    k = torch.randn(B, sk, Hk, d, device="cuda", dtype=dtype)
    v = torch.randn(B, sk, Hk, d, device="cuda", dtype=dtype)
    causal = mode == "causal"
    window = (-1, -1)
    if mode == "local":
        g = torch.Generator().manual_seed(sq * 7 + sk)
        window = tuple(int(x) for x in torch.randint(0, sk, (2,), generator=g))
    out, lse = _fwd(be, q, k, v, causal, window)
    ref, lse_ref = attention_torch(q, k, v, causal, window, upcast=True)
    ref32 = attention_torch(q.float(), k.float(), v.float(), causal, window, upcast=True)[0]
    pt, _ = attention_torch(q, k, v, causal, window, upcast=False, reorder=True)
    err = max_abs(out.float(), ref32)
    err_pt = max_abs(pt.float(), ref32)
    assert err <= 2 * err_pt + 1e-5, (err, err_pt)
    fin = torch.isfinite(lse_ref)
    assert torch.equal(torch.isposinf(lse), ~fin)
    assert max_abs(lse[fin], lse_ref[fin]) < 2e-3
    assert not torch.isnan(out).any()


def test_strided_qkv_packed_views(be):
    """Packed-QKV callers pass views qkv[:, :, i] (flash_attn_interface.py:479,526-528)."""
    torch.manual_seed(1)
    qkv = torch.randn(2, 300, 3, 4, 128, device="cuda", dtype=torch.bfloat16)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    out, lse = _fwd(be, q, k, v, True)
    out2, lse2 = _fwd(be, q.contiguous(), k.contiguous(), v.contiguous(), True)
    assert torch.equal(out, out2) and torch.equal(lse, lse2)


@pytest.mark.parametrize("nw", ["4", "8", "16", "34", "38"])
@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("causal", [False, True])
def test_varlen_equals_per_sequence_bit_exact(be, knobs, d, causal, nw):
    """cu_seqlens indexing check: with the schedule pinned (the default heuristic picks it from max_seqlen, which
    differs between a packed batch and its single sequences), a varlen call is bit-identical to per-sequence calls."""
    knobs.set("FA_FWD_NW", nw)
    torch.manual_seed(2)
    lens_q = [0, 76, 34, 146, 1, 300, 257]
    lens_k = [5, 76, 1, 300, 77, 300, 255]
    H, Hk = 4, 2
    cu_q = torch.tensor([0] + list(np.cumsum(lens_q)), dtype=torch.int32, device="cuda")
    cu_k = torch.tensor([0] + list(np.cumsum(lens_k)), dtype=torch.int32, device="cuda")
    q = torch.randn(int(cu_q[-1]), H, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(int(cu_k[-1]), Hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(int(cu_k[-1]), Hk, d, device="cuda", dtype=torch.bfloat16)
    scale = d ** -0.5
    out, lse, _, _ = be.varlen_fwd(q, k, v, None, cu_q, cu_k, None, None, None, None, max(lens_q), max(lens_k), 0.0, scale,
                                   False, causal, -1, -1, 0.0, False, None)
    for b in range(len(lens_q)):
        a0, a1, b0, b1 = int(cu_q[b]), int(cu_q[b + 1]), int(cu_k[b]), int(cu_k[b + 1])
        if a1 == a0:
            continue
        o1, l1 = _fwd(be, q[None, a0:a1], k[None, b0:b1], v[None, b0:b1], causal)
        assert torch.equal(out[a0:a1], o1[0]), b
        assert torch.equal(lse[:, a0:a1], l1[0]), b


def test_empty_keys_and_validation(be):
    q = torch.randn(1, 8, 2, 64, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(1, 0, 2, 64, device="cuda", dtype=torch.bfloat16)
    out, lse = _fwd(be, q, k, k.clone())
    assert torch.all(out == 0) and torch.all(torch.isposinf(lse))
    with pytest.raises(RuntimeError):
        be.fwd(q, q, q, None, None, 0.0, 0.125, False, -1, -1, 0.0, False, torch.Generator())
    with pytest.raises(RuntimeError):
        be.fwd(q.float(), q.float(), q.float(), None, None, 0.0, 0.125, False, -1, -1, 0.0, False, None)


def test_run_to_run_bitwise_deterministic(be):
    torch.manual_seed(3)
    q = torch.randn(4, 777, 8, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(4, 901, 2, 128, device="cuda", dtype=torch.bfloat16)
    v = torch.randn_like(k)
    o0, l0 = _fwd(be, q, k, v, True)
    for _ in range(20):
        o, l = _fwd(be, q, k, v, True)
        assert torch.equal(o, o0) and torch.equal(l, l0)


@pytest.mark.parametrize("thr", ["0", "8"])
@pytest.mark.parametrize("nw", ["4", "8", "16"])
def test_rescale_branch_is_forced_and_exact(be, knobs, thr, nw):
    """The deferred-rescale branch is rare on random data: force it.  One key per 64-key tile is spiked against
    one query row so that the row's maximum jumps by far more than any threshold at a chosen tile, for every
    schedule and for threshold 0 (reference rule) and 8 (default); checked against the fp64 oracle on the
    FULL tensor (a wrong rescale corrupts whole rows, not the spiked element only)."""
    from oracle import attention_oracle as orc
    knobs.set("FA_RESCALE_THR", thr)
    knobs.set("FA_FWD_NW", nw)
    torch.manual_seed(11)
    B, S, H, D = 1, 1024, 2, 128
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16)
    for i, row in enumerate(range(5, S, 97)):  # rows spread over all waves / halves
        tile = (3 * i + 2) % (S // 64)
        key = tile * 64 + (7 * i) % 64
        k[0, key, :, :] = (q[0, row, :, :].float() * (1.0 + 0.25 * i)).to(torch.bfloat16)  # q.k ~ |q|^2 >> others
    out, lse = _fwd(be, q, k, v)
    ref, lse_ref = orc.attention_fwd(q, k, v)
    ref = torch.from_numpy(ref).cuda()
    assert max_abs(out.float(), ref) < 2e-2
    assert max_abs(lse, torch.from_numpy(lse_ref).cuda().float()) < 2e-3
    # same inputs, threshold 0 vs this threshold agree to rounding
    knobs.set("FA_RESCALE_THR", "0")
    out0, lse0 = _fwd(be, q, k, v)
    assert max_abs(out.float(), out0.float()) < 1.6e-2 and max_abs(lse, lse0) < 1e-4


def test_threshold_accuracy_budget(be, knobs):
    """Error against the fp64 oracle with the default threshold stays within 1.5x of threshold 0."""
    from oracle import attention_oracle as orc
    torch.manual_seed(12)
    q = torch.randn(2, 777, 4, 128, device="cuda", dtype=torch.bfloat16) * 2
    k = torch.randn(2, 1111, 4, 128, device="cuda", dtype=torch.bfloat16) * 2
    v = torch.randn_like(k) * 4
    ref = torch.from_numpy(orc.attention_fwd(q, k, v, causal=True)[0]).cuda()
    errs = {}
    for thr in ("0", "8"):
        knobs.set("FA_RESCALE_THR", thr)
        errs[thr] = max_abs(_fwd(be, q, k, v, True)[0].float(), ref)
    assert errs["8"] <= 1.5 * errs["0"] + 1e-3, errs


def test_varlen_default_schedule_matches_per_sequence_within_rounding(be):
    """Same check with the default schedule heuristic (which may pick different schedules for the packed call and
    the single sequences): equal up to bf16 output rounding."""
    torch.manual_seed(21)
    lens = [513, 77, 1200, 300]
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    q = torch.randn(sum(lens), 4, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(sum(lens), 2, 128, device="cuda", dtype=torch.bfloat16)
    v = torch.randn_like(k)
    out, lse, _, _ = be.varlen_fwd(q, k, v, None, cu, cu, None, None, None, None, max(lens), max(lens), 0.0, 128 ** -0.5, False, True,
                                   -1, -1, 0.0, False, None)
    for b in range(len(lens)):
        s0, s1 = int(cu[b]), int(cu[b + 1])
        o1, l1 = _fwd(be, q[None, s0:s1], k[None, s0:s1], v[None, s0:s1], True)
        assert max_abs(out[s0:s1].float(), o1[0].float()) < 1.6e-2
        assert max_abs(lse[:, s0:s1], l1[0]) < 1e-4


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("sq,h,hk", [(16, 8, 1), (7, 12, 4), (32, 8, 2), (64, 8, 4)])
def test_fwd_packs_grouped_heads_of_short_query_chunks(knobs, sq, h, hk, causal):
    """fa_fwd (no KV cache) packs the g = H / Hk query heads of a KV group into the rows of one block when g * Sq <= 128 (fa_api.cpp
    pack_group; FA3 PackGQA, hopper/pack_gqa.h): bit for bit the unpacked call (same kernel, same arithmetic per row), and within tolerance of
    the fp64 oracle."""
    from flash_attn_amd import backend as be
    from oracle import attention_oracle as orc
    torch.manual_seed(3)
    B, Sk, d = 2, 1500, 128
    g = h // hk
    q = torch.randn(B, sq, h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, Sk, hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn_like(k)
    run = lambda: be.fwd(q, k, v, None, None, 0.0, d ** -0.5, causal, -1, -1, 0.0, False, None)[:2]
    out, lse = run()
    assert be.last_schedule()["fwd_pack"] == g, be.last_schedule()
    knobs.set("FA_PACK_GQA", 0)
    out0, lse0 = run()
    assert be.last_schedule()["fwd_pack"] == 1
    knobs.unset("FA_PACK_GQA")
    assert torch.equal(out, out0) and torch.equal(lse, lse0)
    ref, lse_ref = orc.attention_fwd(q, k, v, d ** -0.5, causal)
    assert float((out.float().cpu() - torch.from_numpy(ref).float()).abs().max()) < 2e-2
    assert float((lse.cpu() - torch.from_numpy(lse_ref).float()).abs().max()) < 2e-3


@pytest.mark.parametrize("window", [(-1, -1), (300, 0)])
@pytest.mark.parametrize("per_batch,dtype", [(False, torch.bfloat16), (True, torch.bfloat16), (False, torch.float16)])
@pytest.mark.parametrize("B,Sq,Sk,H,Hk,D", [(2, 1024, 1024, 8, 8, 128), (1, 1500, 1700, 6, 2, 128), (2, 2048, 2048, 4, 4, 64)])
def test_w64_causal_alibi(be, knobs, B, Sq, Sk, H, Hk, D, per_batch, dtype, window):
    """Causal ALiBi on the 64-rows-per-wave forward (fa_fwd_w64_kernel<.., alibi>: the bias rides in the score chains' C operand, the key tiles are
    walked downwards from the diagonal; mask rewrites and the rescale keep the bias exact): against the fp64 oracle, and against the lock-step kernel
    that served ALiBi before."""
    from oracle import attention_oracle as orc
    torch.manual_seed(B * Sq + D)
    q = torch.randn(B, Sq, H, D, device="cuda", dtype=dtype)
    k = torch.randn(B, Sk, Hk, D, device="cuda", dtype=dtype)
    v = torch.randn_like(k)
    slopes = (torch.rand((B, H) if per_batch else (H,), device="cuda") * 0.6 + 0.004).float()
    run = lambda: be.fwd(q, k, v, None, slopes, 0.0, D ** -0.5, True, window[0], window[1], 0.0, False, None)[:2]
    knobs.set("FA_FWD_NW", "64")
    out, lse = run()
    s = be.last_schedule()
    assert s["fwd_kernel"] == 3 and "alibi" in s["name"], s
    knobs.set("FA_FWD_NW", "8")
    out8, lse8 = run()
    assert be.last_schedule()["fwd_kernel"] == 1
    knobs.unset("FA_FWD_NW")
    ref, lse_ref = orc.attention_fwd(q, k, v, D ** -0.5, True, window, 0.0, slopes.cpu().numpy())
    ref, lse_ref = torch.from_numpy(ref).float(), torch.from_numpy(lse_ref).float()
    assert torch.isfinite(out.float()).all()
    e64, e8 = float((out.float().cpu() - ref).abs().max()), float((out8.float().cpu() - ref).abs().max())
    fin = torch.isfinite(lse_ref)
    el = float((lse.cpu() - lse_ref)[fin].abs().max())
    assert e64 < max(2 * e8, 1.2e-2 if dtype == torch.bfloat16 else 4e-3) and el < 8e-3 and torch.equal(torch.isinf(lse.cpu()), ~fin), (e64, e8, el)


def test_w64_causal_alibi_128k_keys(be, knobs):
    """The ALiBi bias at very long sequences: the key distance is formed in integers before it meets the slope (fa_fwd_w64.hip: arel).  Round 3 added two
    fp32 terms of size slope * 128k that cancel -- ~1e-2 log2 units of error on the keys next to the diagonal at this length.  fp16, so that the kernel's
    own rounding of q * scale * log2e (2^-12 relative) leaves room to see it; sampled query rows against fp64 on the GPU."""
    torch.manual_seed(5)
    S, H, D = 131072, 2, 128
    q = torch.randn(1, S, H, D, device="cuda", dtype=torch.float16)
    k, v = torch.randn_like(q), torch.randn_like(q)
    slopes = torch.tensor([0.5, 2.0 ** -6], device="cuda", dtype=torch.float32)
    knobs.set("FA_FWD_NW", "64")
    out, lse = be.fwd(q, k, v, None, slopes, 0.0, D ** -0.5, True, -1, -1, 0.0, False, None)[:2]
    s = be.last_schedule()
    knobs.unset("FA_FWD_NW")
    assert s["fwd_kernel"] == 3 and "alibi" in s["name"], s
    rows = torch.cat([torch.arange(0, 64), torch.arange(65500, 65600), torch.arange(S - 200, S)]).cuda()
    j = torch.arange(S, device="cuda")
    for h in range(H):
        sc = (q[0, rows, h].double() * D ** -0.5) @ k[0, :, h].double().T            # (rows, S)
        sc = sc - slopes[h].double() * (rows[:, None] - j[None, :]).abs().double()
        sc = sc.masked_fill(j[None, :] > rows[:, None], float("-inf"))
        lse_ref = torch.logsumexp(sc, -1)
        o_ref = torch.softmax(sc, -1) @ v[0, :, h].double()
        e_l = float((lse[0, h, rows].double() - lse_ref).abs().max())
        e_o = float((out[0, rows, h].double() - o_ref).abs().max())
        assert e_l < 2e-3 and e_o < 4e-3, (h, e_l, e_o)


def test_w64_causal_alibi_varlen(be, knobs):
    """The same through the packed variable-length entry (work list, uneven sequences: a block-aligned one, one with a single row in its last block,
    one shorter than a block), judged like the fixed-length cases: against the fp64 oracle, relative to the lock-step kernel's error."""
    from oracle import attention_oracle as orc
    torch.manual_seed(4)
    H, D = 8, 128
    lens = [1300, 70, 2048, 513, 900]
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    tot = int(cu[-1])
    q = torch.randn(tot, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
    slopes = (torch.rand(H, device="cuda") * 0.3 + 0.02).float()
    run = lambda: be.varlen_fwd(q, k, v, None, cu, cu, None, None, None, slopes, max(lens), max(lens), 0.0, D ** -0.5, False, True, -1, -1, 0.0, False, None)[:2]
    knobs.set("FA_FWD_NW", "64")
    out, lse = run()
    s = be.last_schedule()
    knobs.set("FA_FWD_NW", "8")
    out8, lse8 = run()
    s8 = be.last_schedule()
    knobs.unset("FA_FWD_NW")
    assert s["fwd_kernel"] == 3 and "alibi" in s["name"], s
    assert s8["fwd_kernel"] == 1, s8
    assert torch.isfinite(out.float()).all()
    errs = []
    for i, n in enumerate(lens):
        a, b_ = int(cu[i]), int(cu[i + 1])
        ref, lse_ref = orc.attention_fwd(q[a:b_][None], k[a:b_][None], v[a:b_][None], D ** -0.5, True, (-1, -1), 0.0, slopes.cpu().numpy())
        ref, lse_ref = torch.from_numpy(ref[0]).float(), torch.from_numpy(lse_ref[0]).float()
        errs.append((float((out[a:b_].float().cpu() - ref).abs().max()), float((out8[a:b_].float().cpu() - ref).abs().max()),
                     float((lse[:, a:b_].cpu() - lse_ref).abs().max()), float((lse8[:, a:b_].cpu() - lse_ref).abs().max())))
    assert all(e64 < max(2 * e8, 1.2e-2) and l64 < max(2 * l8, 8e-3) for e64, e8, l64, l8 in errs), errs


@pytest.mark.parametrize("mask", [(False, -1, -1), (True, -1, -1), (False, 300, 0), (False, 100, 200)], ids=["full", "causal", "local_causal", "local"])
@pytest.mark.parametrize("softcap,dtype,scale_in", [(15.0, torch.bfloat16, 1.0), (50.0, torch.bfloat16, 1.0), (30.0, torch.float16, 1.0), (30.0, torch.bfloat16, 6.0)])
@pytest.mark.parametrize("B,Sq,Sk,H,Hk,D", [(2, 1024, 1024, 4, 4, 128), (1, 1500, 1700, 6, 2, 128), (1, 1700, 700, 2, 2, 128), (2, 2048, 2048, 4, 2, 64), (1, 333, 2100, 2, 1, 64)])
def test_w64_softcap(be, knobs, B, Sq, Sk, H, Hk, D, softcap, dtype, scale_in, mask):
    """Softcap on the 64-rows-per-wave forward (fa_fwd_w64_kernel<.., softcap>, round 5; reference: flash_fwd_kernel.h:357-368 + utils.h:395-409).  The chains start
    from C = 0 and deliver 2*log2e * score*scale/softcap, the cap is applied per score on the vector ALU (three staged gaps), masked scores keep their -inf through
    tanh's saturation, the row maximum goes through the cap once per row and step.  Against the fp64 oracle and against the lock-step kernel that served softcap
    before.  scale_in = 6 is a stress case (capped scores of +-20 log2 units, the running maximum moves through the whole row, the cap saturates): there the once-rounded
    Q of this schedule shows -- 0.031-0.037 against 0.016 for the lock-step kernel's fp32 scaling, LSE 0.02-0.03, tools/softcap_fwd_diag.py -- and the bound is 3x
    instead of the reference's 2x (FA_STRICT=1 selects the exact kernels).  Sq > Sk under a causal mask: rows that see no key (LSE = +inf, zeros)."""
    from oracle import attention_oracle as orc
    causal, wl, wr = mask
    torch.manual_seed(B * Sq + D + int(softcap))
    q = torch.randn(B, Sq, H, D, device="cuda", dtype=dtype) * scale_in
    k = torch.randn(B, Sk, Hk, D, device="cuda", dtype=dtype)
    v = torch.randn_like(k)
    run = lambda: be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, wl, wr, softcap, False, None)[:2]
    knobs.set("FA_FWD_NW", "64")
    out, lse = run()
    s = be.last_schedule()
    assert s["fwd_kernel"] == 3 and "softcap" in s["name"], s
    out_b, lse_b = run()
    assert torch.equal(out, out_b) and torch.equal(lse, lse_b), "run-to-run"
    knobs.set("FA_FWD_NW", "8")
    out8, lse8 = run()
    assert be.last_schedule()["fwd_kernel"] == 1
    knobs.unset("FA_FWD_NW")
    ref, lse_ref = orc.attention_fwd(q, k, v, D ** -0.5, causal, (wl, wr), softcap, None)
    ref, lse_ref = torch.from_numpy(ref).float(), torch.from_numpy(lse_ref).float()
    assert torch.isfinite(out.float()).all()
    e64, e8 = float((out.float().cpu() - ref).abs().max()), float((out8.float().cpu() - ref).abs().max())
    fin = torch.isfinite(lse_ref)
    el = float((lse.cpu() - lse_ref)[fin].abs().max()) if fin.any() else 0.0
    # (the scores reach the cap through a Q that was scaled and rounded once, as on the plain kernel: the LSE bound of tests/test_baseline_configs_gpu.py lse_tolerance)
    assert e64 < max((2 if scale_in == 1.0 else 3) * e8, 1.2e-2 if dtype == torch.bfloat16 else 4e-3), (e64, e8)
    assert el < (8e-3 if dtype == torch.bfloat16 else 2e-3) * max(1.0, scale_in), el
    assert torch.equal(torch.isinf(lse.cpu()), ~fin)


def test_w64_softcap_varlen_and_default_dispatch(be, knobs):
    """The softcap variant through the packed entry point (work list or dense grid) against the sequences one by one, and the default's choice: softcap takes the
    64-rows-per-wave kernel where plain attention does; softcap together with ALiBi or dropout stays on the lock-step kernel."""
    import itertools
    torch.manual_seed(9)
    lens_q = [700, 33, 1500, 256, 1, 900, 0, 300]
    lens_k = [700, 65, 1500, 300, 77, 513, 5, 2048]
    H, Hk, D, cap = 4, 2, 128, 20.0
    cu_q = torch.tensor([0] + list(itertools.accumulate(lens_q)), dtype=torch.int32, device="cuda")
    cu_k = torch.tensor([0] + list(itertools.accumulate(lens_k)), dtype=torch.int32, device="cuda")
    q = torch.randn(sum(lens_q), H, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(sum(lens_k), Hk, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn_like(k)
    knobs.set("FA_FWD_NW", "64")
    out, lse = be.varlen_fwd(q, k, v, None, cu_q, cu_k, None, None, None, None, max(lens_q), max(lens_k), 0.0, D ** -0.5, False, True, -1, -1, cap, False, None)[:2]
    s = be.last_schedule()
    assert s["fwd_kernel"] == 3 and "softcap" in s["name"], s
    for b in range(len(lens_q)):
        a0, a1, b0, b1 = int(cu_q[b]), int(cu_q[b + 1]), int(cu_k[b]), int(cu_k[b + 1])
        if a1 == a0:
            continue
        o1, l1 = be.fwd(q[None, a0:a1], k[None, b0:b1], v[None, b0:b1], None, None, 0.0, D ** -0.5, True, -1, -1, cap, False, None)[:2]
        if be.last_schedule()["fwd_kernel"] == 3:   # the same kernel on the same rows: bit for bit
            assert torch.equal(out[a0:a1], o1[0]) and torch.equal(lse[:, a0:a1], l1[0]), b
        else:                                        # (a very short sequence on its own may take another schedule: head packing, the 4-wave kernel)
            assert float((out[a0:a1].float() - o1[0].float()).abs().max()) < 2e-2 and float((lse[:, a0:a1] - l1[0]).abs().max()) < 8e-3, b
    knobs.unset("FA_FWD_NW")
    q4 = torch.randn(2, 4096, 8, 128, device="cuda", dtype=torch.bfloat16)
    k4, v4 = torch.randn_like(q4), torch.randn_like(q4)
    be.fwd(q4, k4, v4, None, None, 0.0, 128 ** -0.5, True, -1, -1, 30.0, False, None)
    s = be.last_schedule()
    assert s["fwd_kernel"] == 3 and "softcap" in s["name"], s
    be.fwd(q4, k4, v4, None, torch.full((8,), 0.1, device="cuda"), 0.0, 128 ** -0.5, True, -1, -1, 30.0, False, None)
    assert be.last_schedule()["fwd_kernel"] == 1
    be.fwd(q4, k4, v4, None, None, 0.1, 128 ** -0.5, True, -1, -1, 30.0, False, None)
    assert be.last_schedule()["fwd_kernel"] == 1
