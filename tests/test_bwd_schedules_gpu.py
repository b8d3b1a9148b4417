"""Backward schedules added in round 2 (reference: compute_dq_dk_dv_1colblock, csrc/flash_attn/src/flash_bwd_kernel.h:80-795):
  * fa_bwd_dq_w64_kernel  (FA_BWD_DQ_NW=64, the default from 2k keys at head dim 128): 64 query rows per wave;
(The dS-spill path and the 64-keys-per-wave dK/dV kernel are experiments: their tests live in experiments/test_bwd_schedules_gpu.py.)
It must reproduce the 32-rows-per-wave recomputing kernels: same arithmetic per element, so dq agrees to rounding of the
fp32 accumulation order, and dk / dv -- produced by the same kernel -- bit for bit.  An fp32 PyTorch reference bounds the
error of each in absolute terms (tolerance: twice the error of the established kernel, floor 1e-2 bf16 / 2e-3 fp16)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from flash_attn_amd import backend
    return backend


def ref_grads(q, k, v, do, causal, wl, wr, upcast=True):
    """Gradients of plain PyTorch attention: fp32 math (the reference point), or -- upcast=False -- math in the input dtype: the 'PyTorch baseline' whose error
    calibrates the tolerance in the reference's own tests (tests/test_flash_attn.py: <= 3x that error for gradients)."""
    cast = (lambda x: x.float()) if upcast else (lambda x: x)
    qf, kf, vf = [cast(x).transpose(1, 2).detach().requires_grad_(True) for x in (q, k, v)]
    g = qf.shape[1] // kf.shape[1]
    s = qf @ kf.repeat_interleave(g, 1).transpose(-1, -2) * q.shape[-1] ** -0.5
    Sq, Sk = s.shape[-2:]
    i = torch.arange(Sq, device=q.device)[:, None] + (Sk - Sq)
    j = torch.arange(Sk, device=q.device)[None]
    m = torch.zeros(Sq, Sk, dtype=torch.bool, device=q.device)
    if causal:
        wr = 0
    if wr >= 0:
        m |= j > i + wr
    if wl >= 0:
        m |= j < i - wl
    p = torch.softmax(s.masked_fill(m, float("-inf")), -1).nan_to_num(0.0)
    (p @ vf.repeat_interleave(g, 1)).backward(cast(do).transpose(1, 2))
    return [x.grad.float().transpose(1, 2) for x in (qf, kf, vf)]


def run_bwd(be, q, k, v, do, causal, wl=-1, wr=-1, **feat):
    D = q.shape[-1]
    torch.cuda.manual_seed(11)
    out, lse, _, rng = be.fwd(q, k, v, None, feat.get("alibi"), feat.get("p_drop", 0.0), D ** -0.5, causal, wl, wr, feat.get("softcap", 0.0),
                              False, None)
    dq, dk, dv, delta = be.bwd(do, q, k, v, out, lse, None, None, None, feat.get("alibi"), feat.get("p_drop", 0.0), D ** -0.5, causal, wl, wr,
                               feat.get("softcap", 0.0), False, None, rng)
    return dq, dk, dv, be.last_schedule(), delta


SHAPES = [  # B, Sq, Sk, H, Hk, causal, wl, wr
    (1, 256, 256, 2, 2, False, -1, -1), (1, 512, 512, 2, 1, True, -1, -1), (2, 1024, 1024, 4, 4, True, -1, -1),
    (1, 300, 333, 2, 2, False, -1, -1), (1, 300, 333, 2, 2, True, -1, -1), (1, 777, 1000, 3, 1, False, 100, 50),
    (1, 64, 64, 1, 1, True, -1, -1), (1, 1, 500, 2, 2, False, -1, -1), (2, 2048, 2048, 4, 2, True, -1, -1),
    (1, 1000, 200, 2, 2, True, -1, -1), (1, 513, 1025, 2, 2, False, 64, 0), (1, 33, 97, 2, 2, False, -1, -1),
    (1, 1025, 1025, 1, 1, True, -1, -1), (1, 200, 1000, 4, 1, True, -1, -1), (1, 2000, 2000, 1, 1, False, 0, 0),
    (1, 640, 640, 1, 1, False, 300, -1), (1, 640, 640, 1, 1, False, -1, 300),
]


@pytest.mark.parametrize("d", [128, 64])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "B%d_Sq%d_Sk%d_H%d_%d_c%d_w%d_%d" % s)
def test_dq_w64_kernel_matches_recomputing_kernel_and_fp32(be, knobs, dtype, shape, d):
    B, Sq, Sk, H, Hk, causal, wl, wr = shape
    torch.manual_seed(0)
    q = torch.randn(B, Sq, H, d, device="cuda", dtype=dtype)
    k = torch.randn(B, Sk, Hk, d, device="cuda", dtype=dtype)
    v = torch.randn_like(k)
    do = torch.randn_like(q)
    knobs.set("FA_BWD_DQ_NW", 4)
    a = run_bwd(be, q, k, v, do, causal, wl, wr)
    knobs.set("FA_BWD_DQ_NW", 64)
    w = run_bwd(be, q, k, v, do, causal, wl, wr)
    assert a[3]["bwd_dq_nw"] == 4 and w[3]["bwd_dq_nw"] == 64, (a[3], w[3])
    assert torch.equal(a[2], w[2])                                      # dv: same kernel, and dV does not see softmax_d
    knobs.set("FA_BWD_FUSE_DELTA", 0)                                   # the delta pre-pass for both: then dk is the same kernel on the same inputs too
    w3 = run_bwd(be, q, k, v, do, causal, wl, wr)
    knobs.unset("FA_BWD_FUSE_DELTA")
    assert w3[3]["bwd_dq_nw"] == 64 and torch.equal(a[1], w3[1]) and torch.equal(a[2], w3[2])
    r = ref_grads(q, k, v, do, causal, wl, wr)
    # round 4: the 64-rows-per-wave dQ kernel computes softmax_d = rowsum(dO * O) of its rows itself (another summation order than the pre-pass): dk and dq
    # move by rounding, judged against the fp32 reference like dq below
    for i in (0, 1):
        e_f, e_3 = float((w[i].float() - r[i]).abs().max()), float((w3[i].float() - r[i]).abs().max())
        assert e_f <= max(1.5 * e_3, 1e-2 if dtype == torch.bfloat16 else 2e-3), (i, e_f, e_3)
    # softmax_d as the fused prologue writes it (every row of every block, the caller gets it back) against the pre-pass
    assert float((w[4] - a[4]).abs().max()) <= 1e-4 * max(1.0, float(a[4].abs().max())), float((w[4] - a[4]).abs().max())
    floor = 1e-2 if dtype == torch.bfloat16 else 2e-3
    e4 = float((a[0].float() - r[0]).abs().max())
    e64 = float((w[0].float() - r[0]).abs().max())
    assert torch.isfinite(w[0].float()).all()
    assert e64 <= max(2 * e4, floor), (e64, e4)


def test_dq_w64_default_dispatch_and_fallbacks(be, knobs):
    """Heuristic (fa_api.cpp bwd_dq_schedule): 64 rows per wave from 2k keys at head dim 128 for plain attention and (round 5) for ONE of softcap / dropout /
    causal ALiBi; everything else on the 32-rows-per-wave kernels.  A forced 64 falls back to the 8-wave kernel (same 256-row blocks) where the schedule
    does not apply (a product of features) -- and still computes the right thing."""
    knobs.unset("FA_BWD_DQ_NW")
    torch.manual_seed(1)
    q = torch.randn(1, 2048, 2, 128, device="cuda", dtype=torch.bfloat16)
    k, v, do = torch.randn_like(q), torch.randn_like(q), torch.randn_like(q)
    knobs.set("FA_BWD_MODE", -1)   # (the recomputing pair's own dispatch: round 6 gives plain causal attention at 512 - 2k rows to the fused launch by default, from 32 units on)
    assert run_bwd(be, q, k, v, do, True)[3]["bwd_dq_nw"] == 64
    assert run_bwd(be, q[:, :1024], k[:, :1024], v[:, :1024], do[:, :1024], True)[3]["bwd_dq_nw"] == 4
    knobs.unset("FA_BWD_MODE")
    assert run_bwd(be, q, k, v, do, True)[3]["bwd_spill"] == 0 and run_bwd(be, q, k, v, do, False)[3]["bwd_dq_nw"] == 64   # (two units: the pair)
    q16 = torch.randn(16, 2048, 2, 128, device="cuda", dtype=torch.bfloat16)
    assert run_bwd(be, q16, torch.randn_like(q16), torch.randn_like(q16), torch.randn_like(q16), True)[3]["bwd_spill"] == 3
    assert run_bwd(be, q, k, v, do, True, p_drop=0.1)[3]["bwd_dq_nw"] == 64
    assert run_bwd(be, q, k, v, do, True, softcap=20.0)[3]["bwd_dq_nw"] == 64
    assert run_bwd(be, q, k, v, do, True, p_drop=0.1, softcap=20.0)[3]["bwd_dq_nw"] == 4
    knobs.set("FA_BWD_DQ_NW", 64)
    d_forced = run_bwd(be, q, k, v, do, True, p_drop=0.1, softcap=20.0)
    assert d_forced[3]["bwd_dq_nw"] == 8
    knobs.set("FA_BWD_DQ_NW", 4)
    d_ref = run_bwd(be, q, k, v, do, True, p_drop=0.1, softcap=20.0)
    assert float((d_forced[0].float() - d_ref[0].float()).abs().max()) < 1e-2
    q96 = torch.randn(1, 2048, 2, 96, device="cuda", dtype=torch.bfloat16)   # (head dim 64 has the kernel since round 4; trimmed head dims do not)
    knobs.set("FA_BWD_DQ_NW", 64)
    assert run_bwd(be, q96, torch.randn_like(q96), torch.randn_like(q96), torch.randn_like(q96), False)[3]["bwd_dq_nw"] == 4


@pytest.mark.parametrize("causal", [False, True])
def test_dq_w64_varlen_equals_per_sequence(be, knobs, causal):
    """Packed batch through mha_varlen_bwd with the 64-rows-per-wave dQ kernel (dense grid and work list) == each sequence
    on its own."""
    knobs.set("FA_BWD_DQ_NW", 64)
    knobs.set("FA_FWD_NW", 34)
    torch.manual_seed(5)
    lens_q = [0, 76, 34, 700, 1, 300, 257]
    lens_k = [5, 76, 1, 700, 77, 300, 255]
    H, Hk, d = 4, 2, 128
    cu_q = torch.tensor([0] + list(np.cumsum(lens_q)), dtype=torch.int32, device="cuda")
    cu_k = torch.tensor([0] + list(np.cumsum(lens_k)), dtype=torch.int32, device="cuda")
    q = torch.randn(int(cu_q[-1]), H, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(int(cu_k[-1]), Hk, d, device="cuda", dtype=torch.bfloat16)
    v, do = torch.randn_like(k), torch.randn_like(q)
    sc = d ** -0.5
    out, lse, _, _ = be.varlen_fwd(q, k, v, None, cu_q, cu_k, None, None, None, None, max(lens_q), max(lens_k), 0.0, sc, False, causal,
                                   -1, -1, 0.0, False, None)
    dq, dk, dv, _ = be.varlen_bwd(do, q, k, v, out, lse, None, None, None, cu_q, cu_k, None, max(lens_q), max(lens_k), 0.0, sc, False,
                                  causal, -1, -1, 0.0, False, None, None)
    assert be.last_schedule()["bwd_dq_nw"] == 64
    for b in range(len(lens_q)):
        if lens_q[b] == 0 or lens_k[b] == 0:
            continue
        qs, ks = slice(int(cu_q[b]), int(cu_q[b + 1])), slice(int(cu_k[b]), int(cu_k[b + 1]))
        o1, l1, _, _ = be.fwd(q[qs][None], k[ks][None], v[ks][None], None, None, 0.0, sc, causal, -1, -1, 0.0, False, None)
        g = be.bwd(do[qs][None], q[qs][None], k[ks][None], v[ks][None], o1, l1, None, None, None, None, 0.0, sc, causal, -1, -1, 0.0,
                   False, None, None)
        assert torch.equal(g[0][0], dq[qs]), b
        assert torch.equal(g[1][0], dk[ks]) and torch.equal(g[2][0], dv[ks]), b


FEATS = [dict(), dict(softcap=20.0), dict(alibi=True), dict(p_drop=0.2)]


@pytest.mark.parametrize("shape", [SHAPES[1], SHAPES[5], SHAPES[8]], ids=lambda s: "B%d_Sq%d_Sk%d_H%d_%d_c%d_w%d_%d" % s)
def test_prescaled_k_variant_stays_inside_the_budget_of_the_default(be, knobs, shape):
    """Since round 4 the plain dK/dV kernel scales every score in fp32, as the reference does (flash_bwd_kernel.h:536): FEAT_EXACT.  FA_DKDV_PRESCALE=1
    opts into the variant that multiplies its K fragments by softmax_scale*log2(e) once (rounded to the input dtype) and lets the matrix pipe subtract
    LSE as well (~3 % faster).  Both must sit inside the usual error budget of the fp32 reference; the default is the yardstick, and FA_STRICT=1 must
    not change the backward any more (bit for bit: same kernel)."""
    B, Sq, Sk, H, Hk, causal, wl, wr = shape
    torch.manual_seed(0)
    q = torch.randn(B, Sq, H, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, Sk, Hk, 128, device="cuda", dtype=torch.bfloat16)
    v = torch.randn_like(k)
    do = torch.randn_like(q)
    default = run_bwd(be, q, k, v, do, causal, wl, wr)
    knobs.set("FA_DKDV_PRESCALE", 1)
    fast = run_bwd(be, q, k, v, do, causal, wl, wr)
    knobs.unset("FA_DKDV_PRESCALE")
    r = ref_grads(q, k, v, do, causal, wl, wr)
    for i in range(3):
        e_d = float((default[i].float() - r[i]).abs().max())
        e_f = float((fast[i].float() - r[i]).abs().max())
        assert e_f <= max(2 * e_d, 1e-2), (i, e_f, e_d)
    assert not torch.equal(fast[1], default[1]) or Sk < 64   # (the knob does select another kernel)


# Round 4: the fused backward (FA_BWD_MODE=3, opt-in): dK / dV and dQ = dS.K in one persistent launch, dS handed over through a workspace
# (fa_bwd.hip fa_bwd_fused_kernel; reference: the 5-contraction compute_dq_dk_dv_1colblock, flash_bwd_kernel.h:457-733).
FUSED_SHAPES = [  # B, Sq, Sk, H, Hk, causal
    (1, 256, 256, 2, 2, False), (2, 1024, 1024, 4, 4, True), (1, 300, 333, 2, 2, True), (1, 1, 500, 2, 2, False), (1, 200, 1000, 4, 1, True),
    (3, 512, 512, 32, 32, True), (2, 1024, 1024, 32, 8, True), (7, 640, 640, 6, 6, True), (1, 2048, 2048, 8, 2, False),
]


@pytest.mark.parametrize("d", [128, 64])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", FUSED_SHAPES, ids=lambda s: "B%d_Sq%d_Sk%d_H%d_%d_c%d" % s)
def test_fused_backward_matches_default_and_fp32(be, knobs, dtype, shape, d):
    B, Sq, Sk, H, Hk, causal = shape
    torch.manual_seed(0)
    q = torch.randn(B, Sq, H, d, device="cuda", dtype=dtype)
    k = torch.randn(B, Sk, Hk, d, device="cuda", dtype=dtype)
    v = torch.randn_like(k)
    do = torch.randn_like(q)
    knobs.set("FA_BWD_FUSE_DELTA", 0)   # softmax_d from the pre-pass on both sides: dK / dV come from the same arithmetic on the same inputs
    knobs.set("FA_BWD_DKDV", 8)         # ... and from the same eight-wave dK/dV body (round 5: long sequences default to the 64-keys-per-wave kernel)
    a = run_bwd(be, q, k, v, do, causal)
    knobs.set("FA_BWD_MODE", 3)
    f = run_bwd(be, q, k, v, do, causal)
    f2 = run_bwd(be, q, k, v, do, causal)
    assert a[3]["bwd_spill"] == 0 and f[3]["bwd_spill"] == 3, (a[3], f[3])
    assert torch.equal(a[1], f[1]) and torch.equal(a[2], f[2])                    # dk, dv bit for bit
    assert all(torch.equal(x, y) for x, y in zip(f[:3], f2[:3]))                  # no atomics on data: run-to-run bitwise
    r = ref_grads(q, k, v, do, causal, -1, -1)
    # dQ: the same dS (rounded to the input dtype) contracted with K in another order -- within a few ulps of the default's dQ (measured over 80 cases,
    # profiles/r04_bwd_fused.txt: <= 1e-3 fp16, <= 4e-3 bf16 up to S = 2k) and as close to the fp32 reference as the default is
    floor = 2e-2 if dtype == torch.bfloat16 else 4e-3
    e0, e3 = float((a[0].float() - r[0]).abs().max()), float((f[0].float() - r[0]).abs().max())
    assert torch.isfinite(f[0].float()).all() and e3 <= max(2 * e0, floor), (e3, e0)
    assert float((a[0].float() - f[0].float()).abs().max()) <= floor * max(1.0, float(a[0].float().abs().max())), "dq against the default path"


def test_fused_backward_declines_what_it_does_not_cover(be, knobs):
    knobs.set("FA_BWD_MODE", 3)
    torch.manual_seed(1)
    q = torch.randn(1, 640, 2, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(1, 640, 2, 128, device="cuda", dtype=torch.bfloat16)
    v, do = torch.randn_like(k), torch.randn_like(q)
    for kw in (dict(wl=300, wr=-1), dict(softcap=20.0), dict(p_drop=0.1)):   # left window, softcap, dropout: the default kernels
        wl, wr = kw.pop("wl", -1), kw.pop("wr", -1)
        f = run_bwd(be, q, k, v, do, False, wl, wr, **kw)
        assert f[3]["bwd_spill"] == 0, (kw, f[3])
    f = run_bwd(be, q[:, :200].contiguous(), k, v, do[:, :200].contiguous(), True)   # sk > sq is covered, sq > sk is not
    assert f[3]["bwd_spill"] == 3
    f = run_bwd(be, q, k[:, :200].contiguous(), v[:, :200].contiguous(), do, True)
    assert f[3]["bwd_spill"] == 0


# Round 6: the table (fa_api.cpp bwd_fused_by_table) hands plain attention at head dim 128 with Sq = Sk, >= 32 (batch, kv head) units and <= 1 GiB of packed dS to the
# fused launch BY DEFAULT: causal 512 .. 4096 rows (and from 256 rows on large grids, with or without a mask).  No knob is set here except the workspace poison: this is the call a user makes.
TABLE_SHAPES = [  # B, S, H, Hk, causal
    (16, 512, 2, 2, True), (8, 1024, 4, 4, True), (4, 2048, 8, 8, True), (1, 4096, 32, 32, True), (8, 1024, 16, 4, True), (11, 704, 3, 3, True), (32, 515, 1, 1, True),
    (128, 384, 16, 16, False),   # (without a mask only below 512 rows on large grids: units x rows >= 786432; from 512 rows the pair is level or ahead since its dQ half takes the 64-rows-per-wave kernel from 768 keys)
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", TABLE_SHAPES, ids=lambda s: "B%d_S%d_H%d_%d_c%d" % s)
def test_default_table_region_takes_the_fused_launch(be, knobs, dtype, shape):
    B, S, H, Hk, causal = shape
    torch.manual_seed(5)
    q = torch.randn(B, S, H, 128, device="cuda", dtype=dtype)
    k = torch.randn(B, S, Hk, 128, device="cuda", dtype=dtype)
    v, do = torch.randn_like(k), torch.randn_like(q)
    knobs.set("FA_DEBUG_POISON_WS", 1)   # (backend.py: the workspace arrives full of 0xFF -- a dS tile or a sync word the launch relies on without writing it shows)
    f = run_bwd(be, q, k, v, do, causal)
    f2 = run_bwd(be, q, k, v, do, causal)
    assert f[3]["bwd_spill"] == 3, f[3]
    assert all(torch.equal(x, y) for x, y in zip(f[:3], f2[:3]))   # run-to-run bitwise
    knobs.set("FA_BWD_MODE", -1)
    a = run_bwd(be, q, k, v, do, causal)
    assert a[3]["bwd_spill"] == 0, a[3]
    # the reference's rule (tests/test_flash_attn.py): <= 3x the error of the same arithmetic in the input dtype, with a small floor; and no worse than 1.5x the recomputing pair
    r, rl = ref_grads(q, k, v, do, causal, -1, -1), ref_grads(q, k, v, do, causal, -1, -1, upcast=False)
    for name, x, y, base, lo in zip(("dq", "dk", "dv"), f[:3], a[:3], r, rl):
        e, e_pair, e_pt = float((x.float() - base).abs().max()), float((y.float() - base).abs().max()), float((lo - base).abs().max())
        assert torch.isfinite(x.float()).all() and e <= 3 * e_pt + 1e-4, (name, e, e_pt)
        assert e <= 1.5 * e_pair + 1e-4, (name, e, e_pair)


def test_dS_workspace_out_of_memory_falls_back_to_the_pair(be, knobs, monkeypatch):
    """The fixed-length workspace is the 5-contraction launches' dS area -- a speed-up.  When the allocator cannot supply it (here: made to refuse) the binder passes
    none and the library runs the recomputing pair: same gradients as FA_BWD_MODE=-1, bit for bit."""
    torch.manual_seed(6)
    q = torch.randn(8, 1024, 4, 128, device="cuda", dtype=torch.bfloat16)
    k, v, do = torch.randn_like(q), torch.randn_like(q), torch.randn_like(q)
    assert run_bwd(be, q, k, v, do, True)[3]["bwd_spill"] == 3
    asked = []

    def refuse(nbytes, device):
        asked.append(nbytes)
        raise torch.OutOfMemoryError("test: no room for %d bytes" % nbytes)

    monkeypatch.setattr(be, "_alloc_workspace", refuse)
    f = run_bwd(be, q, k, v, do, True)
    assert asked and asked[0] > (1 << 20) and f[3]["bwd_spill"] == 0, (asked, f[3])
    monkeypatch.undo()
    knobs.set("FA_BWD_MODE", -1)
    a = run_bwd(be, q, k, v, do, True)
    assert all(torch.equal(x, y) for x, y in zip(f[:3], a[:3]))


def _plan_of(B, S, H, Hk, D, causal):
    import ctypes as C
    from flash_attn_amd import _cabi
    lib = _cabi.load()
    a = _cabi.FaBwdParams()
    a.b, a.h, a.h_k, a.d = B, H, Hk, D
    a.seqlen_q, a.seqlen_k, a.total_q, a.total_k = S, S, B * S, B * S
    a.dtype, a.softmax_scale, a.is_causal = 1, D ** -0.5, int(causal)
    a.window_left = a.window_right = -1
    out = (C.c_int32 * 8)()
    assert lib.fa_bwd_plan_query(C.byref(a), out, 8) == 8
    return list(out)


# Round 6 (late): a batch whose dS does not fit the cap is cut into chunks of whole batch entries, one fused launch per chunk on the SAME workspace (stream order).
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape,cap_mb", [((7, 640, 6, 6, True), 6), ((5, 512, 4, 2, False), 5), ((9, 1024, 2, 2, True), 5), ((4, 300, 3, 3, True), 1)],
                         ids=lambda x: "B%d_S%d_H%d_%d_c%d" % x if isinstance(x, tuple) else "cap%d" % x)
def test_fused_backward_in_chunks_of_batch_entries(be, knobs, dtype, shape, cap_mb):
    B, S, H, Hk, causal = shape
    torch.manual_seed(8)
    q = torch.randn(B, S, H, 128, device="cuda", dtype=dtype)
    k = torch.randn(B, S, Hk, 128, device="cuda", dtype=dtype)
    v, do = torch.randn_like(k), torch.randn_like(q)
    knobs.set("FA_BWD_FUSE_DELTA", 0); knobs.set("FA_BWD_DKDV", 8)   # (the pair on the fused launch's arithmetic: pre-pass delta, eight-wave dK/dV body)
    a = run_bwd(be, q, k, v, do, causal)
    knobs.set("FA_BWD_MODE", 3)
    whole = run_bwd(be, q, k, v, do, causal)
    knobs.set("FA_BWD_DS_CAP_MB", cap_mb); knobs.set("FA_DEBUG_POISON_WS", 1); knobs.set("FA_BWD_FUSED_CHECK", 1)
    plan = _plan_of(B, S, H, Hk, 128, causal)
    assert plan[0] == 3 and plan[1] >= 2 and plan[1] * plan[2] >= B > (plan[1] - 1) * plan[2], plan   # really chunked, the last chunk not empty
    f = run_bwd(be, q, k, v, do, causal)
    f2 = run_bwd(be, q, k, v, do, causal)
    assert a[3]["bwd_spill"] == 0 and whole[3]["bwd_spill"] == 3 and f[3]["bwd_spill"] == 3
    assert all(torch.equal(x, y) for x, y in zip(f[:3], whole[:3]))   # chunked == one launch, all three gradients, bit for bit
    assert all(torch.equal(x, y) for x, y in zip(f[:3], f2[:3]))
    assert torch.equal(a[1], f[1]) and torch.equal(a[2], f[2])       # dk, dv == the pair's


def test_default_table_chunks_a_large_batch(be, knobs):
    """64 x 1024 x 32 heads under a causal mask: 2.2 GiB of packed dS -> by default two launches of 32 batch entries on a 1.1 GiB workspace (the launch is +25 % on the
    pair at this size); gradients under the reference's rule and within 1.5x of the pair's error."""
    B, S, H = 64, 1024, 32
    plan = _plan_of(B, S, H, H, 128, True)
    assert plan[:3] == [3, 2, 32] and plan[7] <= 1280, plan
    torch.manual_seed(9)
    q = torch.randn(B, S, H, 128, device="cuda", dtype=torch.bfloat16)
    k, v, do = torch.randn_like(q), torch.randn_like(q), torch.randn_like(q)
    knobs.set("FA_DEBUG_POISON_WS", 1)
    f = run_bwd(be, q, k, v, do, True)
    assert f[3]["bwd_spill"] == 3, f[3]
    knobs.set("FA_BWD_MODE", -1)
    a = run_bwd(be, q, k, v, do, True)
    for sl in (slice(0, 2), slice(31, 33), slice(62, 64)):   # (an fp32 reference of the whole batch is 8 GB per matrix: the first, the seam's and the last entries)
        r = ref_grads(q[sl], k[sl], v[sl], do[sl], True, -1, -1); rl = ref_grads(q[sl], k[sl], v[sl], do[sl], True, -1, -1, upcast=False)
        for name, x, y, base, lo in zip(("dq", "dk", "dv"), f[:3], a[:3], r, rl):
            e, e_pair, e_pt = float((x[sl].float() - base).abs().max()), float((y[sl].float() - base).abs().max()), float((lo - base).abs().max())
            assert e <= 3 * e_pt + 1e-4 and e <= 1.5 * e_pair + 1e-4, (name, sl, e, e_pair, e_pt)
