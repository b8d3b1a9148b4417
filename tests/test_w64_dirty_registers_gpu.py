"""The hand-placed 64-per-wave kernels issue their MFMAs from inline asm: hipcc pads no hazard around them, and a result read a few wait states early comes from
whatever the registers held before -- within every tolerance, visible only as a run-to-run difference (round 5's peeled iteration shipped such a read; the full
suite's bitwise assertions caught it, no tolerance test did).  This file makes that failure mode a test of its own: every 64-per-wave instantiation runs twice on
the same inputs, each time right after a DIFFERENT kernel has filled the whole register file -- the 64-rows-per-wave forward itself (all 512 registers per lane) on
NaN inputs, then on huge finite ones (+-1e30 / +-6e4) -- and the two results must be bitwise equal (reference for what is computed: csrc/flash_attn/src/flash_fwd_kernel.h:309-440,
flash_bwd_kernel.h:457-733; the values themselves are checked elsewhere)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from flash_attn_amd import backend
    return backend


def dirty(be, knobs, kind, dtype):
    """Fill every SIMD's register file: the 512-register forward over every CU, on NaN (kind 0) or +-1e30 (kind 1) data."""
    old = {k: __import__("os").environ.get(k) for k in ("FA_FWD_NW",)}
    knobs.set("FA_FWD_NW", 64)
    big = 1e30 if dtype == torch.bfloat16 else 6e4
    q = torch.full((2, 2048, 64, 128), float("nan") if kind == 0 else big, device="cuda", dtype=dtype)
    if kind == 1:
        q[:, ::2] = -big
    be.fwd(q, q, q, None, None, 0.0, 1.0, True, -1, -1, 0.0, False, None)
    torch.cuda.synchronize()
    if old["FA_FWD_NW"] is None:
        knobs.unset("FA_FWD_NW")
    else:
        knobs.set("FA_FWD_NW", old["FA_FWD_NW"])


FWD = [("plain", {}), ("softcap", {"softcap": 20.0}), ("dropout", {"p": 0.1}), ("alibi", {"alibi": True})]


@pytest.mark.parametrize("d", [128, 64])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("name,feat", FWD, ids=[n for n, _ in FWD])
def test_forward_instantiations(be, knobs, name, feat, causal, dtype, d):
    if feat.get("alibi") and not causal:
        pytest.skip("the ALiBi variant of this kernel serves a causal right bound")
    torch.manual_seed(0)
    B, S, H = 2, 1536, 8
    q = torch.randn(B, S, H, d, device="cuda", dtype=dtype)
    k, v = torch.randn_like(q), torch.randn_like(q)
    al = torch.tensor([2.0 ** (-8.0 * (i + 1) / H) for i in range(H)], device="cuda", dtype=torch.float32) if feat.get("alibi") else None
    res = []
    for kind in (0, 1):
        dirty(be, knobs, kind, dtype)
        knobs.set("FA_FWD_NW", 64)
        torch.cuda.manual_seed(7)
        o, lse, _, _ = be.fwd(q, k, v, None, al, feat.get("p", 0.0), d ** -0.5, causal, -1, -1, feat.get("softcap", 0.0), False, None)
        assert be.last_schedule()["fwd_kernel"] == 3, be.last_schedule()
        res.append((o.clone(), lse.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert torch.isfinite(res[0][0].float()).all()


BWD = [("plain", {}, {}), ("softcap", {"softcap": 20.0}, {}), ("dropout_dq", {"p": 0.1}, {}), ("alibi", {"alibi": True}, {}),
       ("chunked5", {}, {"FA_BWD_MODE": 5, "FA_BWD_FUSE_DELTA": 0})]


@pytest.mark.parametrize("d", [128, 64])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("name,feat,env", BWD, ids=[n for n, _, _ in BWD])
def test_backward_instantiations(be, knobs, name, feat, env, dtype, d):
    if d == 64 and (feat.get("softcap") or feat.get("p")):
        pytest.skip("softcap / dropout variants of the 64-per-wave backward kernels: head dim 128")
    torch.manual_seed(0)
    B, S, H = 2, 1536, 8
    q = torch.randn(B, S, H, d, device="cuda", dtype=dtype)
    k, v, do = torch.randn_like(q), torch.randn_like(q), torch.randn_like(q)
    al = torch.tensor([2.0 ** (-8.0 * (i + 1) / H) for i in range(H)], device="cuda", dtype=torch.float32) if feat.get("alibi") else None
    p, cap = feat.get("p", 0.0), feat.get("softcap", 0.0)
    torch.cuda.manual_seed(7)
    o, lse, _, rng = be.fwd(q, k, v, None, al, p, d ** -0.5, True, -1, -1, cap, False, None)
    res = []
    for kind in (0, 1):
        dirty(be, knobs, kind, dtype)
        knobs.set("FA_BWD_DQ_NW", 64); knobs.set("FA_BWD_DKDV", 64); knobs.set("FA_BWD_MODE", -1)
        for kk, vv in env.items():
            knobs.set(kk, vv)
        dq, dk, dv, _ = be.bwd(do, q, k, v, o, lse, None, None, None, al, p, d ** -0.5, True, -1, -1, cap, False, None, rng)
        sch = be.last_schedule()
        assert sch["bwd_dq_nw"] == 64, sch
        if name in ("plain", "alibi", "softcap", "chunked5"):
            assert sch["bwd_dkdv_nw"] == 64, sch
        if name == "chunked5":
            assert sch["bwd_spill"] == 5, sch
        res.append((dq.clone(), dk.clone(), dv.clone()))
        for kk in ("FA_BWD_DQ_NW", "FA_BWD_DKDV", "FA_BWD_MODE", *env):
            knobs.unset(kk)
    for a, b in zip(*res):
        assert torch.equal(a, b)
        assert torch.isfinite(a.float()).all()
