"""EXPERIMENTS (not collected by `pytest tests/`): run with FA_GFX950_LIB=experiments/libfa_gfx950_experiments.so python -m pytest experiments -m gpu
Backward schedules added in round 2 (reference: compute_dq_dk_dv_1colblock, csrc/flash_attn/src/flash_bwd_kernel.h:80-795):
  * fa_bwd_dq_w64_kernel  (FA_BWD_DQ_NW=64, the default from 2k keys at head dim 128): 64 query rows per wave;
  * the dS-spill path     (FA_BWD_MODE=2): the dK/dV kernel writes dS, dQ = dS.K is one contraction (5 instead of 7).
Both must reproduce the 32-rows-per-wave recomputing kernels: same arithmetic per element, so dq agrees to rounding of the
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


def ref_grads(q, k, v, do, causal, wl, wr):
    qf, kf, vf = [x.float().transpose(1, 2).detach().requires_grad_(True) for x in (q, k, v)]
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
    (p @ vf.repeat_interleave(g, 1)).backward(do.float().transpose(1, 2))
    return [x.grad.transpose(1, 2) for x in (qf, kf, vf)]


def run_bwd(be, q, k, v, do, causal, wl=-1, wr=-1, **feat):
    D = q.shape[-1]
    torch.cuda.manual_seed(11)
    out, lse, _, rng = be.fwd(q, k, v, None, feat.get("alibi"), feat.get("p_drop", 0.0), D ** -0.5, causal, wl, wr, feat.get("softcap", 0.0),
                              False, None)
    dq, dk, dv, _ = be.bwd(do, q, k, v, out, lse, None, None, None, feat.get("alibi"), feat.get("p_drop", 0.0), D ** -0.5, causal, wl, wr,
                           feat.get("softcap", 0.0), False, None, rng)
    return dq, dk, dv, be.last_schedule()


SHAPES = [  # B, Sq, Sk, H, Hk, causal, wl, wr
    (1, 256, 256, 2, 2, False, -1, -1), (1, 512, 512, 2, 1, True, -1, -1), (2, 1024, 1024, 4, 4, True, -1, -1),
    (1, 300, 333, 2, 2, False, -1, -1), (1, 300, 333, 2, 2, True, -1, -1), (1, 777, 1000, 3, 1, False, 100, 50),
    (1, 64, 64, 1, 1, True, -1, -1), (1, 1, 500, 2, 2, False, -1, -1), (2, 2048, 2048, 4, 2, True, -1, -1),
    (1, 1000, 200, 2, 2, True, -1, -1), (1, 513, 1025, 2, 2, False, 64, 0), (1, 33, 97, 2, 2, False, -1, -1),
    (1, 1025, 1025, 1, 1, True, -1, -1), (1, 200, 1000, 4, 1, True, -1, -1), (1, 2000, 2000, 1, 1, False, 0, 0),
    (1, 640, 640, 1, 1, False, 300, -1), (1, 640, 640, 1, 1, False, -1, 300),
]


@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", SHAPES[::2] + [SHAPES[5], SHAPES[9]], ids=lambda s: "B%d_Sq%d_Sk%d_H%d_%d_c%d_w%d_%d" % s)
def test_ds_spill_backward_equals_recomputing_backward(be, knobs, d, dtype, shape):
    """FA_BWD_MODE=2 (5 contractions): every feature is folded into dS by the dK/dV kernel, so softcap / ALiBi / dropout ride
    along.  dk, dv bit-exact (same kernel, the spill only adds stores); dq within accumulation-order rounding."""
    B, Sq, Sk, H, Hk, causal, wl, wr = shape
    torch.manual_seed(2)
    q = torch.randn(B, Sq, H, d, device="cuda", dtype=dtype)
    k = torch.randn(B, Sk, Hk, d, device="cuda", dtype=dtype)
    v, do = torch.randn_like(k), torch.randn_like(q)
    knobs.set("FA_BWD_DQ_NW", 4)
    for ft in FEATS:
        ft = dict(ft)
        if ft.get("alibi"):
            ft["alibi"] = torch.rand(B, H, device="cuda") * 0.3
        knobs.set("FA_BWD_MODE", 1)
        a = run_bwd(be, q, k, v, do, causal, wl, wr, **ft)
        knobs.set("FA_BWD_MODE", 2)
        s = run_bwd(be, q, k, v, do, causal, wl, wr, **ft)
        if s[3]["bwd_spill"] == 0:
            pytest.skip("dS-spill backward not in this build (experiments/build_experiments.py, FA_GFX950_LIB)")
        assert a[3]["bwd_spill"] == 0 and s[3]["bwd_spill"] == 1
        assert torch.equal(a[1], s[1]) and torch.equal(a[2], s[2]), list(ft)
        assert torch.isfinite(s[0].float()).all()
        tol = (1e-2 if dtype == torch.bfloat16 else 2e-3) * max(1.0, float(a[0].float().abs().max()))
        assert float((a[0].float() - s[0].float()).abs().max()) <= tol, list(ft)


def test_ds_spill_needs_its_workspace_and_respects_the_cap(be, knobs):
    """fa_bwd_workspace_bytes reports the dS scratch only under FA_BWD_MODE=2 and only below FA_BWD_DS_CAP_MB; without it the
    backward recomputes."""
    import ctypes as C
    from flash_attn_amd import _cabi
    torch.manual_seed(3)
    q = torch.randn(1, 512, 2, 128, device="cuda", dtype=torch.bfloat16)
    k, v, do = torch.randn_like(q), torch.randn_like(q), torch.randn_like(q)
    knobs.set("FA_BWD_MODE", 2)
    knobs.set("FA_BWD_DS_CAP_MB", 1)            # 1*2*16*16*2 KB = 1 MB fits, twice the heads does not
    if run_bwd(be, q, k, v, do, False)[3]["bwd_spill"] == 0:
        pytest.skip("dS-spill backward not in this build (experiments/build_experiments.py, FA_GFX950_LIB)")
    assert run_bwd(be, q, k, v, do, False)[3]["bwd_spill"] == 1
    q4 = torch.randn(1, 512, 4, 128, device="cuda", dtype=torch.bfloat16)
    assert run_bwd(be, q4, torch.randn_like(q4), torch.randn_like(q4), torch.randn_like(q4), False)[3]["bwd_spill"] == 0
    knobs.unset("FA_BWD_MODE")
    assert run_bwd(be, q, k, v, do, False)[3]["bwd_spill"] == 0
    assert _cabi is not None and C is not None


# (the 64-keys-per-wave dK/dV kernel is a product kernel since round 5: tests/test_bwd_dkdv_w64_gpu.py)
