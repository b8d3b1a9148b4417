"""The chunked 5-contraction backward (FA_BWD_MODE=5, csrc/fa_bwd_dkdv_w64.hip: fa_bwd_c5_kernel; reference: compute_dq_dk_dv_1colblock forms S, dP and dS once and
takes all three gradients from them, csrc/flash_attn/src/flash_bwd_kernel.h:457-733).  Opt-in -- it was measured behind the recomputing pair, profiles/r06_bwd_c5.txt --
and therefore pinned here so that it stays correct:
  * dK / dV bitwise equal to the recomputing path's (the same 64-keys-per-wave kernel text, softmax_d from the pre-pass in both),
  * dQ within the reference's rule against fp32 PyTorch (<= 3x the error of PyTorch attention in the input dtype) and within rounding of the recomputing kernel's,
  * bitwise run to run, with every 16-bit word of the workspace a NaN beforehand (a read of a dS sub-tile nobody wrote shows), several chunks alternating the slots."""
import pytest
import torch

from tests.test_bwd_schedules_gpu import ref_grads, run_bwd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from flash_attn_amd import backend
    return backend


CASES = [  # B, Sq, Sk, H, Hk, causal, wr, cap MB (0 = default)
    (1, 256, 256, 2, 2, False, -1, 0), (1, 512, 512, 2, 1, True, -1, 0), (2, 1024, 1024, 4, 4, True, -1, 0), (1, 300, 333, 2, 2, False, -1, 0), (1, 300, 333, 2, 2, True, -1, 0),
    (1, 64, 64, 1, 1, True, -1, 0), (1, 1, 500, 2, 2, False, -1, 0), (1, 1025, 1025, 1, 1, True, -1, 0), (1, 200, 1000, 4, 1, True, -1, 0), (1, 777, 1000, 3, 1, False, -1, 0),
    (1, 640, 900, 2, 2, False, 100, 0), (1, 640, 640, 2, 2, False, 37, 0),
    # several chunks: many units per XCD, GQA groups as units, a ragged last round of units
    (3, 512, 512, 32, 32, True, -1, 16), (5, 300, 333, 8, 8, False, -1, 16), (3, 768, 1024, 32, 16, True, -1, 32), (7, 640, 640, 6, 6, True, -1, 16), (4, 2048, 2048, 16, 4, True, -1, 512),
]


@pytest.mark.parametrize("d", [128, 64])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("case", CASES, ids=lambda s: "B%d_Sq%d_Sk%d_H%d_%d_c%d_wr%d_cap%d" % s)
def test_chunked_five_contraction_backward(be, knobs, monkeypatch, case, dtype, d):
    B, Sq, Sk, H, Hk, causal, wr, cap = case
    torch.manual_seed(0)
    q = torch.randn(B, Sq, H, d, device="cuda", dtype=dtype)
    k = torch.randn(B, Sk, Hk, d, device="cuda", dtype=dtype)
    v, do = torch.randn_like(k), torch.randn_like(q)
    knobs.set("FA_BWD_FUSE_DELTA", 0)    # softmax_d from the pre-pass in both paths: bitwise dK / dV
    knobs.set("FA_BWD_DKDV", 64)
    knobs.set("FA_BWD_MODE", -1)
    pair = run_bwd(be, q, k, v, do, causal, -1, wr)
    assert pair[3]["bwd_spill"] == 0 and pair[3]["bwd_dkdv_nw"] == 64
    monkeypatch.setenv("FA_DEBUG_POISON_WS", "1")
    if cap:
        knobs.set("FA_BWD_C5_CAP_MB", cap)
    knobs.set("FA_BWD_MODE", 5)
    c5 = run_bwd(be, q, k, v, do, causal, -1, wr)
    again = run_bwd(be, q, k, v, do, causal, -1, wr)
    assert c5[3]["bwd_spill"] == 5, c5[3]
    assert all(torch.isfinite(x.float()).all() for x in c5[:3])
    assert all(torch.equal(a, b) for a, b in zip(c5[:3], again[:3])), "run-to-run"
    assert torch.equal(pair[1], c5[1]) and torch.equal(pair[2], c5[2]), "dK / dV come from the same kernel text"
    r = ref_grads(q, k, v, do, causal, -1, wr)
    pt = ref_grads(q, k, v, do, causal, -1, wr, upcast=False)
    e5, e7, ept = (float((x[0].float() - r[0]).abs().max()) for x in (c5, pair, pt))
    assert e5 <= 3 * ept + 1e-5, (e5, ept)
    assert e5 <= 2 * e7 + 1e-5, (e5, e7)


def test_what_it_does_not_cover_runs_the_pair(be, knobs):
    knobs.set("FA_BWD_MODE", 5)
    for (Sq, Sk, causal, wl, feat) in ((640, 640, False, 300, {}), (1000, 200, True, -1, {}), (512, 512, True, -1, {"softcap": 20.0})):
        q = torch.randn(1, Sq, 2, 128, device="cuda", dtype=torch.bfloat16)
        k = torch.randn(1, Sk, 2, 128, device="cuda", dtype=torch.bfloat16)
        g = run_bwd(be, q, k, torch.randn_like(k), torch.randn_like(q), causal, wl, -1, **feat)
        assert g[3]["bwd_spill"] == 0, (Sq, Sk, g[3])
