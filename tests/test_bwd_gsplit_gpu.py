"""GQA group split of the dK/dV kernels (fa_api.cpp bwd_gsplit_plan, late round 6; reference: per-query-head dK / dV in the input dtype summed by at::sum_out,
csrc/flash_attn/flash_api.cpp:1000-1004).  With few kv heads and a small batch the (batch, kv head, key block) grid does not fill the chip: the group is split into
virtual kv heads, their partial dK / dV go to a workspace and one small kernel sums them.  Checked: dK / dV under the reference's rule against fp32 (<= 3x the error of
PyTorch in the input dtype), dQ bit for bit the unsplit path's, run-to-run bitwise, workspace poisoned -- forced (FA_BWD_GSPLIT=16), automatic and off."""
import pytest
import torch

from tests.test_bwd_schedules_gpu import _plan_of, ref_grads, run_bwd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from flash_attn_amd import backend
    return backend


SHAPES = [  # B, Sq, Sk, H, Hk, D, causal, dtype, features
    (2, 1024, 1024, 32, 2, 128, True, torch.bfloat16, {}), (1, 777, 1000, 16, 1, 128, True, torch.float16, {}), (2, 512, 512, 8, 2, 64, False, torch.bfloat16, {}),
    (1, 300, 300, 12, 3, 96, True, torch.bfloat16, {}), (1, 2500, 2500, 16, 2, 128, True, torch.bfloat16, {}), (1, 1024, 1024, 16, 4, 256, True, torch.bfloat16, {}),
    (3, 200, 200, 6, 3, 32, False, torch.bfloat16, {}), (1, 1, 333, 8, 2, 128, False, torch.float16, {}), (2, 640, 640, 8, 1, 128, False, torch.float16, {"softcap": 20.0}),
    (2, 1024, 1024, 8, 2, 128, True, torch.bfloat16, {"p_drop": 0.1}), (1, 1500, 1500, 8, 2, 128, True, torch.bfloat16, {"alibi": True}),
    (1, 4096, 4096, 8, 2, 128, False, torch.bfloat16, {"wl": 500, "wr": 0}),
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "B%d_Sq%d_Sk%d_H%d_%d_D%d_c%d_%s_%s" % (s[:7] + (str(s[7])[6:], "-".join(s[8]) or "plain")))
def test_split_group_matches_unsplit_and_fp32(be, knobs, shape):
    B, Sq, Sk, H, Hk, D, causal, dtype, feat = shape
    feat = dict(feat)
    wl, wr = feat.pop("wl", -1), feat.pop("wr", -1)
    if feat.pop("alibi", False): feat["alibi"] = torch.tensor([2.0 ** (-8.0 * (i + 1) / H) for i in range(H)], device="cuda", dtype=torch.float32)
    torch.manual_seed(Sq + H)
    q = torch.randn(B, Sq, H, D, device="cuda", dtype=dtype)
    k = torch.randn(B, Sk, Hk, D, device="cuda", dtype=dtype)
    v, do = torch.randn_like(k), torch.randn_like(q)
    knobs.set("FA_DEBUG_POISON_WS", 1)
    res = {}
    for gs in (0, 1, 16):
        knobs.set("FA_BWD_GSPLIT", gs)
        res[gs] = run_bwd(be, q, k, v, do, causal, wl, wr, **feat)
        again = run_bwd(be, q, k, v, do, causal, wl, wr, **feat)
        assert all(torch.equal(x, y) for x, y in zip(res[gs][:3], again[:3])), ("run-to-run", gs)
        assert all(torch.isfinite(x.float()).all() for x in res[gs][:3])
    assert torch.equal(res[0][0], res[16][0]) and torch.equal(res[0][0], res[1][0]), "dq is not the split's"
    assert not torch.equal(res[0][1], res[16][1]) or H // Hk < 2 or Sq == 1, "the forced split ran"   # (partials rounded to the input dtype: equal only by accident)
    if "p_drop" in feat or "softcap" in feat or "alibi" in feat:   # (no plain-PyTorch yardstick for these here: against the unsplit kernels, within the partials' rounding)
        for i in (1, 2):
            scale = float(res[0][i].float().abs().max())
            assert float((res[0][i].float() - res[16][i].float()).abs().max()) <= 2.0 ** -6 * max(scale, 1.0), i
        return
    r, pt = ref_grads(q, k, v, do, causal, wl, wr), ref_grads(q, k, v, do, causal, wl, wr, upcast=False)
    for i in (1, 2):
        for gs in (1, 16):
            e, ept = float((res[gs][i].float() - r[i]).abs().max()), float((pt[i] - r[i]).abs().max())
            assert e <= 3 * ept + 1e-5, (i, gs, e, ept)


def test_automatic_split_fills_the_chip(be, knobs):
    """The plan (fa_bwd_plan_query out[3]): split until ~1024 workgroups, at most 8 virtual heads, never past the group; nothing where the grid is large or the batch is packed."""
    assert _plan_of(2, 1024, 32, 2, 128, True)[3] == 8        # 16 items -> 8 virtual heads per group
    assert _plan_of(1, 4096, 32, 4, 128, True)[3] == 8
    assert _plan_of(4, 4096, 32, 8, 128, True)[3] == 2        # 512 uneven items: two (+10 %)
    assert _plan_of(8, 4096, 32, 8, 128, True)[3] == 0        # 1024 items: unsplit
    assert _plan_of(2, 1024, 4, 2, 128, True)[3] == 2         # the group has two heads
    assert _plan_of(2, 1024, 8, 8, 128, True)[3] == 0         # no group
    plan = _plan_of(8, 2048, 32, 8, 128, True)
    assert plan[0] == 3                                         # (64 units: the fused launch's, which does not split)
