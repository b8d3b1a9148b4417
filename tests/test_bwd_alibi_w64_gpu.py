"""ALiBi on the 64-per-wave backward kernels (csrc/fa_bwd_w64.hip: fa_bwd_dq_w64_kernel<.., ALIBI>, csrc/fa_bwd_dkdv_w64.hip: fa_bwd_dkdv_w64_kernel<.., ALIBI>;
reference: the Has_alibi switch of the one backward kernel, csrc/flash_attn/src/flash_bwd_kernel.h:457-733 + src/alibi.h).  Both run where the bias is linear in
the key -- under a causal right bound -- and are pinned here with FA_BWD_DQ_NW=64 / FA_BWD_DKDV=64 on small shapes.  Compared with
  * an fp32 PyTorch reference carrying the bias -slope * |key - row - (Sk - Sq)|, under the reference suite's rule for gradients (tests/test_flash_attn.py:
    error <= 3x the error of the same attention evaluated by PyTorch in the input dtype), and against the established feature kernels of fa_bwd.hip on the
    same inputs (<= 4x their error: the dQ kernel here multiplies by a Q that was scaled and rounded once, they scale every score in fp32; ALiBi sharpens the
    rows, so that rounding shows -- measured 0.026 against 0.009 at 0.013 for PyTorch-bf16, tools/alibi_bwd_diag.py); floors 1e-2 bf16 / 2e-3 fp16,
  * themselves, run twice (bitwise),
  * the same batch packed (varlen) against its sequences one by one (bitwise)."""
import itertools

import pytest
import torch

from tests.test_bwd_schedules_gpu import run_bwd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from flash_attn_amd import backend
    return backend


def ref_grads_alibi(q, k, v, do, slopes, wl):
    """fp32 gradients of causal (bottom-right aligned) attention with an ALiBi bias; slopes (H) or (B, H)."""
    qf, kf, vf = [x.float().transpose(1, 2).detach().requires_grad_(True) for x in (q, k, v)]
    g = qf.shape[1] // kf.shape[1]
    s = qf @ kf.repeat_interleave(g, 1).transpose(-1, -2) * q.shape[-1] ** -0.5
    Sq, Sk = s.shape[-2:]
    i = torch.arange(Sq, device=q.device)[:, None] + (Sk - Sq)
    j = torch.arange(Sk, device=q.device)[None]
    sl = slopes.float().reshape(-1, q.shape[2])[:, :, None, None]
    s = s - sl * (i - j).abs().float()
    m = j > i
    if wl >= 0:
        m = m | (j < i - wl)
    p = torch.softmax(s.masked_fill(m, float("-inf")), -1).nan_to_num(0.0)
    (p @ vf.repeat_interleave(g, 1)).backward(do.float().transpose(1, 2))
    return [x.grad.transpose(1, 2) for x in (qf, kf, vf)]


def pt_grads_alibi(q, k, v, do, slopes, wl):
    """The same gradients with the matrix products and P in the input dtype (scale applied to K): the 'PyTorch' error the reference's tests calibrate against."""
    qf, kf, vf = [x.transpose(1, 2).detach().requires_grad_(True) for x in (q, k, v)]
    g = qf.shape[1] // kf.shape[1]
    s = (qf @ (kf * q.shape[-1] ** -0.5).repeat_interleave(g, 1).transpose(-1, -2)).float()
    Sq, Sk = s.shape[-2:]
    i = torch.arange(Sq, device=q.device)[:, None] + (Sk - Sq)
    j = torch.arange(Sk, device=q.device)[None]
    s = s - slopes.float().reshape(-1, q.shape[2])[:, :, None, None] * (i - j).abs().float()
    m = j > i
    if wl >= 0:
        m = m | (j < i - wl)
    p = torch.softmax(s.masked_fill(m, float("-inf")), -1).nan_to_num(0.0).to(q.dtype)
    (p @ vf.repeat_interleave(g, 1)).backward(do.transpose(1, 2))
    return [x.grad.transpose(1, 2) for x in (qf, kf, vf)]


def slopes_for(B, H, per_batch):
    base = torch.tensor([2.0 ** (-8.0 * (h + 1) / H) for h in range(H)], device="cuda", dtype=torch.float32)
    if not per_batch:
        return base
    return (base[None] * torch.linspace(0.5, 1.5, B, device="cuda")[:, None]).contiguous()


SHAPES = [  # B, Sq, Sk, H, Hk, wl, slopes per batch
    (2, 512, 512, 4, 4, -1, False), (1, 1024, 1024, 4, 2, -1, True), (2, 333, 777, 6, 2, -1, True), (1, 2048, 2048, 8, 2, 256, False),
    (1, 777, 333, 4, 4, -1, False), (1, 65, 513, 2, 1, -1, False), (1, 4096, 4096, 2, 2, -1, False), (1, 31, 31, 1, 1, -1, False),
    (1, 700, 1, 2, 1, -1, False), (1, 3000, 3000, 4, 1, 1024, True), (3, 200, 1000, 8, 8, 64, True), (1, 1, 700, 2, 2, -1, False),
]


@pytest.mark.parametrize("d", [128, 64])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "B%d_Sq%d_Sk%d_H%d_%d_w%d_pb%d" % s)
def test_alibi_backward_on_the_w64_kernels_against_fp32_and_the_feature_kernels(be, knobs, shape, dtype, d):
    B, Sq, Sk, H, Hk, wl, per_batch = shape
    torch.manual_seed(0)
    q = torch.randn(B, Sq, H, d, device="cuda", dtype=dtype)
    k = torch.randn(B, Sk, Hk, d, device="cuda", dtype=dtype)
    v, do = torch.randn_like(k), torch.randn_like(q)
    sl = slopes_for(B, H, per_batch)
    knobs.set("FA_BWD_DQ_NW", 4)
    knobs.set("FA_BWD_DKDV", 8)
    g_old = run_bwd(be, q, k, v, do, True, wl, 0 if wl >= 0 else -1, alibi=sl)
    assert g_old[3]["bwd_dq_nw"] == 4 and g_old[3]["bwd_dkdv_nw"] == 8, g_old[3]
    knobs.set("FA_BWD_DQ_NW", 64)
    knobs.set("FA_BWD_DKDV", 64)
    g_new = run_bwd(be, q, k, v, do, True, wl, 0 if wl >= 0 else -1, alibi=sl)
    again = run_bwd(be, q, k, v, do, True, wl, 0 if wl >= 0 else -1, alibi=sl)
    assert g_new[3]["bwd_dq_nw"] == 64 and g_new[3]["bwd_dkdv_nw"] == 64, g_new[3]
    assert all(torch.equal(a, b) for a, b in zip(g_new[:3], again[:3])), "run-to-run"
    r = ref_grads_alibi(q, k, v, do, sl, wl)
    pt = pt_grads_alibi(q, k, v, do, sl, wl)
    floor = 1e-2 if dtype == torch.bfloat16 else 2e-3
    for i, name in enumerate(("dq", "dk", "dv")):
        assert torch.isfinite(g_new[i].float()).all(), name
        e_old, e_new = float((g_old[i].float() - r[i]).abs().max()), float((g_new[i].float() - r[i]).abs().max())
        e_pt = float((pt[i].float() - r[i]).abs().max())
        assert e_new <= max(3 * e_pt, floor), (name, e_new, e_pt)
        assert e_new <= max(4 * e_old, floor), (name, e_new, e_old)


def test_schedule_choice_with_alibi(be):
    q = torch.randn(1, 4096, 2, 128, device="cuda", dtype=torch.bfloat16)
    k, v, do = torch.randn_like(q), torch.randn_like(q), torch.randn_like(q)
    sl = slopes_for(1, 2, False)
    s = run_bwd(be, q, k, v, do, True, alibi=sl)[3]
    assert s["bwd_dq_nw"] == 64 and s["bwd_dkdv_nw"] == 64, s          # causal: the 64-per-wave kernels
    s = run_bwd(be, q, k, v, do, False, alibi=sl)[3]
    assert s["bwd_dq_nw"] != 64 and s["bwd_dkdv_nw"] == 8, s           # |key - row| is not linear without the causal bound: the feature kernels
    s = run_bwd(be, q, k, v, do, True, alibi=sl, softcap=15.0)[3]
    assert s["bwd_dq_nw"] != 64 and s["bwd_dkdv_nw"] == 8, s


@pytest.mark.parametrize("d", [128, 64])
def test_packed_alibi_batch_equals_its_sequences_bit_for_bit(be, knobs, d):
    knobs.set("FA_BWD_DKDV", 64)
    knobs.set("FA_BWD_DQ_NW", 64)
    torch.manual_seed(3)
    lens_q = [700, 33, 1500, 256, 64, 1, 900, 257, 0, 300]
    lens_k = [700, 65, 1500, 300, 64, 77, 513, 257, 5, 1]
    H, Hk = 4, 2
    cu_q = torch.tensor([0] + list(itertools.accumulate(lens_q)), dtype=torch.int32, device="cuda")
    cu_k = torch.tensor([0] + list(itertools.accumulate(lens_k)), dtype=torch.int32, device="cuda")
    q = torch.randn(sum(lens_q), H, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(sum(lens_k), Hk, d, device="cuda", dtype=torch.bfloat16)
    v, do = torch.randn_like(k), torch.randn_like(q)
    sl = slopes_for(len(lens_q), H, True)
    sc = d ** -0.5
    out, lse = be.varlen_fwd(q, k, v, None, cu_q, cu_k, None, None, None, sl, max(lens_q), max(lens_k), 0.0, sc, False, True, -1, -1, 0.0, False, None)[:2]
    dq, dk, dv = be.varlen_bwd(do, q, k, v, out, lse, None, None, None, cu_q, cu_k, sl, max(lens_q), max(lens_k), 0.0, sc, False, True, -1, -1, 0.0, False,
                               None, None)[:3]
    s = be.last_schedule()
    assert s["bwd_dkdv_nw"] == 64 and s["bwd_dq_nw"] == 64, s
    for b in range(len(lens_q)):
        a0, a1, b0, b1 = int(cu_q[b]), int(cu_q[b + 1]), int(cu_k[b]), int(cu_k[b + 1])
        if a1 == a0:
            assert torch.all(dk[b0:b1] == 0) and torch.all(dv[b0:b1] == 0)
            continue
        o1, l1 = out[None, a0:a1].contiguous(), lse[None, :, a0:a1].contiguous()
        g = be.bwd(do[None, a0:a1], q[None, a0:a1], k[None, b0:b1], v[None, b0:b1], o1, l1, None, None, None, sl[b:b + 1].contiguous(), 0.0, sc, True, -1, -1, 0.0,
                   False, None, None)
        assert torch.equal(dq[a0:a1], g[0][0]) and torch.equal(dk[b0:b1], g[1][0]) and torch.equal(dv[b0:b1], g[2][0]), b
