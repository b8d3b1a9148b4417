"""GPU parity at the REAL shapes of BASELINE.json configs 2-5, and of every forward kernel the default dispatch can pick.

Reference = plain PyTorch attention in fp32 computed ON THE GPU one (batch, head) at a time (torch matmul / softmax /
autograd only -- independent of the HIP kernels; the fp64 numpy oracle needs hours at these sizes, it checks a sampled
sub-problem instead).  Tolerance = the reference's own acceptance rule (tests/test_flash_attn.py:1121,1130-1132): max
error <= 2x (forward) / 3x (gradients) the error of the same computation done by PyTorch in the input dtype, LSE to 2e-3
absolute.  Every test asserts through the C ABI's fa_last_schedule() which kernel instantiation actually ran, so a
change of the dispatch heuristic cannot silently move a config onto an untested kernel.
"""
import numpy as np
import pytest
import torch

from tests._util import attention_torch, max_abs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from flash_attn_amd import backend
    return backend


def ref_fwd_bwd(q, k, v, do, causal, window, upcast):
    """out, lse, dq, dk, dv of plain PyTorch attention, one (batch, head) at a time (fp32 when upcast)."""
    B, Sq, H, D = q.shape
    Hk = k.shape[2]
    g = H // Hk
    acc = torch.float32
    out = torch.empty(B, Sq, H, D, device=q.device, dtype=q.dtype if not upcast else acc)
    lse = torch.empty(B, H, Sq, device=q.device, dtype=torch.float32)
    dq = torch.empty_like(out)
    dk = torch.zeros(B, k.shape[1], Hk, D, device=q.device, dtype=acc)
    dv = torch.zeros_like(dk)
    for b in range(B):
        for h in range(H):
            hk = h // g
            qs = q[b:b + 1, :, h:h + 1].detach().clone().requires_grad_(do is not None)
            ks = k[b:b + 1, :, hk:hk + 1].detach().clone().requires_grad_(do is not None)
            vs = v[b:b + 1, :, hk:hk + 1].detach().clone().requires_grad_(do is not None)
            if upcast:
                qs, ks, vs = (t.float().detach().requires_grad_(do is not None) for t in (qs, ks, vs))
            o, l = attention_torch(qs, ks, vs, causal, window, upcast=upcast, reorder=not upcast)
            out[b:b + 1, :, h:h + 1] = o.detach()
            lse[b, h] = l[0, 0].detach()
            if do is not None:
                gq, gk, gv = torch.autograd.grad(o, (qs, ks, vs), do[b:b + 1, :, h:h + 1].to(o.dtype))
                dq[b:b + 1, :, h:h + 1] = gq
                dk[b:b + 1, :, hk:hk + 1] += gk.float()
                dv[b:b + 1, :, hk:hk + 1] += gv.float()
    return out, lse, dq, dk, dv


def lse_tolerance(sched, dtype, lse_abs_max=0.0):
    """LSE is not pinned by the reference's tests; ours: 2e-3 absolute -- except bf16 through the 64-rows-per-wave kernel,
    which multiplies Q by softmax_scale*log2(e) ONCE and rounds it to bf16: 2^-9 RELATIVE on every element of q, the same size as
    the bf16 rounding the PyTorch baseline applies to q*scale.  Measured at the BASELINE shapes (profiles/r03_numerics_default_vs_strict.txt):
    max 3.9e-3 / 4.4e-3 at configs 3 / 4-i against 8.8e-3 / 6.5e-3 for PyTorch computing in bf16 and 1.9e-6 for FA_STRICT=1; rows that see few
    keys average the rounding less -- the short sequences of the long-tail batch (config 4-ii) reach 6.75e-3 -- so the bound is 8e-3 for
    N(0,1) data (was 1e-2; 2^-9 |LSE| when LSE itself is large: spiked keys)."""
    if sched["fwd_kernel"] == 3 and dtype == torch.bfloat16:
        return max(8e-3, 2.0 ** -9 * lse_abs_max)
    return 2e-3


def check_against_reference(got, q, k, v, do, causal, window, what, lse_tol=2e-3):
    out, lse, dq, dk, dv = got
    r_out, r_lse, r_dq, r_dk, r_dv = ref_fwd_bwd(q, k, v, do, causal, window, True)
    p_out, _, p_dq, p_dk, p_dv = ref_fwd_bwd(q, k, v, do, causal, window, False)
    err, err_pt = max_abs(out.float(), r_out), max_abs(p_out.float(), r_out)
    assert err <= 2 * err_pt + 1e-5, (what, "out", err, err_pt)
    fin = torch.isfinite(r_lse)
    assert torch.equal(torch.isposinf(lse), ~fin), what
    assert max_abs(lse[fin], r_lse[fin]) < lse_tol, (what, "lse", max_abs(lse[fin], r_lse[fin]))
    assert not torch.isnan(out).any()
    if do is not None:
        for nm, g_, r_, p_ in (("dq", dq, r_dq, p_dq), ("dk", dk, r_dk, p_dk), ("dv", dv, r_dv, p_dv)):
            e, ep = max_abs(g_.float(), r_.float()), max_abs(p_.float(), r_.float())
            assert e <= 3 * ep + 1e-4, (what, nm, e, ep)
            assert not torch.isnan(g_).any()


def fixed_case(be, B, S, H, Hk, D, causal, window, bwd, seed=0):
    torch.manual_seed(seed)
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, S, Hk, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, S, Hk, D, device="cuda", dtype=torch.bfloat16)
    sc = D ** -0.5
    out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, sc, causal, window[0], window[1], 0.0, False, None)
    sched = be.last_schedule()
    do = dq = dk = dv = None
    if bwd:
        do = torch.randn_like(out)
        dq, dk, dv, _ = be.bwd(do, q, k, v, out, lse, None, None, None, None, 0.0, sc, causal, window[0], window[1], 0.0,
                               False, None, None)
    return (q, k, v, do), (out, lse, dq, dk, dv), sched


def test_config2_forward(be):
    """B=8 H=16 S=2048 D=64 bf16 non-causal, forward only."""
    (q, k, v, do), got, sched = fixed_case(be, 8, 2048, 16, 16, 64, False, (-1, -1), False)
    assert sched["fwd_kernel"] in (2, 3) and sched["d"] == 64, sched
    check_against_reference(got, q, k, v, None, False, (-1, -1), "config2 " + sched["name"], lse_tolerance(sched, q.dtype))


def test_config3_forward_backward(be):
    """B=4 H=32 S=4096 D=128 bf16 causal, forward + backward (the headline shape)."""
    (q, k, v, do), got, sched = fixed_case(be, 4, 4096, 32, 32, 128, True, (-1, -1), True)
    assert sched["fwd_kernel"] in (2, 3) and sched["d"] == 128, sched
    check_against_reference(got, q, k, v, do, True, (-1, -1), "config3 " + sched["name"], lse_tolerance(sched, q.dtype))


def test_config5_gqa_window_forward_backward(be):
    """q (2,8192,32,128), k/v (2,8192,8,128), causal + sliding window 1024, forward + backward."""
    (q, k, v, do), got, sched = fixed_case(be, 2, 8192, 32, 8, 128, True, (1024, 0), True)
    assert sched["fwd_kernel"] in (2, 3), sched
    check_against_reference(got, q, k, v, do, True, (1024, 0), "config5 " + sched["name"], lse_tolerance(sched, q.dtype))


def long_tail_lengths(total=65536, seed=0):
    """Config 4-ii: long-tail sequence lengths (recipe of benchmarks/benchmark_varlen_sched.py:76-84: mostly short
    sequences plus a few of the maximum length), generator seed 0, trimmed to `total` tokens."""
    g = torch.Generator().manual_seed(seed)
    lens = []
    while sum(lens) < total:
        x = float(torch.rand(1, generator=g))
        s = int(64 * (1.0 / max(x, 1e-3)) ** 0.9)
        lens.append(max(16, min(s, 16384)))
    lens[-1] -= sum(lens) - total
    if lens[-1] <= 0:
        lens.pop()
        lens[-1] += total - sum(lens)
    return lens


def varlen_case(be, lens, H, D, causal=True, seed=0):
    torch.manual_seed(seed)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    tot = int(cu[-1])
    q = torch.randn(tot, H, D, device="cuda", dtype=torch.bfloat16)
    k, v = torch.randn_like(q), torch.randn_like(q)
    sc, mx = D ** -0.5, max(lens)
    out, lse, _, _ = be.varlen_fwd(q, k, v, None, cu, cu, None, None, None, None, mx, mx, 0.0, sc, False, causal, -1, -1,
                                   0.0, False, None)
    sched = be.last_schedule()
    do = torch.randn_like(out)
    dq, dk, dv, _ = be.varlen_bwd(do, q, k, v, out, lse, None, None, None, cu, cu, None, mx, mx, 0.0, sc, False, causal,
                                  -1, -1, 0.0, False, None, None)
    sched_b = be.last_schedule()
    # reference per sequence
    for i, s in enumerate(lens):
        a, b_ = int(cu[i]), int(cu[i + 1])
        if b_ == a:
            continue
        got = (out[None, a:b_], lse[None, :, a:b_], dq[None, a:b_], dk[None, a:b_], dv[None, a:b_])
        check_against_reference(got, q[None, a:b_], k[None, a:b_], v[None, a:b_], do[None, a:b_], causal, (-1, -1),
                                f"varlen seq {i} len {s} {sched['name']}", lse_tolerance(sched, q.dtype))
    return sched, sched_b


def test_config4i_varlen_constant(be):
    """16 x 4096 tokens packed, H=16 D=128 causal: dense varlen grid."""
    sched, _ = varlen_case(be, [4096] * 16, 16, 128)
    assert sched["fwd_list"] == 0, sched


def test_config4ii_varlen_long_tail_uses_work_list(be):
    """Long-tail packed batch (total 65536 tokens): the work-list scheduler must be active, forward and backward."""
    lens = long_tail_lengths()
    assert sum(lens) == 65536
    sched, sched_b = varlen_case(be, lens, 16, 128)
    assert sched["fwd_list"] == 1, sched
    assert sched_b["bwd_list"] != 0, sched_b


# ---- every forward instantiation the dispatch can select, against the fp32 reference at long key loops -----------
@pytest.mark.parametrize("mode", ["full", "causal", "window"])
@pytest.mark.parametrize("S", [4096, 8192])
@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("nw", ["34", "38", "64"])
def test_pipelined_and_w64_kernels_long_sequences(be, knobs, nw, d, S, mode):
    """FA_FWD_NW pins the 4- / 8-wave pipelined kernels and the 64-rows-per-wave kernel; H = 16 KV heads exercises the
    hpx > 0 XCD mapping of the dense grid, Hk = 8 its GQA unit (both at tiles >= 48, where the heuristic picks the
    8-wave pipelined kernel on its own)."""
    knobs.set("FA_FWD_NW", nw)
    causal = mode != "full"
    window = (S // 4 + 3, 0) if mode == "window" else (-1, -1)
    H, Hk = (16, 16) if d == 64 else (16, 8)
    (q, k, v, _), got, sched = fixed_case(be, 1, S, H, Hk, d, causal, window, False, seed=S + d)
    want = {"34": (2, 4), "38": (2, 8), "64": (3, 4)}[nw]
    assert (sched["fwd_kernel"], sched["fwd_nw"]) == want, sched
    check_against_reference(got, q, k, v, None, causal, window, f"nw={nw} " + sched["name"], lse_tolerance(sched, q.dtype))


@pytest.mark.parametrize("nw", ["34", "38", "64"])
@pytest.mark.parametrize("thr", ["0", "8"])
def test_rescale_branch_forced_pipelined_kernels(be, knobs, nw, thr):
    """Spiked keys force the deferred-rescale branch of the pipelined / 64-row kernels (the lock-step kernels have this
    test in test_fwd_gpu.py); checked on the full tensor against the fp64 oracle."""
    from oracle import attention_oracle as orc
    knobs.set("FA_FWD_NW", nw)
    knobs.set("FA_RESCALE_THR", thr)
    torch.manual_seed(11)
    B, S, H, D = 1, 1024, 2, 128
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16)
    for i, row in enumerate(range(5, S, 97)):
        tile = (3 * i + 2) % (S // 64)
        key = tile * 64 + (7 * i) % 64
        k[0, key, :, :] = (q[0, row, :, :].float() * (1.0 + 0.25 * i)).to(torch.bfloat16)
    out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, D ** -0.5, False, -1, -1, 0.0, False, None)
    sched = be.last_schedule()
    assert (sched["fwd_kernel"], sched["fwd_nw"]) == {"34": (2, 4), "38": (2, 8), "64": (3, 4)}[nw], sched
    ref, lse_ref = orc.attention_fwd(q, k, v)
    assert max_abs(out.float(), torch.from_numpy(ref).cuda()) < 2e-2
    assert max_abs(lse, torch.from_numpy(lse_ref).cuda().float()) < lse_tolerance(sched, q.dtype, float(np.abs(lse_ref).max()))


def test_default_dispatch_covers_only_tested_kernels(be):
    """Sweep the shapes of the headline benchmark through the DEFAULT heuristic and record which kernels it picks: each
    (kernel, waves) pair must be one of those the tests above compare with a reference."""
    tested = {(2, 4), (2, 8), (3, 4), (1, 4), (1, 8)}
    seen = set()
    for D, H in ((128, 16), (64, 32)):
        for causal in (False, True):
            for S in (512, 1024, 2048, 4096, 8192, 16384):
                B = max(1, 16384 // S) if S <= 4096 else 1
                q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16)
                be.fwd(q, q, q, None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None)
                s = be.last_schedule()
                seen.add((s["fwd_kernel"], s["fwd_nw"]))
    assert seen <= tested, seen


def test_sampled_block_against_fp64_oracle_config3(be):
    """The numpy fp64 oracle on a sampled (batch, head) of config 3 -- ties the big-shape GPU reference back to the
    pinned oracle (oracle == attention_ref on the golden vectors, tests/test_oracle_cpu.py)."""
    from oracle import attention_oracle as orc
    torch.manual_seed(0)
    B, S, H, D = 4, 4096, 32, 128
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16)
    k, v = torch.randn_like(q), torch.randn_like(q)
    out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, D ** -0.5, True, -1, -1, 0.0, False, None)
    sched = be.last_schedule()
    for (b, h) in ((0, 0), (3, 31), (1, 17)):
        ref, lse_ref = orc.attention_fwd(q[b:b + 1, :, h:h + 1], k[b:b + 1, :, h:h + 1], v[b:b + 1, :, h:h + 1], None, True)
        assert max_abs(out[b:b + 1, :, h:h + 1].float(), torch.from_numpy(ref).cuda()) < 2e-2
        assert max_abs(lse[b, h], torch.from_numpy(lse_ref[0, 0]).cuda().float()) < lse_tolerance(sched, q.dtype)


def test_sampled_block_gradients_against_fp64_oracle_config3(be):
    """The same tie-back for the BACKWARD: dQ, dK, dV of one sampled (batch, head) of config 3 against the numpy fp64 oracle (pinned to the
    reference's attention_ref + autograd on the golden vectors).  The units of a batch are independent, so the (b, h) slice of the full-shape
    gradients IS the gradient of the slice; the bound is the reference's 3x rule against PyTorch computing the same slice in bf16, and the fp32 GPU
    reference the full-shape test uses (ref_fwd_bwd) is checked against the oracle on the same slice: the chain oracle -> fp32 reference -> kernel
    is pinned end to end."""
    from oracle import attention_oracle as orc
    torch.manual_seed(0)
    B, S, H, D = 4, 4096, 32, 128
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16)
    k, v = torch.randn_like(q), torch.randn_like(q)
    do = torch.randn_like(q)
    out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, D ** -0.5, True, -1, -1, 0.0, False, None)
    dq, dk, dv, _ = be.bwd(do, q, k, v, out, lse, None, None, None, None, 0.0, D ** -0.5, True, -1, -1, 0.0, False, None, None)
    b, h = 2, 19
    sl = lambda t: t[b:b + 1, :, h:h + 1]
    g64 = orc.attention_bwd(sl(do), sl(q), sl(k), sl(v), None, None, None, True)[:3]
    r32 = ref_fwd_bwd(sl(q), sl(k), sl(v), sl(do), True, (-1, -1), True)[2:]
    p16 = ref_fwd_bwd(sl(q), sl(k), sl(v), sl(do), True, (-1, -1), False)[2:]
    for nm, got, r, p_, g in zip(("dq", "dk", "dv"), (dq, dk, dv), r32, p16, g64):
        g = torch.from_numpy(np.asarray(g)).cuda()
        e, e32, e16 = max_abs(sl(got).float(), g), max_abs(r.float(), g), max_abs(p_.float(), g)
        assert e32 < 1e-4, (nm, "fp32 reference vs fp64 oracle", e32)
        assert e <= 3 * e16 + 1e-4, (nm, e, e16)
