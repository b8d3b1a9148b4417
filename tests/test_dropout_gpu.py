"""GPU parity tests of dropout (forward + backward + return_softmax payload) against the fp64 oracle.

The kernels regenerate the keep-mask from (rng_state, batch, head, query, key); ``return_softmax`` hands the
random byte of every pair back (the ROCm backend's payload, reference csrc/flash_attn_ck/mha_fwd.cpp:275-279,
tests/test_flash_attn_ck.py:49-53: kept iff byte <= floor(255 * (1 - p))).  The tests take the mask from that
payload, feed it to the oracle (pinned to the reference's attention_ref on CPU, tests/test_oracle_cpu.py) and
compare outputs and all three gradients -- so a backward kernel that regenerated a different mask would fail.
"""
import math

import numpy as np
import pytest
import torch

from tests._util import max_abs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["ext", "ctypes"])
def be(request):
    if request.param == "ext":
        import flash_attn_2_cuda as m
    else:
        from flash_attn_amd import backend as m
    return m


def _keep(randval, p):
    return randval.to(torch.int32) <= math.floor(255.0 * (1.0 - p))


def _visible(sq, sk, causal, window):
    from oracle import attention_oracle as orc
    _, wl, wr = orc.normalize_window(sq, sk, causal, window[0], window[1])
    return orc.visible_mask(sq, sk, wl, wr)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("p", [0.17, 0.5])
@pytest.mark.parametrize("sq,sk,h,hk,causal,window", [
    (128, 128, 4, 4, False, (-1, -1)), (113, 203, 4, 2, True, (-1, -1)), (256, 130, 2, 1, True, (-1, -1)),
    (200, 333, 4, 4, False, (50, 20)), (97, 400, 6, 2, False, (-1, -1)), (300, 300, 2, 2, True, (64, 0))])
def test_dropout_fwd_bwd_vs_oracle(be, sq, sk, h, hk, causal, window, p, d, dtype):
    from oracle import attention_oracle as orc
    torch.manual_seed(sq * 3 + sk)
    B = 2
    q = torch.randn(B, sq, h, d, device="cuda", dtype=dtype)
    k = torch.randn(B, sk, hk, d, device="cuda", dtype=dtype)
    v = torch.randn(B, sk, hk, d, device="cuda", dtype=dtype)
    do = torch.randn(B, sq, h, d, device="cuda", dtype=dtype)
    scale = d ** -0.5
    out, lse, rv, rng = be.fwd(q, k, v, None, None, p, scale, causal, window[0], window[1], 0.0, True, None)
    assert rv.dtype == torch.uint8 and tuple(rv.shape) == (B, h, sq, sk) and tuple(rng.shape) == (2,)
    keep = _keep(rv, p).cpu().numpy()
    vis = _visible(sq, sk, causal, window)
    if vis.sum() * B * h > 20000:  # kept fraction of the visible pairs (reference get_dropout_fraction check)
        frac = keep[:, :, vis].mean()
        assert abs(frac - (math.floor(255 * (1 - p)) + 1) / 256) < 0.01
    o_ref, l_ref = orc.attention_fwd(q, k, v, scale, causal, window, 0.0, None, p, keep)
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert max_abs(out.float().cpu(), torch.from_numpy(o_ref)) < tol / (1 - p)
    live = np.isfinite(l_ref)
    assert max_abs(lse.cpu()[torch.from_numpy(live)], torch.from_numpy(l_ref[live]).float()) < 2e-3
    dq, dk, dv, _ = be.bwd(do, q, k, v, out, lse, None, None, None, None, p, scale, causal, window[0], window[1], 0.0, False, None, rng)
    rq, rk, rvv, _ = orc.attention_bwd(do, q, k, v, out, lse, scale, causal, window, 0.0, None, p, keep)
    for nm, got, ref in (("dq", dq, rq), ("dk", dk, rk), ("dv", dv, rvv)):
        gtol = (4e-2 if dtype == torch.bfloat16 else 8e-3) * max(1.0, float(np.abs(ref).max())) / (1 - p)
        assert max_abs(got.float().cpu(), torch.from_numpy(ref)) < gtol, (nm, max_abs(got.float().cpu(), torch.from_numpy(ref)), gtol)
    # same rng_state => bitwise the same backward; the forward replays from a seeded generator
    dq2, dk2, dv2, _ = be.bwd(do, q, k, v, out, lse, None, None, None, None, p, scale, causal, window[0], window[1], 0.0, False, None, rng)
    assert torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)


def test_dropout_varlen_and_generator_semantics(be):
    from oracle import attention_oracle as orc
    torch.manual_seed(5)
    lens_q, lens_k = [70, 1, 200, 129], [70, 33, 260, 129]
    H, Hk, d, p = 4, 2, 128, 0.25
    cu_q = torch.tensor([0] + list(np.cumsum(lens_q)), dtype=torch.int32, device="cuda")
    cu_k = torch.tensor([0] + list(np.cumsum(lens_k)), dtype=torch.int32, device="cuda")
    tq, tk = sum(lens_q), sum(lens_k)
    q = torch.randn(tq, H, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(tk, Hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(tk, Hk, d, device="cuda", dtype=torch.bfloat16)
    do = torch.randn(tq, H, d, device="cuda", dtype=torch.bfloat16)
    scale = d ** -0.5
    args = (q, k, v, None, cu_q, cu_k, None, None, None, None, max(lens_q), max(lens_k), p, scale, False, True, -1, -1, 0.0, True, None)
    torch.manual_seed(123)
    out, lse, rv, rng = be.varlen_fwd(*args)
    assert tuple(rv.shape) == (H, tq, max(lens_k)) and rv.dtype == torch.uint8
    torch.manual_seed(123)
    out_b, _, rv_b, rng_b = be.varlen_fwd(*args)       # re-seeded generator => same mask, same output
    assert torch.equal(rng, rng_b) and torch.equal(rv, rv_b) and torch.equal(out, out_b)
    _, _, rv_c, rng_c = be.varlen_fwd(*args)           # the generator's offset advanced => another mask
    assert not torch.equal(rng, rng_c) and not torch.equal(rv, rv_c)
    dq, dk, dv, _ = be.varlen_bwd(do, q, k, v, out, lse, None, None, None, cu_q, cu_k, None, max(lens_q), max(lens_k), p, scale,
                                  False, True, -1, -1, 0.0, False, None, rng)
    for b in range(len(lens_q)):
        qs, ks = slice(int(cu_q[b]), int(cu_q[b + 1])), slice(int(cu_k[b]), int(cu_k[b + 1]))
        keep = _keep(rv[:, qs, : lens_k[b]], p).cpu().numpy()[None]
        o_ref, _ = orc.attention_fwd(q[qs][None], k[ks][None], v[ks][None], scale, True, (-1, -1), 0.0, None, p, keep)
        assert max_abs(out[qs].float().cpu(), torch.from_numpy(o_ref[0])) < 3e-2
        rq, rk, rvv, _ = orc.attention_bwd(do[qs][None], q[qs][None], k[ks][None], v[ks][None], None, None, scale, True, (-1, -1),
                                           0.0, None, p, keep)
        for got, ref in ((dq[qs], rq[0]), (dk[ks], rk[0]), (dv[ks], rvv[0])):
            assert max_abs(got.float().cpu(), torch.from_numpy(ref)) < 6e-2 * max(1.0, float(np.abs(ref).max()))


def test_dropout_random_bytes_are_uniform_and_uncorrelated(be):
    torch.manual_seed(9)
    B, S, H, d = 2, 512, 4, 64
    q = torch.randn(B, S, H, d, device="cuda", dtype=torch.bfloat16)
    _, _, rv, _ = be.fwd(q, q, q, None, None, 0.1, d ** -0.5, False, -1, -1, 0.0, True, None)
    x = rv.cpu().numpy().astype(np.int64)
    n = x.size
    hist = np.bincount(x.ravel(), minlength=256)
    assert np.abs(hist / n - 1 / 256).max() < 5 * math.sqrt((1 / 256) / n)       # every byte value equally likely (5 sigma)
    xf = (x - 127.5) / 73.9
    for a, b in ((xf[..., :, :-1], xf[..., :, 1:]), (xf[..., :-1, :], xf[..., 1:, :]), (xf[:, :-1], xf[:, 1:]), (xf[:-1], xf[1:]),
                 (xf[..., :, :-4], xf[..., :, 4:]), (xf[..., :-4, :], xf[..., 4:, :])):
        assert abs((a * b).mean()) < 5 / math.sqrt(a.size)                       # neighbours along keys / queries / heads / batch


def test_dropout_streams_of_different_heads_do_not_alias(be):
    """Every (batch, head) pair against every other one: the random-byte planes (Sq x Sk) must be uncorrelated at equal
    position AND at any small shift along queries / key groups -- the failure mode of a single additive counter, where one
    head's stream is a shifted copy of another's (reference: Philox keyed per (batch, head), dropout.h:31-90).  B*H = 48
    streams of 256 x 512 bytes: 1128 pairs x 9 shifts, each tested at 6 sigma; and no 4-byte word plane may be EQUAL."""
    torch.manual_seed(10)
    B, S, Sk, H, d = 3, 256, 512, 16, 64
    q = torch.randn(B, S, H, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, Sk, H, d, device="cuda", dtype=torch.bfloat16)
    _, _, rv, _ = be.fwd(q, k, k, None, None, 0.2, d ** -0.5, False, -1, -1, 0.0, True, None)
    x = rv.reshape(B * H, S, Sk).float().cuda()
    xf = (x - 127.5) / 73.9
    n = S * Sk
    worst = 0.0
    for dr, dg in ((0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (0, 2), (3, 1), (1, 3), (0, 8)):
        a = xf[:, : S - dr, : Sk - 4 * dg].reshape(B * H, -1)
        b_ = xf[:, dr:, 4 * dg:].reshape(B * H, -1)
        c = (a @ b_.T) / a.shape[1]                      # correlation of stream i with stream j shifted by (dr rows, dg groups)
        if dr == 0 and dg == 0:
            c = c - torch.diag(torch.diag(c))            # a stream is of course equal to itself at shift 0
        worst = max(worst, float(c.abs().max()) * math.sqrt(a.shape[1]))
    assert worst < 6.0, worst
    words = rv.reshape(B * H, S, Sk // 4, 4).to(torch.int32)
    w = (words[..., 0] | (words[..., 1] << 8) | (words[..., 2] << 16) | (words[..., 3] << 24)).reshape(B * H, -1)
    eq = (w[:, None, :] == w[None, :, :]).float().mean(-1)
    assert float((eq - torch.eye(B * H, device=eq.device)).max()) < 1e-4   # chance level 2^-32


def test_dropout_through_the_public_interface():
    from flash_attn_amd import flash_attn_interface as fi
    from oracle import attention_oracle as orc
    torch.manual_seed(11)
    B, S, H, d, p = 2, 160, 4, 64, 0.17
    q = torch.randn(B, S, H, d, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    k = torch.randn(B, S, 2, d, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    v = torch.randn(B, S, 2, d, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    out, lse, rv = fi.flash_attn_func(q, k, v, dropout_p=p, causal=True, return_attn_probs=True)
    g = torch.randn_like(out)
    dq, dk, dv = torch.autograd.grad(out, (q, k, v), g)
    keep = _keep(rv, p).cpu().numpy()
    o_ref, _ = orc.attention_fwd(q, k, v, None, True, (-1, -1), 0.0, None, p, keep)
    rq, rk, rvv, _ = orc.attention_bwd(g, q, k, v, None, None, None, True, (-1, -1), 0.0, None, p, keep)
    assert max_abs(out.float().cpu(), torch.from_numpy(o_ref)) < 3e-2
    for got, ref in ((dq, rq), (dk, rk), (dv, rvv)):
        assert max_abs(got.float().cpu(), torch.from_numpy(ref)) < 6e-2 * max(1.0, float(np.abs(ref).max()))
    with pytest.raises(RuntimeError, match="return_softmax"):
        fi._flash_attn_forward(q.detach(), k.detach(), v.detach(), 0.0, d ** -0.5, True, -1, -1, 0.0, None, True)


@pytest.mark.parametrize("feature", ["alibi", "softcap"])
def test_dropout_combined_with_alibi_or_softcap(be, feature):
    """The two-feature kernel variants (ALiBi + dropout, softcap + dropout), forward and backward, against the oracle."""
    from oracle import attention_oracle as orc
    torch.manual_seed(21)
    B, sq, sk, h, hk, d, p = 2, 150, 260, 4, 2, 64, 0.2
    q = torch.randn(B, sq, h, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, sk, hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, sk, hk, d, device="cuda", dtype=torch.bfloat16)
    do = torch.randn(B, sq, h, d, device="cuda", dtype=torch.bfloat16)
    alibi = torch.rand(B, h, device="cuda") * 0.3 if feature == "alibi" else None
    cap = 15.0 if feature == "softcap" else 0.0
    scale = d ** -0.5
    out, lse, rv, rng = be.fwd(q, k, v, None, alibi, p, scale, True, -1, -1, cap, True, None)
    keep = _keep(rv, p).cpu().numpy()
    al = None if alibi is None else alibi.cpu().numpy()
    o_ref, _ = orc.attention_fwd(q, k, v, scale, True, (-1, -1), cap, al, p, keep)
    assert max_abs(out.float().cpu(), torch.from_numpy(o_ref)) < 3e-2
    dq, dk, dv, _ = be.bwd(do, q, k, v, out, lse, None, None, None, alibi, p, scale, True, -1, -1, cap, False, None, rng)
    rq, rk, rvv, _ = orc.attention_bwd(do, q, k, v, None, None, scale, True, (-1, -1), cap, al, p, keep)
    for got, ref in ((dq, rq), (dk, rk), (dv, rvv)):
        assert max_abs(got.float().cpu(), torch.from_numpy(ref)) < 6e-2 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("mask", [(False, -1, -1), (True, -1, -1), (True, 300, 0)], ids=["full", "causal", "local_causal"])
@pytest.mark.parametrize("p,dtype", [(0.17, torch.bfloat16), (0.5, torch.bfloat16), (0.25, torch.float16)])
@pytest.mark.parametrize("B,Sq,Sk,H,Hk,D", [(2, 1024, 1024, 4, 4, 128), (1, 1500, 1700, 6, 2, 128), (1, 1700, 700, 2, 2, 128), (2, 2048, 2048, 4, 2, 64)])
def test_w64_dropout_forward(knobs, B, Sq, Sk, H, Hk, D, p, dtype, mask):
    """Round 5: dropout on the 64-rows-per-wave forward (fa_fwd_w64_kernel<.., dropout>; reference: flash_fwd_kernel.h:357-368 + dropout.h).  The random stream is a
    pure function of (rng_state, batch, head, row, key), so this kernel must drop exactly the pairs the lock-step kernel drops: the mask is taken from the lock-step
    kernel's return_softmax payload under the SAME seeded generator and fed to the fp64 oracle; the two kernels' outputs then differ by rounding only, the LSE is that
    of the un-dropped scores, and a backward driven by this forward's (out, lse, rng_state) meets the oracle's gradients (the backward kernels regenerate the mask)."""
    from flash_attn_amd import backend as be
    from oracle import attention_oracle as orc
    causal, wl, wr = mask
    torch.manual_seed(B * Sq + D)
    q = torch.randn(B, Sq, H, D, device="cuda", dtype=dtype)
    k = torch.randn(B, Sk, Hk, D, device="cuda", dtype=dtype)
    v, do = torch.randn_like(k), torch.randn_like(q)
    sc = D ** -0.5
    knobs.set("FA_FWD_NW", "8")
    torch.manual_seed(77)
    out8, lse8, rv, rng8 = be.fwd(q, k, v, None, None, p, sc, causal, wl, wr, 0.0, True, None)
    assert be.last_schedule()["fwd_kernel"] == 1
    knobs.set("FA_FWD_NW", "64")
    torch.manual_seed(77)
    out, lse, _, rng = be.fwd(q, k, v, None, None, p, sc, causal, wl, wr, 0.0, False, None)
    s = be.last_schedule()
    assert s["fwd_kernel"] == 3 and "dropout" in s["name"], s
    assert torch.equal(rng, rng8)
    torch.manual_seed(77)
    out_b, lse_b, _, _ = be.fwd(q, k, v, None, None, p, sc, causal, wl, wr, 0.0, False, None)
    assert torch.equal(out, out_b) and torch.equal(lse, lse_b), "same generator state => the same output"
    torch.manual_seed(77)
    be.fwd(q, k, v, None, None, p, sc, causal, wl, wr, 0.0, True, None)      # the random-byte output is not this kernel's: the lock-step kernel takes over
    assert be.last_schedule()["fwd_kernel"] == 1
    knobs.unset("FA_FWD_NW")
    keep = _keep(rv, p).cpu().numpy()
    o_ref, l_ref = orc.attention_fwd(q, k, v, sc, causal, (wl, wr), 0.0, None, p, keep)
    o_ref, l_ref = torch.from_numpy(o_ref).float(), torch.from_numpy(l_ref).float()
    assert torch.isfinite(out.float()).all()
    e64, e8 = max_abs(out.float().cpu(), o_ref), max_abs(out8.float().cpu(), o_ref)
    assert e64 < max(2 * e8, (1.2e-2 if dtype == torch.bfloat16 else 4e-3) / (1 - p)), (e64, e8)
    fin = torch.isfinite(l_ref)
    assert torch.equal(torch.isposinf(lse.cpu()), ~fin)
    assert max_abs(lse.cpu()[fin], l_ref[fin]) < (8e-3 if dtype == torch.bfloat16 else 2e-3)
    if Sq * Sk <= 1100 * 1100:   # (the fp64 backward oracle on the larger shapes takes minutes)
        dq, dk, dv, _ = be.bwd(do, q, k, v, out, lse, None, None, None, None, p, sc, causal, wl, wr, 0.0, False, None, rng)
        rq, rk, rvv, _ = orc.attention_bwd(do, q, k, v, out, lse, sc, causal, (wl, wr), 0.0, None, p, keep)
        for nm, got, ref in (("dq", dq, rq), ("dk", dk, rk), ("dv", dv, rvv)):
            gtol = (4e-2 if dtype == torch.bfloat16 else 8e-3) * max(1.0, float(np.abs(ref).max())) / (1 - p)
            assert max_abs(got.float().cpu(), torch.from_numpy(ref)) < gtol, nm


def test_w64_dropout_default_dispatch_and_varlen(knobs):
    from flash_attn_amd import backend as be
    import itertools
    q = torch.randn(2, 4096, 8, 128, device="cuda", dtype=torch.bfloat16)
    k, v = torch.randn_like(q), torch.randn_like(q)
    be.fwd(q, k, v, None, None, 0.1, 128 ** -0.5, True, -1, -1, 0.0, False, None)
    s = be.last_schedule()
    assert s["fwd_kernel"] == 3 and "dropout" in s["name"], s
    be.fwd(q, k, v, None, None, 0.1, 128 ** -0.5, True, -1, -1, 25.0, False, None)     # dropout AND softcap: the lock-step kernel
    assert be.last_schedule()["fwd_kernel"] == 1
    # packed batch: each sequence's rows and keys count from 0 in the random stream, as in the lock-step kernel (same mask => outputs equal to rounding)
    lens = [700, 33, 1500, 256, 1, 900]
    cu = torch.tensor([0] + list(itertools.accumulate(lens)), dtype=torch.int32, device="cuda")
    qq = torch.randn(sum(lens), 4, 128, device="cuda", dtype=torch.bfloat16)
    kk = torch.randn(sum(lens), 2, 128, device="cuda", dtype=torch.bfloat16)
    vv = torch.randn_like(kk)
    args = (qq, kk, vv, None, cu, cu, None, None, None, None, max(lens), max(lens), 0.3, 128 ** -0.5, False, True, -1, -1, 0.0, False, None)
    knobs.set("FA_FWD_NW", "64")
    torch.manual_seed(9)
    o64, l64 = be.varlen_fwd(*args)[:2]
    s = be.last_schedule()
    assert s["fwd_kernel"] == 3 and "dropout" in s["name"], s
    knobs.set("FA_FWD_NW", "8")
    torch.manual_seed(9)
    o8, l8 = be.varlen_fwd(*args)[:2]
    knobs.unset("FA_FWD_NW")
    assert max_abs(o64.float(), o8.float()) < 3e-2 and max_abs(l64, l8) < 8e-3
