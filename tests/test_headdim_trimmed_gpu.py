"""GPU parity tests of head dimensions 32 / 96 / 192: the "trimmed" variants of the 64 / 128 / 256 kernels (csrc/fa_fwd.hip,
fa_bwd.hip: same LDS pitch, DV/16 k-steps, DV/32 output blocks).  The tensors go to the kernels as they are -- no padded
copies -- which the NaN-guard tests prove: memory right behind each head's DV columns holds NaN, and a kernel that read a
column >= DV into its arithmetic, or stored one, would show it.

Reference being matched: the reference builds these head dims natively too (csrc/flash_attn/src/static_switch.h:92-110,
flash_fwd_launch_template.h:195-299, flash_bwd_launch_template.h:136-285) and its acceptance suite sweeps them
(tests/test_flash_attn.py:903 `d` list).  Tolerances: the reference's rule, <= 2x (forward) / 3x (gradients) the error of
a same-dtype PyTorch implementation against fp32, LSE to 2e-3 absolute.
"""
import numpy as np
import pytest
import torch

from tests._util import attention_torch, max_abs

pytestmark = pytest.mark.gpu

TRIMMED = [32, 96, 192]


@pytest.fixture(scope="module")
def be():
    from flash_attn_amd import backend
    return backend


def _ref(q, k, v, do, causal, window, upcast):
    qq, kk, vv = (t.detach().clone().requires_grad_() for t in (q, k, v))
    o, l = attention_torch(qq, kk, vv, causal, window, upcast=upcast, reorder=not upcast)
    return (o, l) + torch.autograd.grad(o, (qq, kk, vv), do.to(o.dtype))


def _check(out, lse, grads, q, k, v, do, causal, window):
    o32, l32, q32, k32, v32 = _ref(q.float(), k.float(), v.float(), do.float(), causal, window, True)
    opt, _, qpt, kpt, vpt = _ref(q, k, v, do, causal, window, False)
    assert max_abs(out.float(), o32) <= 2 * max_abs(opt.float(), o32) + 1e-4
    fin = torch.isfinite(l32)
    assert max_abs(lse[fin], l32[fin]) < 2e-3 and torch.equal(torch.isposinf(lse), ~fin)
    for got, r, p_ in zip(grads, (q32, k32, v32), (qpt, kpt, vpt)):
        assert max_abs(got.float(), r) <= 3 * max_abs(p_.float(), r) + 2e-4


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("d", TRIMMED)
@pytest.mark.parametrize("mode", ["full", "causal", "local"])
@pytest.mark.parametrize("sq,sk,h,hk", [(113, 203, 4, 4), (256, 512, 6, 2), (1024, 1024, 2, 1), (1, 300, 4, 2), (384, 129, 4, 4)])
def test_fwd_bwd_vs_fp32_reference(be, sq, sk, h, hk, mode, d, dtype):
    torch.manual_seed(0)
    B = 2
    q = torch.randn(B, sq, h, d, device="cuda", dtype=dtype)
    k = torch.randn(B, sk, hk, d, device="cuda", dtype=dtype)
    v = torch.randn(B, sk, hk, d, device="cuda", dtype=dtype)
    do = torch.randn(B, sq, h, d, device="cuda", dtype=dtype)
    causal = mode == "causal"
    window = (37, 50) if mode == "local" else (-1, -1)
    scale = d ** -0.5
    out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, scale, causal, window[0], window[1], 0.0, False, None)
    sched = be.last_schedule()
    # the trimmed instantiation ran (its name carries the true head dim), on the tensors as given
    assert sched["fwd_kernel"] == 1 and sched["d"] == d and f",{d},4,feat0,lockstep>" in sched["name"], sched
    dq, dk, dv, _ = be.bwd(do, q, k, v, out, lse, None, None, None, None, 0.0, scale, causal, window[0], window[1], 0.0, False, None, None)
    assert be.last_schedule()["bwd_dq_nw"] == 4
    _check(out, lse, (dq, dk, dv), q, k, v, do, causal, window)


def _guarded(shape, d, dtype, gen_scale=1.0):
    """A (..., d) view whose rows are followed in memory by NaN columns: base tensor (..., d + 32) full of NaN."""
    base = torch.full(shape[:-1] + (d + 32,), float("nan"), device="cuda", dtype=dtype)
    view = base[..., :d]
    view.copy_(torch.randn(shape, device="cuda", dtype=dtype) * gen_scale)
    return base, view


@pytest.mark.parametrize("d", TRIMMED)
@pytest.mark.parametrize("causal", [False, True])
def test_columns_past_the_head_dim_are_never_touched(d, causal):
    """Inputs are strided views with NaN right behind every head; gradients are written into NaN-guarded buffers too."""
    import flash_attn_2_cuda as ext
    torch.manual_seed(1)
    B, sq, sk, h, hk = 2, 200, 333, 4, 2
    dtype = torch.bfloat16
    _, q = _guarded((B, sq, h, d), d, dtype)
    _, k = _guarded((B, sk, hk, d), d, dtype)
    _, v = _guarded((B, sk, hk, d), d, dtype)
    _, do = _guarded((B, sq, h, d), d, dtype)
    scale = d ** -0.5
    out, lse = ext.fwd(q, k, v, None, None, 0.0, scale, causal, -1, -1, 0.0, False, None)[:2]
    assert torch.isfinite(out).all() and torch.isfinite(lse).all()
    oc, lc = ext.fwd(q.contiguous(), k.contiguous(), v.contiguous(), None, None, 0.0, scale, causal, -1, -1, 0.0, False, None)[:2]
    assert torch.equal(out, oc) and torch.equal(lse, lc)
    dqb, dq = _guarded((B, sq, h, d), d, dtype)
    dkb, dk = _guarded((B, sk, hk, d), d, dtype)
    dvb, dv = _guarded((B, sk, hk, d), d, dtype)
    ext.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, scale, causal, -1, -1, 0.0, False, None, None)
    for base, view in ((dqb, dq), (dkb, dk), (dvb, dv)):
        assert torch.isfinite(view).all()
        assert torch.isnan(base[..., d:]).all()   # nothing was stored past the head dim
    c = ext.bwd(do.contiguous(), q.contiguous(), k.contiguous(), v.contiguous(), out, lse, None, None, None, None, 0.0, scale, causal, -1, -1,
                0.0, False, None, None)
    assert torch.equal(dq, c[0]) and torch.equal(dk, c[1]) and torch.equal(dv, c[2])


@pytest.mark.parametrize("d", TRIMMED)
def test_varlen_equals_per_sequence_and_features(be, d):
    from oracle import attention_oracle as orc
    torch.manual_seed(2)
    H, Hk = 4, 2
    lens_q, lens_k = [70, 1, 200, 33], [90, 64, 200, 257]
    cq = torch.tensor([0] + list(np.cumsum(lens_q)), dtype=torch.int32, device="cuda")
    ck = torch.tensor([0] + list(np.cumsum(lens_k)), dtype=torch.int32, device="cuda")
    q = torch.randn(sum(lens_q), H, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(sum(lens_k), Hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn_like(k)
    do = torch.randn_like(q)
    scale = d ** -0.5
    out, lse = be.varlen_fwd(q, k, v, None, cq, ck, None, None, None, None, max(lens_q), max(lens_k), 0.0, scale, False, True, -1, -1, 0.0,
                             False, None)[:2]
    dq, dk, dv, _ = be.varlen_bwd(do, q, k, v, out, lse, None, None, None, cq, ck, None, max(lens_q), max(lens_k), 0.0, scale, False, True,
                                  -1, -1, 0.0, False, None, None)
    for b in range(len(lens_q)):
        qs, ks = slice(int(cq[b]), int(cq[b + 1])), slice(int(ck[b]), int(ck[b + 1]))
        o1, l1 = be.fwd(q[qs][None], k[ks][None], v[ks][None], None, None, 0.0, scale, True, -1, -1, 0.0, False, None)[:2]
        assert torch.equal(out[qs], o1[0]) and torch.equal(lse[:, qs], l1[0])
        g = be.bwd(do[qs][None], q[qs][None], k[ks][None], v[ks][None], o1, l1, None, None, None, None, 0.0, scale, True, -1, -1, 0.0, False,
                   None, None)
        assert torch.equal(dq[qs], g[0][0]) and torch.equal(dk[ks], g[1][0]) and torch.equal(dv[ks], g[2][0])
    # softcap + ALiBi (run-time-checked all-features variant) against the fp64 oracle, forward and backward
    B, S = 2, 160
    q4 = torch.randn(B, S, H, d, device="cuda", dtype=torch.bfloat16)
    k4 = torch.randn(B, S, Hk, d, device="cuda", dtype=torch.bfloat16)
    v4 = torch.randn_like(k4)
    do4 = torch.randn_like(q4)
    slopes = torch.tensor([0.5, 0.25, 0.125, 0.0625], device="cuda", dtype=torch.float32)
    for softcap, alibi in ((15.0, None), (0.0, slopes), (15.0, slopes)):
        o, l = be.fwd(q4, k4, v4, None, alibi, 0.0, scale, True, -1, -1, softcap, False, None)[:2]
        assert be.last_schedule()["fwd_feat"] == 7 and be.last_schedule()["d"] == d
        g = be.bwd(do4, q4, k4, v4, o, l, None, None, None, alibi, 0.0, scale, True, -1, -1, softcap, False, None, None)
        a = None if alibi is None else alibi.cpu().numpy().astype(np.float64)
        o_ref, l_ref = orc.attention_fwd(q4.float().cpu().numpy(), k4.float().cpu().numpy(), v4.float().cpu().numpy(), scale, True, (-1, -1),
                                         softcap=softcap, alibi_slopes=a)
        gq, gk, gv, _ = orc.attention_bwd(do4.float().cpu().numpy(), q4.float().cpu().numpy(), k4.float().cpu().numpy(),
                                          v4.float().cpu().numpy(), None, None, scale, True, (-1, -1), softcap=softcap, alibi_slopes=a)
        assert max_abs(o.float().cpu(), torch.from_numpy(o_ref).float()) < 2e-2
        assert max_abs(l.cpu(), torch.from_numpy(l_ref).float()) < 2e-3
        for got, r in zip(g[:3], (gq, gk, gv)):
            r = torch.from_numpy(r).float()
            assert max_abs(got.float().cpu(), r) < 2e-2 * max(1.0, float(r.abs().max()))


@pytest.mark.parametrize("d", TRIMMED)
def test_dropout_replays_in_the_backward(be, d):
    from oracle import attention_oracle as orc
    torch.manual_seed(3)
    B, S, H = 2, 130, 2
    q = torch.randn(B, S, H, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn_like(q); v = torch.randn_like(q); do = torch.randn_like(q)
    p, scale = 0.25, d ** -0.5
    out, lse, rv, rng = be.fwd(q, k, v, None, None, p, scale, True, -1, -1, 0.0, True, None)
    dq, dk, dv, _ = be.bwd(do, q, k, v, out, lse, None, None, None, None, p, scale, True, -1, -1, 0.0, False, None, rng)
    keep = (rv.cpu().numpy() <= int(np.floor((1.0 - p) * 255.0)))
    f = lambda t: t.float().cpu().numpy()
    o_ref, _ = orc.attention_fwd(f(q), f(k), f(v), scale, True, (-1, -1), dropout_p=p, dropout_mask=keep)
    gq, gk, gv, _ = orc.attention_bwd(f(do), f(q), f(k), f(v), None, None, scale, True, (-1, -1), dropout_p=p, dropout_mask=keep)
    assert max_abs(out.float().cpu(), torch.from_numpy(o_ref).float()) < 3e-2
    for got, r in ((dq, gq), (dk, gk), (dv, gv)):
        r = torch.from_numpy(r).float()
        assert max_abs(got.float().cpu(), r) < 3e-2 * max(1.0, float(r.abs().max()))


@pytest.mark.parametrize("binder", ["ext", "ctypes"])
@pytest.mark.parametrize("d", TRIMMED)
@pytest.mark.parametrize("sq,causal,num_splits", [(1, False, 0), (1, False, 7), (5, True, 1), (77, True, 0)])
def test_kvcache_decode_append_and_split(binder, d, sq, causal, num_splits):
    """fwd_kvcache at the trimmed head dims: in-place append, cache_batch_idx, GQA head packing at Sq = 1, split-KV merge
    (partial rows have the 64 / 128 / 256 pitch), against the fp64 oracle."""
    from oracle import attention_oracle as orc
    if binder == "ext":
        import flash_attn_2_cuda as kv
    else:
        from flash_attn_amd import backend as kv
    torch.manual_seed(4)
    B, H, hk, Scache, Bc = 3, 8, 2, 1100, 4
    q = torch.randn(B, sq, H, d, device="cuda", dtype=torch.bfloat16)
    kc = torch.randn(Bc, Scache, hk, d, device="cuda", dtype=torch.bfloat16)
    vc = torch.randn_like(kc)
    kn = torch.randn(B, sq, hk, d, device="cuda", dtype=torch.bfloat16)
    vn = torch.randn_like(kn)
    lens = torch.tensor([0, 333, Scache - sq], dtype=torch.int32, device="cuda")
    idx = torch.tensor([3, 0, 2], dtype=torch.int32, device="cuda")
    out, lse = kv.fwd_kvcache(q, kc, vc, kn, vn, lens, None, None, idx, None, None, None, None, d ** -0.5, causal, -1, -1, 0.0, True, num_splits)
    for b in range(B):
        r, L = int(idx[b]), int(lens[b])
        assert torch.equal(kc[r, L:L + sq], kn[b]) and torch.equal(vc[r, L:L + sq], vn[b])
    f = lambda t: t.float().cpu().numpy()
    for b in range(B):
        L = int(lens[b]) + sq
        o_ref, l_ref = orc.attention_fwd(f(q[b:b + 1]), f(kc[int(idx[b])][None, :L]), f(vc[int(idx[b])][None, :L]), None, causal, (-1, -1))
        assert max_abs(out[b:b + 1].float().cpu(), torch.from_numpy(o_ref).float()) < 2e-2
        assert max_abs(lse[b:b + 1].cpu(), torch.from_numpy(l_ref).float()) < 2e-3


def test_interface_pads_only_to_the_next_native_size():
    """flash_attn_func with head dims between the native sizes: 40 -> 64, 72 / 80 -> 96, 160 -> 192 (not 128 / 256 as before)."""
    from flash_attn_amd import flash_attn_interface as fi, backend as be
    torch.manual_seed(5)
    for d, dn in ((40, 64), (80, 96), (160, 192), (24, 32)):
        q = torch.randn(1, 130, 2, d, device="cuda", dtype=torch.bfloat16, requires_grad=True)
        k = torch.randn(1, 130, 2, d, device="cuda", dtype=torch.bfloat16, requires_grad=True)
        v = torch.randn(1, 130, 2, d, device="cuda", dtype=torch.bfloat16, requires_grad=True)
        out = fi.flash_attn_func(q, k, v, causal=True)
        assert be.last_schedule()["d"] == dn
        o32, _ = attention_torch(q.float(), k.float(), v.float(), True)
        opt, _ = attention_torch(q, k, v, True, upcast=False, reorder=True)
        assert max_abs(out.float(), o32) <= 2 * max_abs(opt.float(), o32) + 1e-4
        out.sum().backward()
        assert q.grad.shape == q.shape and torch.isfinite(q.grad).all()


@pytest.mark.parametrize("binder", ["ext", "ctypes"])
@pytest.mark.parametrize("d", [8, 40, 72, 80, 104, 160, 224])
@pytest.mark.parametrize("sq,causal,num_splits,paged", [(1, False, 0, False), (1, False, 5, True), (3, True, 1, False), (70, True, 0, True)])
def test_kvcache_head_dims_between_the_built_sizes(binder, d, sq, causal, num_splits, paged):
    """fwd_kvcache takes any head dim that is a multiple of 8 (reference mha_fwd_kvcache, flash_api.cpp:1300-1310, predicates the
    columns in-kernel).  Here the next built kernel runs with a run-time column bound (FwdK::d_chunks): chunks behind the head dim
    read as zeros -- the memory there is the NEXT head's data, so a kernel that read it would miss the oracle -- and are never
    stored (the output rows are followed by the next head's output)."""
    from oracle import attention_oracle as orc
    if binder == "ext":
        import flash_attn_2_cuda as kv
    else:
        from flash_attn_amd import backend as kv
    from flash_attn_amd import backend as be
    torch.manual_seed(6)
    B, H, hk, cap = 2, 4, 2, 1024
    q = torch.randn(B, sq, H, d, device="cuda", dtype=torch.bfloat16)
    kn = torch.randn(B, sq, hk, d, device="cuda", dtype=torch.bfloat16)
    vn = torch.randn_like(kn)
    lens = torch.tensor([cap - sq, 130], dtype=torch.int32, device="cuda")
    if paged:
        page, per = 256, cap // 256
        kc = torch.randn(B * per + 1, page, hk, d, device="cuda", dtype=torch.bfloat16)
        vc = torch.randn_like(kc)
        table = torch.randperm(B * per + 1, device="cuda")[: B * per].reshape(B, per).to(torch.int32)
    else:
        kc = torch.randn(B, cap, hk, d, device="cuda", dtype=torch.bfloat16)
        vc = torch.randn_like(kc)
        table = None
    out_buf = torch.full((B, sq, H, d), float("nan"), device="cuda", dtype=torch.bfloat16)
    out, lse = kv.fwd_kvcache(q, kc, vc, kn, vn, lens, None, None, None, None, table, None, out_buf, d ** -0.5, causal, -1, -1, 0.0, True, num_splits)
    s = be.last_schedule()
    dk = next(n for n in (32, 64, 96, 128, 192, 256) if d <= n)
    assert s["d"] == dk and s["fwd_kernel"] == 1, s
    k_log = kc[table.long()].reshape(B, cap, hk, d) if paged else kc
    v_log = vc[table.long()].reshape(B, cap, hk, d) if paged else vc
    f = lambda t: t.float().cpu().numpy()
    for b in range(B):
        L = int(lens[b]) + sq
        assert torch.equal(k_log[b, L - sq:L], kn[b]) and torch.equal(v_log[b, L - sq:L], vn[b])
        o_ref, l_ref = orc.attention_fwd(f(q[b:b + 1]), f(k_log[b:b + 1, :L]), f(v_log[b:b + 1, :L]), None, causal, (-1, -1))
        assert torch.isfinite(out[b]).all()
        assert max_abs(out[b:b + 1].float().cpu(), torch.from_numpy(o_ref).float()) < 2e-2
        assert max_abs(lse[b:b + 1].cpu(), torch.from_numpy(l_ref).float()) < 2e-3


BETWEEN = [40, 72, 80, 104, 160, 224]


@pytest.mark.parametrize("d", BETWEEN)
@pytest.mark.parametrize("mode", ["full", "causal", "local"])
def test_training_head_dims_between_the_built_sizes_without_copies(be, d, mode):
    """Head dims between the built sizes TRAIN without padded copies (round 3): the next built size's kernels run with a run-time column bound
    (FwdK / BwdK::d_chunks) in the forward, the delta pre-pass, the dK/dV kernel and the 4-wave dQ kernel -- the reference rounds internally
    the same way (flash_api.cpp:458,872; Is_even_K).  Inputs are strided views with NaN right behind every head's d columns, the gradients
    are written into NaN-guarded buffers: a kernel that read a column >= d into its arithmetic, or stored one, would show it; results equal
    those on contiguous tensors bit for bit and meet the 2x / 3x rule against fp32."""
    import flash_attn_2_cuda as ext
    torch.manual_seed(3)
    B, sq, sk, h, hk = 2, 200, 333, 4, 2
    dtype = torch.bfloat16
    causal = mode == "causal"
    window = (37, 50) if mode == "local" else (-1, -1)
    _, q = _guarded((B, sq, h, d), d, dtype)
    _, k = _guarded((B, sk, hk, d), d, dtype)
    _, v = _guarded((B, sk, hk, d), d, dtype)
    _, do = _guarded((B, sq, h, d), d, dtype)
    scale = d ** -0.5
    ob, out = _guarded((B, sq, h, d), d, dtype)
    out_r, lse = ext.fwd(q, k, v, out, None, 0.0, scale, causal, window[0], window[1], 0.0, False, None)[:2]
    assert out_r.data_ptr() == out.data_ptr() and torch.isfinite(out).all() and torch.isnan(ob[..., d:]).all()
    oc, lc = ext.fwd(q.contiguous(), k.contiguous(), v.contiguous(), None, None, 0.0, scale, causal, window[0], window[1], 0.0, False, None)[:2]
    assert torch.equal(out, oc) and torch.equal(lse, lc)
    dqb, dq = _guarded((B, sq, h, d), d, dtype)
    dkb, dk = _guarded((B, sk, hk, d), d, dtype)
    dvb, dv = _guarded((B, sk, hk, d), d, dtype)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base_mem = torch.cuda.memory_allocated()
    ext.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, scale, causal, window[0], window[1], 0.0, False, None, None)
    torch.cuda.synchronize()
    # no padded copies of dout / q / k / v / out and no padded gradient buffers: the call allocates softmax_d (+ nothing of tensor size)
    assert torch.cuda.max_memory_allocated() - base_mem < q.numel() * 2 // 2, torch.cuda.max_memory_allocated() - base_mem
    for base, view in ((dqb, dq), (dkb, dk), (dvb, dv)):
        assert torch.isfinite(view).all()
        assert torch.isnan(base[..., d:]).all()   # nothing was stored past the head dim
    c = ext.bwd(do.contiguous(), q.contiguous(), k.contiguous(), v.contiguous(), out.contiguous(), lse, None, None, None, None, 0.0, scale, causal,
                window[0], window[1], 0.0, False, None, None)
    assert torch.equal(dq, c[0]) and torch.equal(dk, c[1]) and torch.equal(dv, c[2])
    _check(out.contiguous(), lse, (dq.contiguous(), dk.contiguous(), dv.contiguous()), q.contiguous(), k.contiguous(), v.contiguous(), do.contiguous(), causal, window)


@pytest.mark.parametrize("d", [40, 160])
def test_between_head_dims_varlen_and_features(be, d):
    """Same head dims through the varlen entry points with softcap + ALiBi + dropout replay (the run-time-checked all-features kernels)."""
    from oracle import attention_oracle as orc
    torch.manual_seed(4)
    H, Hk = 4, 2
    lens_q, lens_k = [70, 1, 200, 33], [90, 64, 200, 257]
    cq = torch.tensor([0] + list(np.cumsum(lens_q)), dtype=torch.int32, device="cuda")
    ck = torch.tensor([0] + list(np.cumsum(lens_k)), dtype=torch.int32, device="cuda")
    q = torch.randn(sum(lens_q), H, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(sum(lens_k), Hk, d, device="cuda", dtype=torch.bfloat16)
    v, do = torch.randn_like(k), torch.randn_like(q)
    scale = d ** -0.5
    alibi = torch.rand(H, device="cuda") * 0.3
    out, lse = be.varlen_fwd(q, k, v, None, cq, ck, None, None, None, alibi, max(lens_q), max(lens_k), 0.0, scale, False, True, -1, -1, 15.0,
                             False, None)[:2]
    dq, dk, dv, _ = be.varlen_bwd(do, q, k, v, out, lse, None, None, None, cq, ck, alibi, max(lens_q), max(lens_k), 0.0, scale, False, True,
                                  -1, -1, 15.0, False, None, None)
    o_ref, _ = orc.varlen_fwd(q, k, v, cq.cpu().numpy(), ck.cpu().numpy(), scale, True, (-1, -1), 15.0, alibi.cpu().numpy())
    g_ref = orc.varlen_bwd(do, q, k, v, cq.cpu().numpy(), ck.cpu().numpy(), scale, True, (-1, -1), 15.0, alibi.cpu().numpy())
    assert float(np.abs(out.float().cpu().numpy() - o_ref).max()) < 2e-2
    for got, ref in zip((dq, dk, dv), g_ref[:3]):
        assert float(np.abs(got.float().cpu().numpy() - ref).max()) < 6e-2
