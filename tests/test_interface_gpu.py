"""GPU tests of the drop-in boundary: the flash_attn_func / flash_attn_varlen_func autograd path over the
flash_attn_2_cuda torch extension, and extension == ctypes binder bit-for-bit (same C ABI underneath)."""
import numpy as np
import pytest
import torch

from tests._util import attention_torch, max_abs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fi():
    from flash_attn_amd import flash_attn_interface
    assert flash_attn_interface.flash_attn_gpu.__name__ == "flash_attn_2_cuda", "C++ extension must be the loaded backend on the GPU box"
    return flash_attn_interface


@pytest.mark.parametrize("d", [40, 59, 64, 96, 111, 128, 160, 192, 224, 256])
@pytest.mark.parametrize("causal", [False, True])
def test_flash_attn_func_autograd(fi, d, causal):
    torch.manual_seed(0)
    B, Sq, Sk, H, Hk = 2, 217, 333, 6, 2
    q = torch.randn(B, Sq, H, d, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    k = torch.randn(B, Sk, Hk, d, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    v = torch.randn(B, Sk, Hk, d, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    do = torch.randn(B, Sq, H, d, device="cuda", dtype=torch.bfloat16)
    out = fi.flash_attn_func(q, k, v, causal=causal)
    assert out.shape == (B, Sq, H, d)
    dq, dk, dv = torch.autograd.grad(out, (q, k, v), do)
    qf, kf, vf = (t.detach().float().requires_grad_() for t in (q, k, v))
    ref, _ = attention_torch(qf, kf, vf, causal, upcast=True)
    rq, rk, rv = torch.autograd.grad(ref, (qf, kf, vf), do.float())
    qb, kb, vb = (t.detach().clone().requires_grad_() for t in (q, k, v))
    pt, _ = attention_torch(qb, kb, vb, causal, upcast=False, reorder=True)
    pq, pk, pv = torch.autograd.grad(pt, (qb, kb, vb), do)
    assert max_abs(out.float(), ref) <= 2 * max_abs(pt.float(), ref) + 1e-5
    for got, r, p_ in ((dq, rq, pq), (dk, rk, pk), (dv, rv, pv)):
        assert max_abs(got.float(), r) <= 3 * max_abs(p_.float(), r) + 1e-4


def test_packed_variants_and_return_lse(fi):
    torch.manual_seed(1)
    qkv = torch.randn(2, 130, 3, 4, 64, device="cuda", dtype=torch.float16, requires_grad=True)
    out, lse, _ = fi.flash_attn_qkvpacked_func(qkv, causal=True, return_attn_probs=True)
    ref, lse_ref = attention_torch(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], True)
    assert max_abs(out.float(), ref.float()) < 5e-3 and max_abs(lse, lse_ref) < 2e-3
    (g,) = torch.autograd.grad(out, qkv, torch.ones_like(out))
    assert g.shape == qkv.shape and not torch.isnan(g).any()
    kv = torch.randn(2, 77, 2, 2, 64, device="cuda", dtype=torch.float16)
    q = torch.randn(2, 50, 4, 64, device="cuda", dtype=torch.float16)
    o2 = fi.flash_attn_kvpacked_func(q, kv, window_size=(10, 5))
    r2, _ = attention_torch(q, kv[:, :, 0], kv[:, :, 1], False, (10, 5))
    assert max_abs(o2.float(), r2.float()) < 5e-3


def test_varlen_func_autograd_matches_padded(fi, knobs):
    knobs.set("FA_FWD_NW", "34")
    torch.manual_seed(2)
    lens = [33, 128, 1, 200]
    H, Hk, d = 4, 4, 128
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    q = torch.randn(sum(lens), H, d, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    k = torch.randn(sum(lens), Hk, d, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    v = torch.randn(sum(lens), Hk, d, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    do = torch.randn_like(q)
    out = fi.flash_attn_varlen_func(q, k, v, cu, cu, max(lens), max(lens), causal=True)
    dq, dk, dv = torch.autograd.grad(out, (q, k, v), do)
    for b in range(len(lens)):
        s0, s1 = int(cu[b]), int(cu[b + 1])
        qq, kk, vv = (t[None, s0:s1].detach().clone().requires_grad_() for t in (q, k, v))
        o1 = fi.flash_attn_func(qq, kk, vv, causal=True)
        g = torch.autograd.grad(o1, (qq, kk, vv), do[None, s0:s1])
        assert torch.equal(out[s0:s1], o1[0])
        assert torch.equal(dq[s0:s1], g[0][0]) and torch.equal(dk[s0:s1], g[1][0]) and torch.equal(dv[s0:s1], g[2][0])


def test_extension_and_ctypes_binders_agree_bitwise():
    import flash_attn_2_cuda as ext
    from flash_attn_amd import backend as cty
    torch.manual_seed(3)
    q = torch.randn(2, 300, 8, 128, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(2, 411, 2, 128, device="cuda", dtype=torch.bfloat16)
    v = torch.randn_like(k)
    do = torch.randn_like(q)
    sc = 128 ** -0.5
    a = ext.fwd(q, k, v, None, None, 0.0, sc, True, -1, -1, 0.0, False, None)
    b = cty.fwd(q, k, v, None, None, 0.0, sc, True, -1, -1, 0.0, False, None)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    ga = ext.bwd(do, q, k, v, a[0], a[1], None, None, None, None, 0.0, sc, True, -1, -1, 0.0, False, None, None)
    gb = cty.bwd(do, q, k, v, b[0], b[1], None, None, None, None, 0.0, sc, True, -1, -1, 0.0, False, None, None)
    for x, y in zip(ga, gb):
        assert torch.equal(x, y)


def test_extension_error_messages(fi):
    import flash_attn_2_cuda as ext
    q = torch.randn(1, 8, 2, 64, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="block_table"):
        ext.varlen_fwd(q[0], q[0], q[0], None, torch.tensor([0, 8], dtype=torch.int32, device="cuda"),
                       torch.tensor([0, 8], dtype=torch.int32, device="cuda"), None, None, torch.zeros(1, 1, dtype=torch.int32, device="cuda"),
                       None, 8, 8, 0.0, 0.125, False, False, -1, -1, 0.0, False, None)
    with pytest.raises(RuntimeError, match="num_splits > 1 is not supported"):
        ext.varlen_fwd(q[0], q[0], q[0], None, torch.tensor([0, 8], dtype=torch.int32, device="cuda"),
                       torch.tensor([0, 8], dtype=torch.int32, device="cuda"), None, None, None, None, 8, 8, 0.0, 0.125, False, False,
                       -1, -1, 0.0, False, None, 2)
    with pytest.raises(RuntimeError):
        ext.fwd(q.float(), q.float(), q.float(), None, None, 0.0, 0.125, False, -1, -1, 0.0, False, None)


def test_torch_compile_fullgraph_and_opcheck(fi):
    """The registered custom ops trace without graph breaks (fullgraph) through forward and backward and agree with eager;
    torch.library.opcheck validates schema, fake implementation and autograd registration of the forward op
    (reference precedent: hopper/test_torch_compile_and_export.py)."""
    torch.manual_seed(3)
    q = torch.randn(2, 128, 4, 64, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    k = torch.randn(2, 160, 2, 64, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    v = torch.randn(2, 160, 2, 64, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    g = torch.randn(2, 128, 4, 64, device="cuda", dtype=torch.bfloat16)

    def f(q, k, v):
        return fi.flash_attn_func(q, k, v, causal=True, window_size=(64, 0)) * 2.0

    eager = f(q, k, v)
    ge = torch.autograd.grad(eager, (q, k, v), g)
    compiled = torch.compile(f, backend="aot_eager", fullgraph=True)
    out = compiled(q, k, v)
    gc = torch.autograd.grad(out, (q, k, v), g)
    assert torch.equal(out, eager)
    for a, b in zip(gc, ge):
        assert torch.equal(a, b)
    torch.library.opcheck(torch.ops.flash_attn_amd._flash_attn_forward.default,
                          (q.detach(), k.detach(), v.detach(), 0.0, 0.125, True, -1, -1, 0.0, None, False),
                          test_utils=("test_schema", "test_faketensor"))


@pytest.mark.parametrize("varlen", [False, True])
def test_single_query_row_grouped_heads_swap(varlen):
    """seqlen_q == 1 with grouped heads takes the head-packing path of mha_fwd / mha_varlen_fwd (flash_api.cpp:431, :622):
    results (out, LSE shapes and values) equal the per-head computation, through both binders."""
    import flash_attn_2_cuda as ext
    from flash_attn_amd import backend as be
    from oracle import attention_oracle as orc
    torch.manual_seed(5)
    B, H, Hk, D = 3, 8, 2, 128
    lens = [300, 77, 513]
    q = torch.randn(B, 1, H, D, device="cuda", dtype=torch.bfloat16)
    if not varlen:
        Sk = 333
        k = torch.randn(B, Sk, Hk, D, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k)
        for mod in (be, ext):
            out, lse = mod.fwd(q, k, v, None, None, 0.0, D ** -0.5, True, -1, -1, 0.0, False, None)[:2]
            assert out.shape == (B, 1, H, D) and lse.shape == (B, H, 1)
            ref, lse_ref = orc.attention_fwd(q, k, v, None, False)
            assert float((out.float().cpu() - torch.from_numpy(ref)).abs().max()) < 2e-2
            assert float((lse.cpu() - torch.from_numpy(lse_ref)).abs().max()) < 2e-3
    else:
        cu_k = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
        cu_q = torch.arange(0, B + 1, dtype=torch.int32, device="cuda")
        k = torch.randn(sum(lens), Hk, D, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k)
        for mod in (be, ext):
            out, lse = mod.varlen_fwd(q[:, 0], k, v, None, cu_q, cu_k, None, None, None, None, 1, max(lens), 0.0, D ** -0.5, False, False,
                                      -1, -1, 0.0, False, None)[:2]
            assert out.shape == (B, H, D) and lse.shape == (H, B)
            for b in range(B):
                ks = k[None, int(cu_k[b]):int(cu_k[b + 1])]; vs = v[None, int(cu_k[b]):int(cu_k[b + 1])]
                ref, lse_ref = orc.attention_fwd(q[b:b + 1], ks, vs, None, False)
                assert float((out[b].float().cpu() - torch.from_numpy(ref)[0, 0]).abs().max()) < 2e-2
                assert float((lse[:, b].cpu() - torch.from_numpy(lse_ref)[0, :, 0]).abs().max()) < 2e-3


def test_first_calls_from_two_threads_and_per_thread_schedule_report():
    """Launcher state under threads, in a fresh process so that these ARE the first calls: the dynamic-LDS attribute of every
    kernel instantiation is set on first use (per device, lock-free: fa_launch.h ensure_dyn_lds) and fa_last_schedule /
    fa_last_error are thread-local.  Two threads start at a barrier, one on head dim 64 and one on 128 (kernels with > 64 KB
    of dynamic LDS), each on its own stream; both must match the single-threaded results bit for bit and each must read its
    own schedule report."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, threading
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
from flash_attn_amd import backend as be
torch.manual_seed(0)
shapes = {64: (2, 1024, 4, 64), 128: (2, 1024, 4, 128)}
inp = {d: [torch.randn(*s, device="cuda", dtype=torch.bfloat16) for _ in range(4)] for d, s in shapes.items()}
res, sched, errs = {}, {}, []
bar = threading.Barrier(2)
def work(d):
    try:
        q, k, v, do = inp[d]
        st = torch.cuda.Stream()
        bar.wait()
        with torch.cuda.stream(st):
            for _ in range(3):
                out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, d ** -0.5, True, -1, -1, 0.0, False, None)
                s1 = be.last_schedule()
                g = be.bwd(do, q, k, v, out, lse, None, None, None, None, 0.0, d ** -0.5, True, -1, -1, 0.0, False, None, None)
        st.synchronize()
        res[d] = (out, lse) + tuple(g[:3]); sched[d] = s1
    except Exception as e:  # noqa
        errs.append(repr(e))
ts = [threading.Thread(target=work, args=(d,)) for d in (64, 128)]
[t.start() for t in ts]; [t.join() for t in ts]
assert not errs, errs
for d in (64, 128):
    assert sched[d]["d"] == d, sched
    q, k, v, do = inp[d]
    out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, d ** -0.5, True, -1, -1, 0.0, False, None)
    g = be.bwd(do, q, k, v, out, lse, None, None, None, None, 0.0, d ** -0.5, True, -1, -1, 0.0, False, None, None)
    for a, b in zip(res[d], (out, lse) + tuple(g[:3])):
        assert torch.equal(a, b)
print("threads ok")
''' % (root, os.path.join(root, "flash-attention_amd"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "threads ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("varlen", [False, True])
@pytest.mark.parametrize("d", [64, 80, 59])
def test_packed_backward_writes_one_packed_gradient(fi, varlen, d):
    """The four packed entry points (reference FlashAttn{,Varlen}{QKV,KV}PackedFunc, flash_attn_interface.py:461-825): gradients equal those of
    the unpacked call bit for bit, come back as ONE packed tensor (dq / dk / dv are views of it), and the backward allocates the packed
    gradient and nothing of its size beside it -- no slice-backward temporaries (three zero-filled packed tensors + adds)."""
    torch.manual_seed(5)
    H, Hk = 4, 4
    if varlen:
        lens = [70, 1, 333, 128]
        cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
        qkv = torch.randn(sum(lens), 3, H, d, device="cuda", dtype=torch.bfloat16, requires_grad=True)
        call_p = lambda x: fi.flash_attn_varlen_qkvpacked_func(x, cu, max(lens), causal=True)
        call_u = lambda q, k, v: fi.flash_attn_varlen_func(q, k, v, cu, cu, max(lens), max(lens), causal=True)
        call_kv = lambda q, kv: fi.flash_attn_varlen_kvpacked_func(q, kv, cu, cu, max(lens), max(lens), causal=True)
    else:
        qkv = torch.randn(2, 300, 3, H, d, device="cuda", dtype=torch.bfloat16, requires_grad=True)
        call_p = lambda x: fi.flash_attn_qkvpacked_func(x, causal=True)
        call_u = lambda q, k, v: fi.flash_attn_func(q, k, v, causal=True)
        call_kv = lambda q, kv: fi.flash_attn_kvpacked_func(q, kv, causal=True)
    out = call_p(qkv)
    g = torch.randn_like(out)
    (dqkv,) = torch.autograd.grad(out, qkv, g)
    assert dqkv.shape == qkv.shape
    q, k, v = (t.detach().clone().requires_grad_() for t in qkv.unbind(-3))
    out_u = call_u(q, k, v)
    dq, dk, dv = torch.autograd.grad(out_u, (q, k, v), g)
    assert torch.equal(out, out_u)
    for i, t in enumerate((dq, dk, dv)):
        assert torch.equal(dqkv.select(-3, i), t)
    # kv-packed: dq separate, dkv one tensor
    qs = qkv.detach().select(-3, 0).clone().requires_grad_()
    kv = qkv.detach()[..., 1:, :, :].clone().requires_grad_()
    out_kv = call_kv(qs, kv)
    dq2, dkv = torch.autograd.grad(out_kv, (qs, kv), g)
    assert torch.equal(out_kv, out) and torch.equal(dq2, dq) and torch.equal(dkv.select(-3, 0), dk) and torch.equal(dkv.select(-3, 1), dv)
    # memory: the backward's peak above its starting point is the packed gradient (+ softmax_d, + padded copies only for head dims that
    # are not a built size) -- far below the 3 extra packed tensors of the slice-backward path
    if d == 64:
        out = call_p(qkv)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        (dqkv,) = torch.autograd.grad(out, qkv, g)
        torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated() - base
        packed_bytes = qkv.numel() * qkv.element_size()
        assert peak < 1.5 * packed_bytes + (1 << 20), (peak, packed_bytes)
