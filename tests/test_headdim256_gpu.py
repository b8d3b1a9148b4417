"""GPU parity tests of head dimensions 129..256 (native 256-wide kernels: 4-wave workgroups, one wave per SIMD)."""
import numpy as np
import pytest
import torch

from tests._util import attention_torch, max_abs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from flash_attn_amd import backend
    return backend


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("d", [192, 256])
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
    dq, dk, dv, _ = be.bwd(do, q, k, v, out, lse, None, None, None, None, 0.0, scale, causal, window[0], window[1], 0.0, False, None, None)

    def ref(qq, kk, vv, dd, upcast):
        qq, kk, vv = (t.detach().clone().requires_grad_() for t in (qq, kk, vv))
        o, l = attention_torch(qq, kk, vv, causal, window, upcast=upcast, reorder=not upcast)
        return (o, l) + torch.autograd.grad(o, (qq, kk, vv), dd.to(o.dtype))

    o32, l32, q32, k32, v32 = ref(q.float(), k.float(), v.float(), do.float(), True)
    opt, _, qpt, kpt, vpt = ref(q, k, v, do, False)
    assert max_abs(out.float(), o32) <= 2 * max_abs(opt.float(), o32) + 1e-4
    fin = torch.isfinite(l32)
    assert max_abs(lse[fin], l32[fin]) < 2e-3 and torch.equal(torch.isposinf(lse), ~fin)
    for got, r, p_ in ((dq, q32, qpt), (dk, k32, kpt), (dv, v32, vpt)):
        assert max_abs(got.float(), r) <= 3 * max_abs(p_.float(), r) + 2e-4


def test_varlen_dropout_and_kvcache_at_256(be):
    from oracle import attention_oracle as orc
    torch.manual_seed(1)
    d, H, Hk = 256, 4, 2
    lens = [70, 200, 33]
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    q = torch.randn(sum(lens), H, d, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(sum(lens), Hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn_like(k)
    out, lse, rv, rng = be.varlen_fwd(q, k, v, None, cu, cu, None, None, None, None, max(lens), max(lens), 0.2, d ** -0.5, False, True,
                                      -1, -1, 0.0, True, None)
    do = torch.randn_like(q)
    dq, dk, dv, _ = be.varlen_bwd(do, q, k, v, out, lse, None, None, None, cu, cu, None, max(lens), max(lens), 0.2, d ** -0.5, False, True,
                                  -1, -1, 0.0, False, None, rng)
    for b, L in enumerate(lens):
        sl = slice(int(cu[b]), int(cu[b + 1]))
        keep = (rv[:, sl, :L].to(torch.int32) <= int(255 * 0.8)).cpu().numpy()[None]
        o_ref, _ = orc.attention_fwd(q[sl][None], k[sl][None], v[sl][None], None, True, (-1, -1), 0.0, None, 0.2, keep)
        assert max_abs(out[sl].float().cpu(), torch.from_numpy(o_ref[0])) < 3e-2
        rq, rk, rvv, _ = orc.attention_bwd(do[sl][None], q[sl][None], k[sl][None], v[sl][None], None, None, None, True, (-1, -1), 0.0, None, 0.2, keep)
        for got, ref in ((dq[sl], rq[0]), (dk[sl], rk[0]), (dv[sl], rvv[0])):
            assert max_abs(got.float().cpu(), torch.from_numpy(ref)) < 6e-2 * max(1.0, float(np.abs(ref).max()))
    # decode with split keys at head dim 256
    qd = torch.randn(2, 1, 8, d, device="cuda", dtype=torch.bfloat16)
    kc = torch.randn(2, 2048, 2, d, device="cuda", dtype=torch.bfloat16)
    vc = torch.randn_like(kc)
    cl = torch.tensor([2048, 513], dtype=torch.int32, device="cuda")
    o, l = be.fwd_kvcache(qd, kc, vc, None, None, cl, None, None, None, None, None, None, None, d ** -0.5, False, -1, -1, 0.0, True, 0)
    for b in range(2):
        L = int(cl[b])
        o_ref, l_ref = orc.attention_fwd(qd[b:b + 1], kc[b:b + 1, :L], vc[b:b + 1, :L])
        assert max_abs(o[b:b + 1].float().cpu(), torch.from_numpy(o_ref)) < 2e-2
        assert max_abs(l[b:b + 1].cpu(), torch.from_numpy(l_ref).float()) < 2e-3
