"""The reference's acceptance MATRICES, restated in our own words so that every GPU box runs them without the reference tree:

  test_output_matrix          tests/test_flash_attn.py:864-1132   (test_flash_attn_output: fixed-length batches, kv-packed or not, MHA / MQA / GQA, ALiBi, local,
                                                                   causal, softcap, deterministic, head dims 32 .. 256 incl. 40 / 59 / 111)
  test_varlen_output_matrix   tests/test_flash_attn.py:1135-1451  (test_flash_attn_varlen_output: the same through unpad -> varlen -> pad with random lengths)
  test_causal_matrix          tests/test_flash_attn.py:1454-1560  (test_flash_attn_causal: Sq != Sk, bottom-right alignment, local windows, both orders of the lengths)
  test_varlen_causal_matrix   tests/test_flash_attn.py:1563-1729  (test_flash_attn_varlen_causal, incl. the paged KV of its paged_kv_block_size parameter)
  test_kvcache_matrix         tests/test_flash_attn.py:1859-2140  (test_flash_attn_kvcache: append, rotary, paged, cache_batch_idx, leftpad, local, ALiBi, split-KV)
  test_zero_length_queries    tests/test_flash_attn_ck.py:1522-1560 (seqlen_q = 0 in a packed batch)

Each function draws a deterministic sample (seeded) of the reference's parameter grid -- the grids themselves are ~0.9 M cases (profiles/r04_reference_suite.txt
records the sampled runs of the reference's own files on this module).  Acceptance rule = the reference's: the error against an fp32 PyTorch attention is at most
2x (output) / 3x (gradients; kvcache: 3x, 5x with ALiBi, + 1e-5) the error of the same PyTorch attention computed in the input dtype.  The dropout cases of the
reference decode the CUDA kernels' S_dmask register layout and do not apply to this backend; dropout has its own tests (test_dropout_gpu.py).
The PyTorch attention below is written for this file (fp32 or input-dtype math, padding masks, left padding, ALiBi on true lengths, softcap, bottom-right windows)."""
import itertools
import math
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    import flash_attn_amd
    return flash_attn_amd


def _sample(n, seed, keep=None, **axes):
    names = list(axes)
    cases = [dict(zip(names, vals)) for vals in itertools.product(*[axes[k] for k in names])]
    if keep:
        cases = [c for c in cases if keep(c)]
    rng = random.Random(seed)
    picked = rng.sample(cases, min(n, len(cases)))
    return picked


def _cid(c):
    return "-".join(f"{k}={v if not isinstance(v, torch.dtype) else str(v)[6:]}" for k, v in c.items())


def _masked_pairs(sq, sk, window, qmask, kmask, leftpad, device):
    """True where (row, key) is hidden by the (bottom-right aligned) window; lengths are the TRUE lengths of each batch entry."""
    row = torch.arange(sq, device=device)[:, None]
    col = torch.arange(sk, device=device)[None, :].expand(1, 1, 1, sk)
    if leftpad is not None:
        lp = leftpad.view(-1, 1, 1, 1).long()
        col = torch.where(col >= lp, col - lp, torch.full_like(col.expand(lp.shape[0], 1, 1, sk), 2 ** 32))
    lk = sk if kmask is None else kmask.sum(-1).view(-1, 1, 1, 1)
    lq = sq if qmask is None else qmask.sum(-1).view(-1, 1, 1, 1)
    wl, wr = window
    if wl < 0:
        return col > row + lk - lq + wr
    lkt = torch.full_like(col, sk) if kmask is None else lk
    return (col > torch.minimum(row + lk - lq + wr, lkt)) | (col < row + lk - lq - wl)


def _alibi_bias(slopes, sq, sk, qmask, kmask, causal, leftpad=None):
    sl = slopes.view(slopes.shape[0], slopes.shape[1], 1, 1)
    dev = slopes.device
    if causal:
        return torch.arange(-sk + 1, 1, device=dev, dtype=torch.float32) * sl
    row = torch.arange(sq, device=dev)[:, None]
    col = torch.arange(sk, device=dev)[None, :].expand(1, 1, 1, sk)
    if leftpad is not None:
        lp = leftpad.view(-1, 1, 1, 1).long()
        col = torch.where(col >= lp, col - lp, torch.full_like(col.expand(lp.shape[0], 1, 1, sk), 2 ** 32))
    lk = sk if kmask is None else kmask.sum(-1).view(-1, 1, 1, 1)
    lq = sq if qmask is None else qmask.sum(-1).view(-1, 1, 1, 1)
    return -sl * (row + lk - lq - col).abs().float()


def _attention(q, k, v, qmask=None, kmask=None, bias=None, causal=False, window=(-1, -1), softcap=0.0, exact=True, leftpad=None):
    """exact=True: fp32 math (the yardstick); exact=False: the same attention in the input dtype with the scale applied to K -- the 'PyTorch' error it calibrates."""
    dt = q.dtype
    if causal:
        window = (window[0], 0)
    if exact:
        q, k, v = q.float(), k.float(), v.float()
    B, Sq, H, D = q.shape
    Sk, g = k.shape[1], H // k.shape[2]
    k, v = k.repeat_interleave(g, 2), v.repeat_interleave(g, 2)
    s = torch.einsum("bthd,bshd->bhts", q / math.sqrt(D), k) if exact else torch.einsum("bthd,bshd->bhts", q, k / math.sqrt(D))
    if softcap > 0:
        s = torch.tanh(s / softcap) * softcap
    if kmask is not None:
        s = s.masked_fill(~kmask[:, None, None, :], float("-inf"))
    hidden = None
    if window[0] >= 0 or window[1] >= 0:
        hidden = _masked_pairs(Sq, Sk, window, qmask, kmask, leftpad, q.device)
        s = s.masked_fill(hidden, float("-inf"))
    if bias is not None:
        s = s + bias
    p = torch.softmax(s, -1).to(v.dtype)
    if hidden is not None:
        p = p.masked_fill(hidden.all(-1, keepdim=True), 0.0)   # rows that see no key
    if qmask is not None:
        p = p.masked_fill(~qmask[:, None, :, None], 0.0)
    o = torch.einsum("bhts,bshd->bthd", p, v)
    if qmask is not None:
        o = o.masked_fill(~qmask[:, :, None, None], 0.0)
    return o.to(dt)


def _err(a, b):
    return float((a.float() - b.float()).abs().max()) if a.numel() else 0.0


def _heads(mha_type, h, gqa_kv):
    return h if mha_type == "mha" else (1 if mha_type == "mqa" else gqa_kv)


def _check_fwd_bwd(tag, out, out_ref, out_pt, leaves, g, rtol_out=2, rtol_grad=3):
    assert _err(out, out_ref) <= rtol_out * _err(out_pt, out_ref), (tag, "out", _err(out, out_ref), _err(out_pt, out_ref))
    got = torch.autograd.grad(out, leaves["ours"], g)
    ref = torch.autograd.grad(out_ref, leaves["ref"], g)
    pt = torch.autograd.grad(out_pt, leaves["ref"], g)
    for name, a, r, pp in zip(leaves["names"], got, ref, pt):
        assert _err(a, r) <= rtol_grad * _err(pp, r), (tag, name, _err(a, r), _err(pp, r))


SEQ_OUT = [(113, 203), (128, 217), (113, 211), (108, 256), (256, 512), (512, 256), (1024, 1024), (1023, 1024), (1024, 1023), (2048, 2048)]
DIMS = [32, 40, 59, 64, 96, 111, 128, 160, 192, 224, 256]

OUT_CASES = _sample(130, 1, kvpacked=[True, False], dtype=[torch.float16, torch.bfloat16], mha_type=["mha", "mqa", "gqa"], deterministic=[False, True],
                    alibi=[False, True], local=[False, True], causal=[False, True], d=DIMS, seq=SEQ_OUT, softcap=[0.0, 50.0])


@pytest.mark.parametrize("c", OUT_CASES, ids=_cid)
def test_output_matrix(fa, c):
    torch.manual_seed(0)
    (sq, sk), d, dtype, softcap = c["seq"], c["d"], c["dtype"], c["softcap"]
    B, H = 4, (6 if softcap == 0.0 else 4)
    Hk = _heads(c["mha_type"], H, 2)
    window = (-1, -1) if not c["local"] else tuple(int(x) for x in torch.randint(0, sk, (2,)))
    q = torch.randn(B, sq, H, d, device="cuda", dtype=dtype)
    if softcap > 0:
        q = q * softcap
    q.requires_grad_(True)
    slopes = torch.rand(B, H, device="cuda", dtype=torch.float32) * 0.3 if c["alibi"] else None
    bias = _alibi_bias(slopes, sq, sk, None, None, c["causal"]) if c["alibi"] else None
    kw = dict(causal=c["causal"], window_size=window, softcap=softcap, alibi_slopes=slopes, deterministic=c["deterministic"])
    if c["kvpacked"]:
        kv = torch.randn(B, sk, 2, Hk, d, device="cuda", dtype=dtype, requires_grad=True)
        out = fa.flash_attn_kvpacked_func(q, kv, 0.0, **kw)
        k, v = kv.unbind(2)
        leaves = dict(ours=(q, kv), ref=(q, kv), names=("dq", "dkv"))
    else:
        k = torch.randn(B, sk, Hk, d, device="cuda", dtype=dtype, requires_grad=True)
        v = torch.randn(B, sk, Hk, d, device="cuda", dtype=dtype, requires_grad=True)
        out = fa.flash_attn_func(q, k, v, 0.0, **kw)
        leaves = dict(ours=(q, k, v), ref=(q, k, v), names=("dq", "dk", "dv"))
    out_ref = _attention(q, k, v, None, None, bias, c["causal"], window, softcap, True)
    out_pt = _attention(q, k, v, None, None, bias, c["causal"], window, softcap, False)
    _check_fwd_bwd(_cid(c), out, out_ref, out_pt, leaves, torch.randn_like(out))


def _random_lengths(max_len, B, mode="random"):
    if mode == "full":
        lens = torch.full((B,), max_len, dtype=torch.int64)
    elif mode == "third":
        lens = torch.randint(max_len // 3, max_len + 1, (B,))
    else:
        lens = torch.randint(max(1, max_len - 20), max_len + 1, (B,))
    return (torch.arange(max_len)[None, :] < lens[:, None]).cuda()


def _cu(mask):
    lens = mask.sum(-1, dtype=torch.int32)
    return torch.nn.functional.pad(torch.cumsum(lens, 0, dtype=torch.int32), (1, 0)), int(lens.max())


VAR_CASES = _sample(130, 2, kvpacked=[True, False], dtype=[torch.float16, torch.bfloat16], mha_type=["mha", "mqa", "gqa"], deterministic=[False, True],
                    alibi=[False, True], local=[False, True], causal=[False, True], d=[32, 59, 64, 80, 96, 111, 128, 160, 192, 224, 256],
                    seq=[(1, 147), (113, 203), (128, 217), (113, 211), (108, 256), (256, 512), (512, 256), (1024, 1024), (1023, 1024), (1024, 1023), (2048, 2048)],
                    softcap=[0.0, 50.0])


@pytest.mark.parametrize("c", VAR_CASES, ids=_cid)
def test_varlen_output_matrix(fa, c):
    from flash_attn_amd.bert_padding import pad_input, unpad_input
    torch.manual_seed(0)
    (sq, sk), d, dtype, softcap = c["seq"], c["d"], c["dtype"], c["softcap"]
    B, H = 4, (6 if softcap == 0.0 else 4)
    Hk = _heads(c["mha_type"], H, 2)
    window = (-1, -1) if not c["local"] else tuple(int(x) for x in torch.randint(0, sk, (2,)))
    q = torch.randn(B, sq, H, d, device="cuda", dtype=dtype)
    if softcap > 0:
        q = q * softcap
    q.requires_grad_(True)
    k = torch.randn(B, sk, Hk, d, device="cuda", dtype=dtype, requires_grad=True)
    v = torch.randn(B, sk, Hk, d, device="cuda", dtype=dtype, requires_grad=True)
    qmask, kmask = _random_lengths(sq, B), _random_lengths(sk, B)
    slopes = torch.rand(B, H, device="cuda", dtype=torch.float32) * 0.3 if c["alibi"] else None
    bias = _alibi_bias(slopes, sq, sk, qmask, kmask, c["causal"]) if c["alibi"] else None
    q_u, idx_q, cu_q, max_q = unpad_input(q, qmask)[:4]
    k_u, _, cu_k, max_k = unpad_input(k, kmask)[:4]
    v_u = unpad_input(v, kmask)[0]
    kw = dict(causal=c["causal"], window_size=window, softcap=softcap, alibi_slopes=slopes, deterministic=c["deterministic"])
    if c["kvpacked"]:
        out_u = fa.flash_attn_varlen_kvpacked_func(q_u, torch.stack([k_u, v_u], 1), cu_q, cu_k, max_q, max_k, 0.0, **kw)
    else:
        out_u = fa.flash_attn_varlen_func(q_u, k_u, v_u, cu_q, cu_k, max_q, max_k, 0.0, **kw)
    out = pad_input(out_u, idx_q, B, sq)
    out_ref = _attention(q, k, v, qmask, kmask, bias, c["causal"], window, softcap, True)
    out_pt = _attention(q, k, v, qmask, kmask, bias, c["causal"], window, softcap, False)
    assert _err(out, out_ref) <= 2 * _err(out_pt, out_ref), (_err(out, out_ref), _err(out_pt, out_ref))
    g = torch.randn_like(out)
    got = torch.autograd.grad(out, (q, k, v), g)
    ref = torch.autograd.grad(out_ref, (q, k, v), g)
    pt = torch.autograd.grad(out_pt, (q, k, v), g)
    pad_q, pad_k = ~qmask[:, :, None, None], ~kmask[:, :, None, None]
    for name, a, r, pp, pad in zip(("dq", "dk", "dv"), got, ref, pt, (pad_q, pad_k, pad_k)):
        a, r, pp = a.masked_fill(pad, 0), r.masked_fill(pad, 0), pp.masked_fill(pad, 0)   # (padded rows carry no gradient in the packed call)
        assert _err(a, r) <= 3 * _err(pp, r), (name, _err(a, r), _err(pp, r))


CAUSAL_SEQ = [(1, 239), (3, 799), (127, 512), (127, 513), (113, 203), (128, 217), (113, 211), (108, 256), (256, 512), (1023, 1024)]
CAUSAL_CASES = _sample(90, 3, dtype=[torch.float16, torch.bfloat16], local=[False, True], d=[32, 40, 59, 64, 80, 96, 111, 128, 160, 192, 224, 256],
                       swap=[False, True], seq=CAUSAL_SEQ)


@pytest.mark.parametrize("c", CAUSAL_CASES, ids=_cid)
def test_causal_matrix(fa, c):
    torch.manual_seed(0)
    sq, sk = c["seq"][::-1] if c["swap"] else c["seq"]
    d, dtype = c["d"], c["dtype"]
    B, H = 8, 9
    window = (-1, -1) if not c["local"] else tuple(int(x) for x in torch.randint(0, sk, (2,)))
    q = torch.randn(B, sq, H, d, device="cuda", dtype=dtype, requires_grad=True)
    k = torch.randn(B, sk, H, d, device="cuda", dtype=dtype, requires_grad=True)
    v = torch.randn(B, sk, H, d, device="cuda", dtype=dtype, requires_grad=True)
    out = fa.flash_attn_func(q, k, v, 0.0, causal=True, window_size=window)
    out_ref = _attention(q, k, v, None, None, None, True, window, 0.0, True)
    out_pt = _attention(q, k, v, None, None, None, True, window, 0.0, False)
    assert _err(out, out_ref) <= 2 * _err(out_pt, out_ref) + 1e-5
    g = torch.randn_like(out)
    got, ref, pt = (torch.autograd.grad(o, (q, k, v), g) for o in (out, out_ref, out_pt))
    for name, a, r, pp in zip(("dq", "dk", "dv"), got, ref, pt):
        assert _err(a, r) <= 2 * _err(pp, r) + 1e-5, (name, _err(a, r), _err(pp, r))


def _paged(k, v, page, device):
    """Scatter (B, Sk, Hk, D) into a paged cache with a random block table (reference _generate_block_kvcache: pages in random order, spare pages)."""
    B, Sk, Hk, D = k.shape
    per = (Sk + page - 1) // page
    nblk = B * per * 3
    table = torch.randperm(nblk, device=device)[: B * per].reshape(B, per).to(torch.int32)
    kp = torch.randn(nblk, page, Hk, D, device=device, dtype=k.dtype)
    vp = torch.randn(nblk, page, Hk, D, device=device, dtype=k.dtype)
    for b in range(B):
        for j in range(per):
            n = min(page, Sk - j * page)
            kp[table[b, j], :n] = k[b, j * page: j * page + n]
            vp[table[b, j], :n] = v[b, j * page: j * page + n]
    return kp, vp, table


VC_CASES = _sample(80, 4, dtype=[torch.float16, torch.bfloat16], local=[False, True], d=[32, 64, 96, 128, 160, 192, 224, 256], swap=[False, True],
                   paged=[None, 256, 512], seq=[(1, 239), (3, 799), (127, 512), (127, 513), (113, 203), (128, 217), (113, 211), (108, 256), (256, 512), (1023, 1024)])


@pytest.mark.parametrize("c", VC_CASES, ids=_cid)
def test_varlen_causal_matrix(fa, c):
    from flash_attn_amd.bert_padding import pad_input, unpad_input
    torch.manual_seed(0)
    sq, sk = c["seq"][::-1] if c["swap"] else c["seq"]
    d, dtype, page = c["d"], c["dtype"], c["paged"]
    B, H = 8, 9
    window = (-1, -1) if not c["local"] else tuple(int(x) for x in torch.randint(0, sk, (2,)))
    q = torch.randn(B, sq, H, d, device="cuda", dtype=dtype, requires_grad=True)
    k = torch.randn(B, sk, H, d, device="cuda", dtype=dtype, requires_grad=True)
    v = torch.randn(B, sk, H, d, device="cuda", dtype=dtype, requires_grad=True)
    qmask, kmask = _random_lengths(sq, B), _random_lengths(sk, B)
    q_u, idx_q, cu_q, max_q = unpad_input(q, qmask)[:4]
    k_u, _, cu_k, max_k = unpad_input(k, kmask)[:4]
    v_u = unpad_input(v, kmask)[0]
    if page is None:
        out_u = fa.flash_attn_varlen_func(q_u, k_u, v_u, cu_q, cu_k, max_q, max_k, 0.0, causal=True, window_size=window)
    else:   # keys read from pages: the padded k / v scattered into a paged cache, lengths through cu_seqlens_k (forward only, as in the reference)
        kp, vp, table = _paged(k.detach(), v.detach(), page, "cuda")
        out_u = fa.flash_attn_varlen_func(q_u, kp, vp, cu_q, cu_k, max_q, max_k, 0.0, causal=True, window_size=window, block_table=table)
    out = pad_input(out_u, idx_q, B, sq)
    out_ref = _attention(q, k, v, qmask, kmask, None, True, window, 0.0, True)
    out_pt = _attention(q, k, v, qmask, kmask, None, True, window, 0.0, False)
    assert _err(out, out_ref) <= 2 * _err(out_pt, out_ref) + 1e-5, (_err(out, out_ref), _err(out_pt, out_ref))
    if page is None:
        g = torch.randn_like(out)
        got, ref, pt = (torch.autograd.grad(o, (q, k, v), g) for o in (out, out_ref, out_pt))
        pad_q, pad_k = ~qmask[:, :, None, None], ~kmask[:, :, None, None]
        for name, a, r, pp, pad in zip(("dq", "dk", "dv"), got, ref, pt, (pad_q, pad_k, pad_k)):
            a, r, pp = a.masked_fill(pad, 0), r.masked_fill(pad, 0), pp.masked_fill(pad, 0)
            assert _err(a, r) <= 2 * _err(pp, r) + 1e-5, (name, _err(a, r), _err(pp, r))


def _rotate(x, cos, sin, offsets, interleaved, per_token):
    """Rotary embedding of x (B, S, H, D) at positions offsets[b] (+ s when per_token), first 2 * cos.shape[1] channels (reference flash_attn/layers/rotary.py)."""
    B, S, H, D = x.shape
    rd = 2 * cos.shape[1]
    pos = offsets.view(B, 1).long() + (torch.arange(S, device=x.device)[None, :] if per_token else 0)
    pos = pos.expand(B, S)
    c, s = cos.float()[pos][:, :, None, :], sin.float()[pos][:, :, None, :]
    xf, out = x.float(), x.float().clone()
    if interleaved:
        x1, x2 = xf[..., 0:rd:2], xf[..., 1:rd:2]
        out[..., 0:rd:2], out[..., 1:rd:2] = x1 * c - x2 * s, x1 * s + x2 * c
    else:
        x1, x2 = xf[..., : rd // 2], xf[..., rd // 2: rd]
        out[..., : rd // 2], out[..., rd // 2: rd] = x1 * c - x2 * s, x1 * s + x2 * c
    return out.to(x.dtype)


def _kv_keep(c):
    sq, sk = c["seq"]
    if sq > sk and c["new_kv"]:
        return False
    if not c["new_kv"] and c["rot"] > 0.0:
        return False
    if (c["batch_idx"] or c["leftpad"]) and c["paged"] is not None:
        return False
    return True


KV_CASES = _sample(130, 5, keep=_kv_keep, num_splits=[1, 0], mha_type=["mha", "mqa", "gqa"], new_kv=[False, True], alibi=[False, True], local=[False, True],
                   causal=[False, True], new_eq_q=[True, False], interleaved=[False, True], rot=[0.0, 0.5, 1.0], paged=[None, 256], leftpad=[False, True],
                   batch_idx=[False, True], d=[32, 59, 64, 80, 128, 256],
                   seq=[(1, 128), (1, 339), (3, 1024), (64, 800), (64, 256), (3, 799), (64, 2048), (16, 20000), (1, 128 * 1024), (16, 128 * 1024), (128, 128)])


@pytest.mark.parametrize("c", KV_CASES, ids=_cid)
def test_kvcache_matrix(fa, c):
    torch.manual_seed(0)
    (sq, sk), d, dtype = c["seq"], c["d"], torch.float16
    B, H = 2, 6
    Bc = B * 2 if c["batch_idx"] else B
    Hk = _heads(c["mha_type"], H, 3)
    rd = int(c["rot"] * d) // 16 * 16
    window = (-1, -1) if not c["local"] else tuple(int(x) for x in torch.randint(0, sk, (2,)))
    q = torch.randn(B, sq, H, d, device="cuda", dtype=dtype)
    s_new = sq if c["new_eq_q"] else int(torch.randint(1, sq + 1, (1,)))
    k_new = torch.randn(B, s_new, Hk, d, device="cuda", dtype=dtype) if c["new_kv"] else None
    v_new = torch.randn(B, s_new, Hk, d, device="cuda", dtype=dtype) if c["new_kv"] else None
    kc = torch.randn(Bc, sk, Hk, d, device="cuda", dtype=dtype)
    vc = torch.randn(Bc, sk, Hk, d, device="cuda", dtype=dtype)
    if c["paged"] is not None:
        kp, vp, table = _paged(kc, vc, c["paged"], "cuda")
    hi = (sk - (sq if (c["causal"] or c["local"]) and rd > 1 else s_new) + 1) if c["new_kv"] else (sk + 1)
    lens = torch.randint(0 if c["new_kv"] else 1, hi, (B,), dtype=torch.int32, device="cuda")
    leftpad = None
    if c["leftpad"]:
        leftpad = torch.cat([torch.randint(0, int(lens[i]), (1,), dtype=torch.int32, device="cuda") if int(lens[i]) > 0
                             else torch.zeros(1, dtype=torch.int32, device="cuda") for i in range(B)])
    ar = torch.arange(sk, device="cuda")[None, :]
    kmask = ar < (lens[:, None] + (s_new if c["new_kv"] else 0))
    if leftpad is not None:
        kmask = kmask & (ar >= leftpad[:, None])
    bidx = torch.randperm(Bc, dtype=torch.int32, device="cuda")[:B] if c["batch_idx"] else None
    slopes = torch.rand(B, H, device="cuda", dtype=torch.float32) * 0.3 if c["alibi"] else None
    bias = _alibi_bias(slopes, sq, sk, None, kmask, c["causal"], leftpad) if c["alibi"] else None
    cos = sin = None
    q_ro, k_ro = q, k_new
    if rd > 0:
        ang = torch.rand(sk if c["paged"] is None else kp.shape[0] * c["paged"], rd // 2, device="cuda") * 2 * math.pi
        cos, sin = torch.cos(ang).to(dtype), torch.sin(ang).to(dtype)
        q_ro = _rotate(q, cos, sin, lens, c["interleaved"], c["causal"] or c["local"])
        k_ro = _rotate(k_new, cos, sin, lens, c["interleaved"], True)
    k_ref = (kc if bidx is None else kc[bidx.long()]).clone()
    v_ref = (vc if bidx is None else vc[bidx.long()]).clone()
    if c["new_kv"]:
        upd = (lens[:, None] <= ar) & (ar < lens[:, None] + s_new)
        k_ref[upd] = k_ro.reshape(-1, Hk, d)
        v_ref[upd] = v_new.reshape(-1, Hk, d)
    out = fa.flash_attn_with_kvcache(q, kc if c["paged"] is None else kp, vc if c["paged"] is None else vp, k_new, v_new, rotary_cos=cos, rotary_sin=sin,
                                     cache_seqlens=lens, cache_batch_idx=bidx, cache_leftpad=leftpad, block_table=None if c["paged"] is None else table,
                                     causal=c["causal"], window_size=window, rotary_interleaved=c["interleaved"], alibi_slopes=slopes,
                                     num_splits=c["num_splits"])
    out_ref = _attention(q_ro, k_ref, v_ref, None, kmask, bias, c["causal"], window, 0.0, True, leftpad)
    out_pt = _attention(q_ro, k_ref, v_ref, None, kmask, bias, c["causal"], window, 0.0, False, leftpad)
    if c["new_kv"]:   # the cache was updated in place (rotated keys to rounding, values exactly)
        if c["paged"] is None:
            k_sel, v_sel = (kc if bidx is None else kc[bidx.long()]), (vc if bidx is None else vc[bidx.long()])
        else:
            k_sel = kp[table.long().flatten()].reshape(B, -1, Hk, d)[:, :sk]
            v_sel = vp[table.long().flatten()].reshape(B, -1, Hk, d)[:, :sk]
        assert torch.allclose(k_sel, k_ref, rtol=1e-3, atol=1e-3)
        assert torch.equal(v_sel, v_ref)
    mult = 3 if not c["alibi"] else 5
    assert _err(out, out_ref) <= mult * _err(out_pt, out_ref) + 1e-5, (_err(out, out_ref), _err(out_pt, out_ref))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("sk", [8, 256])
def test_zero_length_queries(fa, sk, d, causal, dtype):
    """A packed batch whose every sequence has no query rows: nothing to compute, nothing to fault on; out has no rows, dk / dv are zeros."""
    torch.manual_seed(0)
    B, H = 4, 6
    q = torch.randn(0, H, d, device="cuda", dtype=dtype, requires_grad=True)
    k = torch.randn(B * sk, H, d, device="cuda", dtype=dtype, requires_grad=True)
    v = torch.randn(B * sk, H, d, device="cuda", dtype=dtype, requires_grad=True)
    cu_q = torch.zeros(B + 1, dtype=torch.int32, device="cuda")
    cu_k = torch.arange(0, (B + 1) * sk, sk, dtype=torch.int32, device="cuda")
    out = fa.flash_attn_varlen_func(q, k, v, cu_q, cu_k, 0, sk, 0.0, causal=causal)
    assert out.shape == (0, H, d)
    dq, dk, dv = torch.autograd.grad(out, (q, k, v), torch.randn_like(out))
    assert dq.shape == q.shape and not dk.any() and not dv.any()
