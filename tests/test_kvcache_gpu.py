"""GPU parity tests of the KV-cache inference path (fwd_kvcache): contiguous / batch-indexed / paged caches,
in-place append, decode (one query row, GQA head packing) and chunked prefill, against the fp64 oracle."""
import numpy as np
import pytest
import torch

from tests._util import max_abs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["ext", "ctypes"])
def kv(request):
    if request.param == "ext":
        import flash_attn_2_cuda as m
    else:
        from flash_attn_amd import backend as m
    return m


def _oracle(q, kc, vc, lens, causal, window=(-1, -1)):
    from oracle import attention_oracle as orc
    outs, lses = [], []
    for b in range(q.shape[0]):
        L = int(lens[b])
        o, l = orc.attention_fwd(q[b:b + 1], kc[b:b + 1, :L], vc[b:b + 1, :L], None, causal, window)
        outs.append(o); lses.append(l)
    return np.concatenate(outs), np.concatenate(lses)


@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("sq,causal", [(1, False), (1, True), (5, True), (77, True), (300, False)])
@pytest.mark.parametrize("hk", [1, 2, 8])
def test_kvcache_contiguous_with_append_and_batch_idx(kv, d, sq, causal, hk):
    torch.manual_seed(0)
    B, H, Scache, Bc, snew = 3, 8, 700, 5, sq
    q = torch.randn(B, sq, H, d, device="cuda", dtype=torch.bfloat16)
    kc = torch.randn(Bc, Scache, hk, d, device="cuda", dtype=torch.bfloat16)
    vc = torch.randn(Bc, Scache, hk, d, device="cuda", dtype=torch.bfloat16)
    kn = torch.randn(B, snew, hk, d, device="cuda", dtype=torch.bfloat16)
    vn = torch.randn(B, snew, hk, d, device="cuda", dtype=torch.bfloat16)
    lens = torch.tensor([0, 333, 700 - snew], dtype=torch.int32, device="cuda")
    idx = torch.tensor([4, 0, 2], dtype=torch.int32, device="cuda")
    kc0, vc0 = kc.clone(), vc.clone()
    out, lse = kv.fwd_kvcache(q, kc, vc, kn, vn, lens, None, None, idx, None, None, None, None, d ** -0.5, causal, -1, -1, 0.0, True, 0)
    # the cache was updated in place at the right rows, and nowhere else
    for b in range(B):
        r, L = int(idx[b]), int(lens[b])
        assert torch.equal(kc[r, L:L + snew], kn[b]) and torch.equal(vc[r, L:L + snew], vn[b])
        assert torch.equal(kc[r, :L], kc0[r, :L]) and torch.equal(kc[r, L + snew:], kc0[r, L + snew:])
    for r in (1, 3):
        assert torch.equal(kc[r], kc0[r]) and torch.equal(vc[r], vc0[r])
    o_ref, l_ref = _oracle(q, kc[idx.long()], vc[idx.long()], (lens + snew).cpu().numpy(), causal)
    assert out.shape == q.shape and lse.shape == (B, H, sq)
    assert max_abs(out.float().cpu(), torch.from_numpy(o_ref)) < 2e-2
    assert max_abs(lse.cpu(), torch.from_numpy(l_ref).float()) < 2e-3


@pytest.mark.parametrize("sq,causal", [(1, False), (64, True)])
def test_kvcache_paged(kv, sq, causal):
    torch.manual_seed(1)
    B, H, hk, d, page, nblk_per = 4, 8, 2, 128, 256, 3
    num_blocks = B * nblk_per + 2
    kp = torch.randn(num_blocks, page, hk, d, device="cuda", dtype=torch.bfloat16)
    vp = torch.randn(num_blocks, page, hk, d, device="cuda", dtype=torch.bfloat16)
    perm = torch.randperm(num_blocks, device="cuda")[: B * nblk_per].reshape(B, nblk_per).to(torch.int32)
    lens = torch.tensor([1, 256, 300, 768 - sq], dtype=torch.int32, device="cuda")
    q = torch.randn(B, sq, H, d, device="cuda", dtype=torch.bfloat16)
    kn = torch.randn(B, sq, hk, d, device="cuda", dtype=torch.bfloat16)
    vn = torch.randn(B, sq, hk, d, device="cuda", dtype=torch.bfloat16)
    out, lse = kv.fwd_kvcache(q, kp, vp, kn, vn, lens, None, None, None, None, perm, None, None, d ** -0.5, causal, -1, -1, 0.0, True, 0)
    # gather the logical caches back from the pages
    kc = kp[perm.long()].reshape(B, nblk_per * page, hk, d)
    vc = vp[perm.long()].reshape(B, nblk_per * page, hk, d)
    for b in range(B):
        L = int(lens[b])
        assert torch.equal(kc[b, L:L + sq], kn[b]) and torch.equal(vc[b, L:L + sq], vn[b])
    o_ref, l_ref = _oracle(q, kc, vc, (lens + sq).cpu().numpy(), causal)
    assert max_abs(out.float().cpu(), torch.from_numpy(o_ref)) < 2e-2
    assert max_abs(lse.cpu(), torch.from_numpy(l_ref).float()) < 2e-3


def test_kvcache_interface_and_errors():
    from flash_attn_amd import flash_attn_interface as fi
    torch.manual_seed(2)
    q = torch.randn(2, 1, 8, 128, device="cuda", dtype=torch.float16)
    kc = torch.randn(2, 512, 2, 128, device="cuda", dtype=torch.float16)
    vc = torch.randn_like(kc)
    out, lse = fi.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=300, return_softmax_lse=True)
    o_ref, l_ref = _oracle(q, kc, vc, [300, 300], False)
    assert max_abs(out.float().cpu(), torch.from_numpy(o_ref)) < 5e-3 and max_abs(lse.cpu(), torch.from_numpy(l_ref).float()) < 2e-3
    out_w = fi.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=300, window_size=(100, 0))
    ow_ref, _ = _oracle(q, kc, vc, [300, 300], False, (100, 0))
    assert max_abs(out_w.float().cpu(), torch.from_numpy(ow_ref)) < 5e-3
    with pytest.raises(RuntimeError, match="rotary"):   # rotary needs new keys to append (flash_api.cpp:1455)
        fi.flash_attn_with_kvcache(q, kc, vc, rotary_cos=torch.zeros(8, 16, device="cuda", dtype=torch.float16),
                                   rotary_sin=torch.zeros(8, 16, device="cuda", dtype=torch.float16))
    with pytest.raises(RuntimeError, match="divisible by 256"):
        fi.flash_attn_with_kvcache(q, kc[:, :128].contiguous(), vc[:, :128].contiguous(), block_table=torch.zeros(2, 1, dtype=torch.int32, device="cuda"))
    kn = torch.randn(2, 1, 2, 128, device="cuda", dtype=torch.float16)
    with pytest.raises(RuntimeError, match="cos/sin seqlen must be at least"):   # flash_api.cpp:1470
        fi.flash_attn_with_kvcache(q, kc, vc, k=kn, v=kn.clone(), cache_seqlens=300, rotary_cos=torch.zeros(8, 16, device="cuda", dtype=torch.float16),
                                   rotary_sin=torch.zeros(8, 16, device="cuda", dtype=torch.float16))
    with pytest.raises(RuntimeError, match="seqlen <= the seqlen of the KV cache"):   # flash_api.cpp:1397
        qq = torch.randn(2, 9, 8, 128, device="cuda", dtype=torch.float16)
        k9 = torch.randn(2, 9, 2, 128, device="cuda", dtype=torch.float16)
        fi.flash_attn_with_kvcache(qq, kc[:, :8], vc[:, :8], k=k9, v=k9.clone(), cache_seqlens=0)


@pytest.mark.parametrize("num_splits", [0, 1, 2, 7, 64])
@pytest.mark.parametrize("sq,causal,paged", [(1, False, False), (1, False, True), (6, True, False), (33, True, True)])
def test_kvcache_split_kv(kv, sq, causal, paged, num_splits):
    """Split-KV decode: every split count (0 = heuristic) gives the unsplit result up to fp32 merge rounding, including
    splits that hold no key of a short sequence."""
    torch.manual_seed(3)
    B, H, hk, d, cap = 2, 16, 4, 128, 4096
    q = torch.randn(B, sq, H, d, device="cuda", dtype=torch.bfloat16)
    lens = torch.tensor([cap - 7, 300], dtype=torch.int32, device="cuda")
    if paged:
        page, per = 256, cap // 256
        kc = torch.randn(B * per + 3, page, hk, d, device="cuda", dtype=torch.bfloat16)
        vc = torch.randn_like(kc)
        table = torch.randperm(B * per + 3, device="cuda")[: B * per].reshape(B, per).to(torch.int32)
        k_log, v_log = kc[table.long()].reshape(B, cap, hk, d), vc[table.long()].reshape(B, cap, hk, d)
    else:
        kc = torch.randn(B, cap, hk, d, device="cuda", dtype=torch.bfloat16)
        vc = torch.randn_like(kc)
        table, k_log, v_log = None, kc, vc
    out, lse = kv.fwd_kvcache(q, kc, vc, None, None, lens, None, None, None, None, table, None, None, d ** -0.5, causal, -1, -1, 0.0, True,
                              num_splits)
    o_ref, l_ref = _oracle(q, k_log, v_log, lens.cpu().numpy(), causal)
    assert max_abs(out.float().cpu(), torch.from_numpy(o_ref)) < 2e-2
    assert max_abs(lse.cpu(), torch.from_numpy(l_ref).float()) < 2e-3


def _rotary_ref(x, cos, sin, offsets, interleaved, per_token):
    """apply_rotary_emb semantics (reference flash_attn/layers/rotary.py:25-60 + seqlen_offsets), fp32 math."""
    B, S, H, D = x.shape
    rd = 2 * cos.shape[1]
    xf = x.float().cpu()
    out = xf.clone()
    for b in range(B):
        pos = int(offsets[b]) + (torch.arange(S) if per_token else torch.zeros(S, dtype=torch.long))
        c = cos.float().cpu()[pos][:, None, :]
        s = sin.float().cpu()[pos][:, None, :]
        xr = xf[b, :, :, :rd]
        if interleaved:
            x1, x2 = xr[..., 0::2], xr[..., 1::2]
            out[b, :, :, 0:rd:2] = x1 * c - x2 * s
            out[b, :, :, 1:rd:2] = x1 * s + x2 * c
        else:
            x1, x2 = xr[..., : rd // 2], xr[..., rd // 2:]
            out[b, :, :, : rd // 2] = x1 * c - x2 * s
            out[b, :, :, rd // 2: rd] = x1 * s + x2 * c
    return out.to(x.dtype)


@pytest.mark.parametrize("interleaved", [False, True])
@pytest.mark.parametrize("rotary_frac", [0.25, 1.0])
@pytest.mark.parametrize("sq,causal", [(1, False), (9, True), (9, False)])
@pytest.mark.parametrize("d", [64, 128])
def test_kvcache_rotary(kv, d, sq, causal, rotary_frac, interleaved):
    torch.manual_seed(4)
    B, H, hk, cap = 3, 8, 2, 512
    rd = int(rotary_frac * d) // 16 * 16
    q = torch.randn(B, sq, H, d, device="cuda", dtype=torch.bfloat16)
    kc = torch.randn(B, cap, hk, d, device="cuda", dtype=torch.bfloat16)
    vc = torch.randn_like(kc)
    kn = torch.randn(B, sq, hk, d, device="cuda", dtype=torch.bfloat16)
    vn = torch.randn_like(kn)
    lens = torch.tensor([0, 100, cap - sq], dtype=torch.int32, device="cuda")
    ang = torch.rand(cap, rd // 2, device="cuda") * 2 * np.pi
    cos, sin = torch.cos(ang).bfloat16(), torch.sin(ang).bfloat16()
    kc_ref = kc.clone()
    out, lse = kv.fwd_kvcache(q, kc, vc, kn, vn, lens, cos, sin, None, None, None, None, None, d ** -0.5, causal, -1, -1, 0.0, interleaved, 0)
    k_ro = _rotary_ref(kn, cos, sin, lens.cpu(), interleaved, True)
    q_ro = _rotary_ref(q, cos, sin, lens.cpu(), interleaved, causal)
    for b in range(B):
        L = int(lens[b])
        assert max_abs(kc[b, L:L + sq].float().cpu(), k_ro[b].float()) < 4e-2   # one bf16 ulp of |x| <= 4 (fma vs mul+sub)
        assert torch.equal(vc[b, L:L + sq], vn[b])
        kc_ref[b, L:L + sq] = kc[b, L:L + sq]                                    # attend over exactly what the cache holds
    assert (kc[:, :, :, rd:] == kc_ref[:, :, :, rd:]).all()
    o_ref, l_ref = _oracle(q_ro, kc_ref, vc, (lens + sq).cpu().numpy(), causal)
    assert max_abs(out.float().cpu(), torch.from_numpy(o_ref)) < 3e-2
    assert max_abs(lse.cpu(), torch.from_numpy(l_ref).float()) < 3e-2


@pytest.mark.parametrize("sq,causal", [(1, False), (5, True), (40, False)])
def test_kvcache_leftpad(kv, sq, causal):
    torch.manual_seed(6)
    B, H, hk, d, cap = 3, 8, 4, 128, 640
    q = torch.randn(B, sq, H, d, device="cuda", dtype=torch.bfloat16)
    kc = torch.randn(B, cap, hk, d, device="cuda", dtype=torch.bfloat16)
    vc = torch.randn_like(kc)
    kn = torch.randn(B, sq, hk, d, device="cuda", dtype=torch.bfloat16)
    vn = torch.randn_like(kn)
    lens = torch.tensor([70, 333, cap - sq], dtype=torch.int32, device="cuda")      # counted from row 0 (includes the pad)
    pad = torch.tensor([0, 200, 65], dtype=torch.int32, device="cuda")
    out, lse = kv.fwd_kvcache(q, kc, vc, kn, vn, lens, None, None, None, pad, None, None, None, d ** -0.5, causal, -1, -1, 0.0, True, 0)
    for b in range(B):
        L, P = int(lens[b]), int(pad[b])
        assert torch.equal(kc[b, L:L + sq], kn[b])
        o_ref, l_ref = _oracle(q[b:b + 1], kc[b:b + 1, P:], vc[b:b + 1, P:], [L + sq - P], causal)
        assert max_abs(out[b:b + 1].float().cpu(), torch.from_numpy(o_ref)) < 2e-2
        assert max_abs(lse[b:b + 1].cpu(), torch.from_numpy(l_ref).float()) < 2e-3


def test_kvcache_paged_capacity_guard_and_single_row_window(kv):
    torch.manual_seed(7)
    d, page = 64, 256
    kp = torch.randn(4, page, 1, d, device="cuda", dtype=torch.float16)
    vp = torch.randn_like(kp)
    table = torch.zeros(1, 1, dtype=torch.int32, device="cuda")
    q = torch.randn(1, 1, 1, d, device="cuda", dtype=torch.float16)
    over = torch.full((1,), page + 1, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError, match="block_table"):   # reference tests/test_flash_attn.py:2587-2634
        kv.fwd_kvcache(q, kp, vp, None, None, over, None, None, None, None, table, None, None, d ** -0.5, False, -1, -1, 0.0, True, 0)
    full = torch.full((1,), page, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError, match="block_table"):
        kv.fwd_kvcache(q, kp, vp, q, q, full, None, None, None, None, table, None, None, d ** -0.5, False, -1, -1, 0.0, True, 0)
    out, _ = kv.fwd_kvcache(q, kp, vp, None, None, full, None, None, None, None, table, None, None, d ** -0.5, False, -1, -1, 0.0, True, 0)
    assert out.shape == (1, 1, 1, d) and not out.isnan().any()
    # one query row, GQA, a right window bound: the bound hides nothing (the head-packing swap must not apply it to heads)
    q = torch.randn(2, 1, 16, d, device="cuda", dtype=torch.float16)
    kc = torch.randn(2, 300, 2, d, device="cuda", dtype=torch.float16)
    vc = torch.randn_like(kc)
    lens = torch.tensor([300, 77], dtype=torch.int32, device="cuda")
    out, lse = kv.fwd_kvcache(q, kc, vc, None, None, lens, None, None, None, None, None, None, None, d ** -0.5, False, -1, 3, 0.0, True, 0)
    o_ref, l_ref = _oracle(q, kc, vc, lens.cpu().numpy(), False)
    assert max_abs(out.float().cpu(), torch.from_numpy(o_ref)) < 5e-3 and max_abs(lse.cpu(), torch.from_numpy(l_ref).float()) < 2e-3


@pytest.mark.parametrize("causal", [False, True])
def test_varlen_forward_with_paged_kv_and_with_leftpad(kv, causal):
    """mha_varlen_fwd's optional arguments (flash_api.cpp:586-649): K/V in pages addressed by block_table (lengths from
    cu_seqlens_k), and leftpad_k on packed K/V; both against the oracle on the logical sequences."""
    from oracle import attention_oracle as orc
    torch.manual_seed(8)
    H, hk, d, page, per = 8, 2, 128, 256, 3
    lens_q, lens_k = [50, 1, 700], [300, 1, 700]
    cu_q = torch.tensor([0] + list(np.cumsum(lens_q)), dtype=torch.int32, device="cuda")
    cu_k = torch.tensor([0] + list(np.cumsum(lens_k)), dtype=torch.int32, device="cuda")
    q = torch.randn(sum(lens_q), H, d, device="cuda", dtype=torch.bfloat16)
    kp = torch.randn(3 * per + 2, page, hk, d, device="cuda", dtype=torch.bfloat16)
    vp = torch.randn_like(kp)
    table = torch.randperm(3 * per + 2, device="cuda")[: 3 * per].reshape(3, per).to(torch.int32)
    out, lse, _, _ = kv.varlen_fwd(q, kp, vp, None, cu_q, cu_k, None, None, table, None, max(lens_q), max(lens_k), 0.0, d ** -0.5, False,
                                   causal, -1, -1, 0.0, False, None)
    k_log, v_log = kp[table.long()].reshape(3, per * page, hk, d), vp[table.long()].reshape(3, per * page, hk, d)
    for b in range(3):
        qs = slice(int(cu_q[b]), int(cu_q[b + 1]))
        o_ref, l_ref = orc.attention_fwd(q[qs][None], k_log[b:b + 1, :lens_k[b]], v_log[b:b + 1, :lens_k[b]], None, causal)
        assert max_abs(out[qs].float().cpu(), torch.from_numpy(o_ref[0])) < 2e-2
        assert max_abs(lse[:, qs].cpu(), torch.from_numpy(l_ref[0]).float()) < 2e-3
    # leftpad on packed keys: sequence b uses rows cu_k[b] + pad[b] .. cu_k[b+1] - 1
    k = torch.randn(sum(lens_k), hk, d, device="cuda", dtype=torch.bfloat16)
    v = torch.randn_like(k)
    pad = torch.tensor([17, 0, 130], dtype=torch.int32, device="cuda")
    out, lse, _, _ = kv.varlen_fwd(q, k, v, None, cu_q, cu_k, None, pad, None, None, max(lens_q), max(lens_k), 0.0, d ** -0.5, False,
                                   causal, -1, -1, 0.0, False, None)
    for b in range(3):
        qs = slice(int(cu_q[b]), int(cu_q[b + 1]))
        ks = slice(int(cu_k[b]) + int(pad[b]), int(cu_k[b + 1]))
        o_ref, l_ref = orc.attention_fwd(q[qs][None], k[ks][None], v[ks][None], None, causal)
        assert max_abs(out[qs].float().cpu(), torch.from_numpy(o_ref[0])) < 2e-2
        fin = torch.from_numpy(np.isfinite(l_ref[0]))   # causal with fewer keys than queries: the first rows see no key (lse = +inf)
        assert torch.equal(torch.isposinf(lse[:, qs].cpu()), ~fin)
        assert max_abs(lse[:, qs].cpu()[fin], torch.from_numpy(l_ref[0]).float()[fin]) < 2e-3


@pytest.mark.parametrize("feature", ["plain", "local", "alibi", "softcap"])
@pytest.mark.parametrize("paged,num_splits", [(False, 0), (False, 1), (True, 5)])
@pytest.mark.parametrize("sq,h,hk", [(1, 8, 2), (2, 16, 2), (5, 8, 1), (16, 8, 1), (7, 12, 4), (32, 8, 2)])
def test_kvcache_packed_query_heads(kv, knobs, sq, h, hk, paged, num_splits, feature):
    """Short query chunks with grouped heads (speculative decoding / chunked prefill): the g = H / Hk query heads of a KV
    group are packed into the rows of one block when g * Sq <= 128 (fa_api.cpp pack_group; FA3's PackGQA, the
    generalisation of the reference's single-row swap flash_api.cpp:429-437).  Checked against the fp64 oracle and, bit for
    bit, against the same call with packing switched off (same kernel, same arithmetic per row)."""
    from flash_attn_amd import backend as be
    from oracle import attention_oracle as orc
    torch.manual_seed(11)
    B, d, cap = 3, 128, 1024
    g = h // hk
    causal = True
    window = (100, 0) if feature == "local" else (-1, -1)
    softcap = 20.0 if feature == "softcap" else 0.0
    slopes = (torch.rand(h, device="cuda") * 0.3) if feature == "alibi" else None
    q = torch.randn(B, sq, h, d, device="cuda", dtype=torch.bfloat16)
    kn = torch.randn(B, sq, hk, d, device="cuda", dtype=torch.bfloat16)
    vn = torch.randn_like(kn)
    lens = torch.tensor([cap - sq, 300, 0], dtype=torch.int32, device="cuda")
    if paged:
        page, per = 256, cap // 256
        kc = torch.randn(B * per + 2, page, hk, d, device="cuda", dtype=torch.bfloat16)
        vc = torch.randn_like(kc)
        table = torch.randperm(B * per + 2, device="cuda")[: B * per].reshape(B, per).to(torch.int32)
    else:
        kc = torch.randn(B, cap, hk, d, device="cuda", dtype=torch.bfloat16)
        vc = torch.randn_like(kc)
        table = None

    def run():
        kcc, vcc = kc.clone(), vc.clone()
        o, l = kv.fwd_kvcache(q, kcc, vcc, kn, vn, lens, None, None, None, None, table, slopes, None, d ** -0.5, causal, window[0], window[1],
                              softcap, True, num_splits)
        return o, l, kcc, vcc

    out, lse, kc1, vc1 = run()
    # the binder's own single-row swap (no window, no ALiBi) hands the library a (g, Hk) problem; everything else packs inside
    swapped = sq == 1 and feature in ("plain", "softcap")
    assert be.last_schedule()["fwd_pack"] == (1 if swapped else g), be.last_schedule()
    knobs.set("FA_PACK_GQA", 0)
    out0, lse0, _, _ = run()
    assert be.last_schedule()["fwd_pack"] == 1
    knobs.unset("FA_PACK_GQA")
    if num_splits == 1:   # unsplit: identical per-row arithmetic (a heuristic split count may differ between the two grids)
        assert torch.equal(out, out0) and torch.equal(lse, lse0)
    k_log = kc1[table.long()].reshape(B, cap, hk, d) if paged else kc1
    v_log = vc1[table.long()].reshape(B, cap, hk, d) if paged else vc1
    f = lambda t: t.float().cpu().numpy()
    a = None if slopes is None else slopes.double().cpu().numpy()
    for b in range(B):
        L = int(lens[b]) + sq
        o_ref, l_ref = orc.attention_fwd(f(q[b:b + 1]), f(k_log[b:b + 1, :L]), f(v_log[b:b + 1, :L]), None, causal, window, softcap, a)
        assert max_abs(out[b:b + 1].float().cpu(), torch.from_numpy(o_ref).float()) < 2e-2
        fin = np.isfinite(l_ref)
        assert max_abs(lse[b:b + 1].cpu()[torch.from_numpy(fin)], torch.from_numpy(l_ref[fin]).float()) < 2e-3
        assert max_abs(out0[b:b + 1].float().cpu(), torch.from_numpy(o_ref).float()) < 2e-2


def _paged_pool(num_pages, page, hk, d, dtype, poison):
    kp = torch.randn(num_pages, page, hk, d, device="cuda", dtype=dtype)
    vp = torch.randn_like(kp)
    return kp, vp


@pytest.mark.parametrize("d,dtype", [(128, torch.bfloat16), (64, torch.bfloat16), (128, torch.float16)])
@pytest.mark.parametrize("page", [256, 512])
@pytest.mark.parametrize("mask", [(False, -1, -1), (True, -1, -1), (True, 300, 0)], ids=["full", "causal", "local_causal"])
def test_paged_prefill_on_the_w64_kernel(knobs, d, dtype, page, mask):
    """Round 5: plain attention over a PAGED cache on the 64-rows-per-wave forward (fa_fwd_w64_kernel<.., paged>: a buffer descriptor per 64-key tile, the table entry
    of the tile after next requested an iteration ahead; reference: the block_table path of mha_varlen_fwd / mha_fwd_kvcache, flash_api.cpp:586-649, :1243-1531).
    Packed queries against a shuffled page table, lengths that end inside a tile and inside a page, grouped heads; the rows of the last page behind a sequence's
    end hold NaN (a cache page holds whatever was there: the kernel must not let it into a product -- the tile's descriptor ends at the last key).  Against the
    fp64 oracle on the logical sequences, within 2x the error of the lock-step kernel that served paged caches before."""
    from flash_attn_amd import backend as be
    from oracle import attention_oracle as orc
    causal, wl, wr = mask
    torch.manual_seed(page + d)
    H, hk = 8, 2
    lens_q = [700, 1300, 64, 2048, 1]
    lens_k = [700, 1300 + 77, 1024, 2048 + 300, 513]
    per = (max(lens_k) + page - 1) // page
    B = len(lens_q)
    kp, vp = _paged_pool(B * per + 3, page, hk, d, dtype, True)
    table = torch.randperm(B * per + 3, device="cuda")[: B * per].reshape(B, per).to(torch.int32)
    for b in range(B):   # poison what lies behind the sequence's end in its last page
        pg, r = lens_k[b] // page, lens_k[b] % page
        if r:
            kp[int(table[b, pg]), r:] = float("nan"); vp[int(table[b, pg]), r:] = float("nan")
    cu_q = torch.tensor([0] + list(np.cumsum(lens_q)), dtype=torch.int32, device="cuda")
    cu_k = torch.tensor([0] + list(np.cumsum(lens_k)), dtype=torch.int32, device="cuda")
    q = torch.randn(sum(lens_q), H, d, device="cuda", dtype=dtype)
    run = lambda: be.varlen_fwd(q, kp, vp, None, cu_q, cu_k, None, None, table, None, max(lens_q), max(lens_k), 0.0, d ** -0.5, False, causal, wl, wr, 0.0, False, None)[:2]
    knobs.set("FA_FWD_NW", "64")
    out, lse = run()
    s = be.last_schedule()
    assert s["fwd_kernel"] == 3 and "paged" in s["name"], s
    out_b, lse_b = run()
    assert torch.equal(out, out_b) and torch.equal(lse, lse_b), "run-to-run"
    knobs.set("FA_FWD_NW", "8")
    out8, lse8 = run()
    assert be.last_schedule()["fwd_kernel"] == 1
    knobs.unset("FA_FWD_NW")
    assert torch.isfinite(out.float()).all()
    k_log, v_log = kp[table.long()].reshape(B, per * page, hk, d), vp[table.long()].reshape(B, per * page, hk, d)
    tol_o, tol_l = (1.2e-2, 8e-3) if dtype == torch.bfloat16 else (4e-3, 2e-3)
    for b in range(B):
        qs = slice(int(cu_q[b]), int(cu_q[b + 1]))
        o_ref, l_ref = orc.attention_fwd(q[qs][None], k_log[b:b + 1, :lens_k[b]], v_log[b:b + 1, :lens_k[b]], None, causal, (wl, wr))
        o_ref, l_ref = torch.from_numpy(o_ref[0]).float(), torch.from_numpy(l_ref[0]).float()
        e64, e8 = max_abs(out[qs].float().cpu(), o_ref), max_abs(out8[qs].float().cpu(), o_ref)
        assert e64 < max(2 * e8, tol_o), (b, e64, e8)
        fin = torch.isfinite(l_ref)
        assert torch.equal(torch.isposinf(lse[:, qs].cpu()), ~fin)
        assert max_abs(lse[:, qs].cpu()[fin], l_ref[fin]) < tol_l, b


def test_paged_kvcache_chunked_prefill_and_a_pool_above_4_gib(knobs):
    """fwd_kvcache with a paged cache and a long query chunk takes the same kernel (append first, then attention over cache + chunk); and a pool whose pages lie
    more than 4 GiB apart (the per-tile descriptor carries a 64-bit base: no 32-bit offset ever spans more than a tile)."""
    from flash_attn_amd import backend as be
    from oracle import attention_oracle as orc
    torch.manual_seed(4)
    B, H, hk, d, page, sq = 2, 8, 2, 128, 256, 1024
    n_pages = 9000     # x 256 rows x 2 heads x 128 x 2 B = 1.2 GB per tensor; the > 4 GiB case follows below
    kp = torch.randn(n_pages, page, hk, d, device="cuda", dtype=torch.bfloat16)
    vp = torch.randn_like(kp)
    lens = torch.tensor([1500, 333], dtype=torch.int32, device="cuda")
    per = (1500 + sq + page - 1) // page
    table = torch.tensor([[8999 - 7 * i for i in range(per)], [5 + 11 * i for i in range(per)]], dtype=torch.int32, device="cuda")
    q = torch.randn(B, sq, H, d, device="cuda", dtype=torch.bfloat16)
    kn = torch.randn(B, sq, hk, d, device="cuda", dtype=torch.bfloat16)
    vn = torch.randn_like(kn)
    knobs.set("FA_FWD_NW", "64")
    out, lse = be.fwd_kvcache(q, kp, vp, kn, vn, lens, None, None, None, None, table, None, None, d ** -0.5, True, -1, -1, 0.0, True, 1)
    s = be.last_schedule()
    assert s["fwd_kernel"] == 3 and "paged" in s["name"], s
    knobs.unset("FA_FWD_NW")
    for b in range(B):
        L = int(lens[b]) + sq
        kl = kp[table[b].long()].reshape(per * page, hk, d)[:L]
        vl = vp[table[b].long()].reshape(per * page, hk, d)[:L]
        assert torch.equal(kl[L - sq:], kn[b])
        o_ref, l_ref = orc.attention_fwd(q[b:b + 1], kl[None], vl[None], None, True)
        assert max_abs(out[b].float().cpu(), torch.from_numpy(o_ref[0]).float()) < 1.2e-2
        assert max_abs(lse[b].cpu(), torch.from_numpy(l_ref[0]).float()) < 8e-3
    # pages more than 4 GiB apart: 40000 pages x 256 rows x 1 head x 128 x 2 B = 2.6 GB per 10000 pages -> page 39999 sits 10.5 GB behind page 0
    hk1, n_big = 1, 40000
    kp = torch.zeros(n_big, page, hk1, d, device="cuda", dtype=torch.bfloat16)
    vp = torch.zeros_like(kp)
    ids = torch.tensor([39999, 3, 20001, 17000], dtype=torch.int64, device="cuda")
    kp[ids] = torch.randn(4, page, hk1, d, device="cuda", dtype=torch.bfloat16)
    vp[ids] = torch.randn(4, page, hk1, d, device="cuda", dtype=torch.bfloat16)
    table = ids.to(torch.int32)[None].contiguous()
    L = 4 * page - 100
    q = torch.randn(1, 600, 4, d, device="cuda", dtype=torch.bfloat16)
    knobs.set("FA_FWD_NW", "64")
    out, lse = be.fwd_kvcache(q, kp, vp, None, None, torch.tensor([L], dtype=torch.int32, device="cuda"), None, None, None, None, table, None, None, d ** -0.5, True,
                              -1, -1, 0.0, True, 1)
    s = be.last_schedule()
    knobs.unset("FA_FWD_NW")
    assert s["fwd_kernel"] == 3 and "paged" in s["name"], s
    kl, vl = kp[ids].reshape(4 * page, hk1, d)[:L], vp[ids].reshape(4 * page, hk1, d)[:L]
    o_ref, l_ref = orc.attention_fwd(q, kl[None], vl[None], None, True)
    assert max_abs(out.float().cpu(), torch.from_numpy(o_ref).float()) < 1.2e-2 and max_abs(lse.cpu(), torch.from_numpy(l_ref).float()) < 8e-3
