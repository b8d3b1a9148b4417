"""flash_attn_padded_func (fused unpad -> attention -> pad, C ABI v6 seqused_q / seqused_k) against the reference's three-pass chain
unpad_input -> flash_attn_varlen_func -> pad_input (flash_attn/bert_padding.py:98-128, 204-218), forward and backward."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mask(B, S, mode, gen):
    lens = torch.randint(1, S + 1, (B,), generator=gen)
    lens[0] = S
    if B > 2:
        lens[2] = 0          # an empty entry
    pos = torch.arange(S)[None, :]
    if mode == "right":
        return pos < lens[:, None]
    return pos >= (S - lens)[:, None]


@pytest.mark.parametrize("mode", ["right", "left"])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("B,S,H,Hk,D", [(4, 384, 4, 2, 128), (3, 1100, 2, 2, 64), (5, 200, 4, 4, 96), (2, 300, 2, 2, 100), (3, 260, 4, 1, 72)])
def test_padded_matches_unpad_varlen_pad(mode, causal, B, S, H, Hk, D):
    from flash_attn_amd import flash_attn_padded_func, flash_attn_varlen_func
    from flash_attn_amd.bert_padding import pad_input, padded_batch_args, unpad_input
    gen = torch.Generator().manual_seed(B * S + D)
    mask = _mask(B, S, mode, gen).cuda()
    q = torch.randn(B, S, H, D, generator=gen).cuda().to(torch.bfloat16).requires_grad_()
    k = torch.randn(B, S, Hk, D, generator=gen).cuda().to(torch.bfloat16).requires_grad_()
    v = torch.randn(B, S, Hk, D, generator=gen).cuda().to(torch.bfloat16).requires_grad_()
    do = torch.randn(B, S, H, D, generator=gen).cuda().to(torch.bfloat16)

    # the reference's chain: three gather / scatter passes around the varlen kernels
    qu, idx, cu, mx, _ = unpad_input(q, mask)
    ku, _, _, _, _ = unpad_input(k, mask)
    vu, _, _, _, _ = unpad_input(v, mask)
    ref = pad_input(flash_attn_varlen_func(qu, ku, vu, cu, cu, mx, mx, causal=causal), idx, B, S)
    gq, gk, gv = torch.autograd.grad(ref, (q, k, v), do)

    lens, starts = padded_batch_args(mask)
    out = flash_attn_padded_func(q, k, v, lens, starts_q=starts, causal=causal)
    dq, dk, dv = torch.autograd.grad(out, (q, k, v), do)

    valid = mask[:, :, None, None]
    assert out.shape == ref.shape and torch.isfinite(out.float()).all()
    for got in (out, dq, dk, dv):   # padded rows: exact zeros, as pad_input leaves them
        assert float(got.float().masked_fill(valid, 0).abs().max()) == 0.0
    # same kernels on the same sequences (the schedule may differ with the maximum length the dispatch sees): bf16 round-off only
    assert float((out.float() - ref.float()).abs().max()) < 1.6e-2
    for got, want in ((dq, gq), (dk, gk), (dv, gv)):
        assert float((got.float() - want.float()).abs().max()) < 6e-2, (mode, causal)


def test_padded_cross_attention_and_memory():
    """Different padded lengths / masks for q and k; no unpadded copies of q, k, v are made (peak memory stays below the chain's)."""
    from flash_attn_amd import flash_attn_padded_func, flash_attn_varlen_func
    from flash_attn_amd.bert_padding import pad_input, padded_batch_args, unpad_input
    gen = torch.Generator().manual_seed(5)
    B, Sq, Sk, H, D = 6, 512, 1024, 8, 128
    mq, mk = _mask(B, Sq, "right", gen).cuda(), _mask(B, Sk, "left", gen).cuda()
    mk[2] = mk[1]   # (entry 2 of q is empty; give its keys something)
    q = torch.randn(B, Sq, H, D, generator=gen).cuda().to(torch.bfloat16)
    k = torch.randn(B, Sk, H, D, generator=gen).cuda().to(torch.bfloat16)
    v = torch.randn(B, Sk, H, D, generator=gen).cuda().to(torch.bfloat16)
    lq, sq_ = padded_batch_args(mq)
    lk, sk_ = padded_batch_args(mk)
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats(); base = torch.cuda.memory_allocated()
    out = flash_attn_padded_func(q, k, v, lq, lk, sq_, sk_)
    torch.cuda.synchronize(); peak_fused = torch.cuda.max_memory_allocated() - base
    del out
    torch.cuda.reset_peak_memory_stats(); base = torch.cuda.memory_allocated()
    qu, iq, cq, xq, _ = unpad_input(q, mq)
    ku, _, ck, xk, _ = unpad_input(k, mk)
    vu, _, _, _, _ = unpad_input(v, mk)
    ref = pad_input(flash_attn_varlen_func(qu, ku, vu, cq, ck, xq, xk), iq, B, Sq)
    torch.cuda.synchronize(); peak_chain = torch.cuda.max_memory_allocated() - base
    out = flash_attn_padded_func(q, k, v, lq, lk, sq_, sk_)
    assert float((out.float() - ref.float()).abs().max()) < 1.6e-2
    assert float(out.float().masked_fill(mq[:, :, None, None], 0).abs().max()) == 0.0
    assert peak_fused < peak_chain, (peak_fused, peak_chain)
