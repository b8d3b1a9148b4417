"""CPU: packed-batch helpers (producer of cu_seqlens) round-trip and match the reference's semantics."""
import torch

from flash_attn_amd.bert_padding import pad_input, unpad_input


def test_unpad_pad_round_trip_and_cu_seqlens():
    torch.manual_seed(0)
    B, S, H, D = 4, 9, 2, 8
    lens = torch.tensor([9, 0, 4, 7])
    mask = torch.arange(S)[None, :] < lens[:, None]
    x = torch.randn(B, S, H, D, requires_grad=True)
    xp, idx, cu, mx, used = unpad_input(x, mask)
    assert cu.dtype == torch.int32 and cu.tolist() == [0, 9, 9, 13, 20] and mx == 9 and used.tolist() == [9, 0, 4, 7]
    assert xp.shape == (20, H, D) and torch.equal(xp[9:13], x[2, :4])
    back = pad_input(xp, idx, B, S)
    assert torch.equal(back[mask], x[mask]) and torch.all(back[~mask] == 0)
    (g,) = torch.autograd.grad(back.sum(), x)
    assert torch.equal(g, mask[:, :, None, None].expand_as(x).to(g.dtype))


def test_unused_mask_keeps_allocated_slots():
    mask = torch.tensor([[1, 1, 0, 0], [1, 0, 0, 0]])
    unused = torch.tensor([[0, 0, 1, 0], [0, 1, 1, 0]])
    x = torch.arange(8.0).reshape(2, 4, 1)
    xp, idx, cu, mx, used = unpad_input(x, mask, unused)
    assert cu.tolist() == [0, 3, 6] and used.tolist() == [2, 1] and mx == 3
    assert xp.flatten().tolist() == [0.0, 1.0, 2.0, 4.0, 5.0, 6.0]
