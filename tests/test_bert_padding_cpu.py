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


def test_unpad_for_concatenated_sequences_matches_the_documented_example_and_the_reference():
    """The example of the reference's docstring (bert_padding.py:136-148): rows [2,3 | 3,2 | 6] of a (3, 6) batch -> five sequences;
    and, when the reference tree is readable, equality with its function on a random case."""
    from flash_attn_amd.bert_padding import unpad_input_for_concatenated_sequences as ours
    mil = torch.tensor([[2, 3, 0, 0, 0, 0], [3, 2, 0, 0, 0, 0], [6, 0, 0, 0, 0, 0]])
    x = torch.arange(18.0).reshape(3, 6, 1)
    xp, idx, cu, mx = ours(x, mil)
    assert cu.dtype == torch.int32 and cu.tolist() == [0, 2, 5, 8, 10, 16] and mx == 6
    assert idx.tolist() == [0, 1, 2, 3, 4, 6, 7, 8, 9, 10, 12, 13, 14, 15, 16, 17]
    assert torch.equal(xp.flatten(), x.flatten()[idx])
    import importlib.util, os
    ref_path = "/root/reference/flash_attn/bert_padding.py"
    if os.path.exists(ref_path):
        spec = importlib.util.spec_from_file_location("ref_bert_padding", ref_path)
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
        torch.manual_seed(1)
        mil = torch.tensor([[4, 1, 2, 0, 0, 0, 0, 0, 0], [9, 0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0, 0], [1, 1, 1, 1, 0, 0, 0, 0, 0]])
        x = torch.randn(4, 9, 2, 8)
        a, b = ours(x, mil), ref.unpad_input_for_concatenated_sequences(x, mil)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and a[3] == b[3]


def test_padded_batch_args_left_and_right_padding():
    from flash_attn_amd.bert_padding import padded_batch_args, padded_batch_is_contiguous, unpad_input
    m = torch.tensor([[1, 1, 1, 0, 0], [0, 0, 1, 1, 1], [0, 0, 0, 0, 0], [1, 1, 1, 1, 1]], dtype=torch.bool)
    lens, starts = padded_batch_args(m)
    assert lens.tolist() == [3, 3, 0, 5] and starts.tolist() == [0, 2, 0, 0] and lens.dtype == torch.int32
    assert padded_batch_is_contiguous(m)
    assert not padded_batch_is_contiguous(torch.tensor([[1, 0, 1, 0]], dtype=torch.bool))
    # rows starts[b] .. starts[b] + lens[b] - 1 of the flattened batch are exactly unpad_input's indices
    S = m.shape[1]
    rows = torch.cat([torch.arange(int(s), int(s + n)) + b * S for b, (n, s) in enumerate(zip(lens, starts))])
    assert rows.tolist() == unpad_input(torch.zeros(4, 5, 1), m)[1].tolist()
