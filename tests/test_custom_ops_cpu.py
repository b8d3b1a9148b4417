"""CPU suite: the custom-op registration of the interface mirror (schemas, fake implementations, autograd tracing with
fake CUDA tensors) -- the part of torch.compile / export conformance that needs no GPU."""
import pytest
import torch
from torch._subclasses.fake_tensor import FakeTensorMode


def test_ops_are_registered_with_mutation_annotations():
    from flash_attn_amd import flash_attn_interface as fi  # noqa: F401
    ns = torch.ops.flash_attn_amd
    for name in ("_flash_attn_forward", "_flash_attn_varlen_forward", "_flash_attn_backward", "_flash_attn_varlen_backward"):
        assert hasattr(ns, name)
    schema = str(ns._flash_attn_backward.default._schema)
    assert "Tensor(a2!)? dq" in schema or "dq" in schema and "!" in schema   # dq / dk / dv are declared as mutated
    assert str(ns._flash_attn_forward.default._schema).count("Tensor") >= 8


def test_fake_implementations_give_reference_shapes():
    from flash_attn_amd import flash_attn_interface as fi
    with FakeTensorMode():
        q = torch.empty(2, 100, 4, 64, device="cuda", dtype=torch.bfloat16)
        k = torch.empty(2, 130, 2, 64, device="cuda", dtype=torch.bfloat16)
        out, lse, p, rng = fi._flash_attn_forward(q, k, k, 0.1, 0.125, True, -1, -1, 0.0, None, True)
        assert out.shape == q.shape and lse.shape == (2, 4, 100) and lse.dtype == torch.float32
        assert p.shape == (2, 4, 100, 130) and p.dtype == torch.uint8 and rng.shape == (2,) and rng.dtype == torch.int64
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(k)
        d = fi._flash_attn_backward(out, q, k, k, out, lse, dq, dk, dv, 0.1, 0.125, True, -1, -1, 0.0, None, False, rng)
        assert d.shape == (2, 4, 100)
        qv = torch.empty(230, 4, 64, device="cuda", dtype=torch.float16)
        kv = torch.empty(300, 2, 64, device="cuda", dtype=torch.float16)
        cu = torch.empty(4, device="cuda", dtype=torch.int32)
        out, lse, p, rng = fi._flash_attn_varlen_forward(qv, kv, kv, cu, cu, 100, 120, 0.0, 0.125, False)
        assert out.shape == qv.shape and lse.shape == (4, 230) and p.numel() == 0
        d = fi._flash_attn_varlen_backward(out, qv, kv, kv, out, lse, None, None, None, cu, cu, 100, 120, 0.0, 0.125, False, -1, -1,
                                           0.0, None, False)
        assert d.shape == (4, 230)


def test_public_function_traces_with_fake_tensors():
    from flash_attn_amd import flash_attn_interface as fi
    with FakeTensorMode():
        q = torch.empty(2, 64, 4, 80, device="cuda", dtype=torch.bfloat16)   # 80 -> padded to 128 inside
        k = torch.empty(2, 96, 2, 80, device="cuda", dtype=torch.bfloat16)
        out, lse, _ = fi.flash_attn_func(q, k, k, causal=True, return_attn_probs=True)
        assert out.shape == q.shape and lse.shape == (2, 4, 64)
