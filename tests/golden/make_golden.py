"""Generate golden vectors from the REFERENCE's own oracle (run in the build container only).

    python tests/golden/make_golden.py

Imports ``attention_ref`` from /root/reference/tests/test_util.py (the function the
reference's acceptance tests compare against, tests/test_flash_attn.py:217-304) and
torch autograd for the gradients, on CPU in fp32 (``upcast=True``), with seeded inputs.
Inputs are quantised to bf16-representable values first so that the same tensors can
be fed to the HIP kernels bit-for-bit.  Also records the fixed-``cu_seqlens`` varlen
known-answer layouts the reference tests use (tests/test_flash_attn.py:2363-2380,
tests/test_flash_attn_ck.py:1515-1560).  /root/reference is not available on the GPU
box, hence committed fixtures.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _import_ref():
    # tests/test_util.py imports flash_attn.bert_padding -> flash_attn/__init__ -> backend module.
    # Provide an empty stand-in backend so the pure-python oracle imports on CPU.
    sys.modules.setdefault("flash_attn_2_cuda", types.ModuleType("flash_attn_2_cuda"))
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "tests"))
    import test_util  # noqa
    return test_util


CASES = [
    # name, B, Sq, Sk, H, Hk, D, causal, window, softcap, alibi
    ("mha_full_d64", 2, 96, 96, 2, 2, 64, False, (-1, -1), 0.0, False),
    ("mha_causal_d128", 1, 113, 171, 2, 2, 128, True, (-1, -1), 0.0, False),
    ("gqa_causal_sq_gt_sk", 1, 203, 113, 4, 2, 64, True, (-1, -1), 0.0, False),
    ("mqa_local_d128", 1, 128, 177, 2, 1, 128, False, (37, 11), 0.0, False),
    ("gqa_causal_window_d128", 1, 160, 160, 4, 2, 128, True, (64, 0), 0.0, False),
    # one-sided left window is written (20, Sk): the reference *test oracle* reads a literal -1 on the right as
    # "col > row + shift - 1" (tests/test_util.py:176-181) whereas the API/kernel treat it as unbounded
    # (flash_api.cpp:159-160); an explicit right bound >= Sk means the same thing to both.
    ("local_left_only_d64", 1, 99, 160, 2, 2, 64, False, (20, 160), 0.0, False),
    ("local_right_only_d64", 1, 160, 99, 2, 1, 64, False, (-1, 13), 0.0, False),
    ("tiny_sq1", 2, 1, 77, 2, 2, 128, True, (-1, -1), 0.0, False),
    ("softcap_d64", 1, 64, 96, 2, 2, 64, True, (-1, -1), 15.0, False),
    ("alibi_d64", 2, 80, 112, 2, 1, 64, True, (-1, -1), 0.0, True),
    ("d32_full", 1, 70, 70, 2, 2, 32, False, (-1, -1), 0.0, False),
    ("d96_causal", 1, 65, 129, 2, 1, 96, True, (-1, -1), 0.0, False),
    ("d256_causal", 1, 48, 80, 2, 2, 256, True, (-1, -1), 0.0, False),
]


def main():
    tu = _import_ref()
    out = {}
    for (name, B, Sq, Sk, H, Hk, D, causal, window, softcap, alibi) in CASES:
        g = torch.Generator().manual_seed(sum(map(ord, name)))
        q = torch.randn(B, Sq, H, D, generator=g).bfloat16().float().requires_grad_()
        k = torch.randn(B, Sk, Hk, D, generator=g).bfloat16().float().requires_grad_()
        v = torch.randn(B, Sk, Hk, D, generator=g).bfloat16().float().requires_grad_()
        do = torch.randn(B, Sq, H, D, generator=g).bfloat16().float()
        bias = None
        slopes = None
        if alibi:
            slopes = (torch.rand(B, H, generator=g) * 0.3).float()
            # bias definition of tests/test_flash_attn.py:29-58 (that module needs a GPU at import,
            # :23-26, so the three lines are restated): -slope * |i + Sk - Sq - j|
            ii = torch.arange(Sq)[:, None]
            jj = torch.arange(Sk)[None, :]
            bias = -slopes[:, :, None, None] * (ii + Sk - Sq - jj).abs().float()
        o, _ = tu.attention_ref(q, k, v, None, None, bias, 0.0, None, causal=causal,
                                window_size=window, softcap=softcap)
        try:
            dq, dk, dv = torch.autograd.grad(o, (q, k, v), do)
        except RuntimeError:
            # the reference's softcap branch applies tanh in place (tests/test_util.py:234-237),
            # which autograd cannot differentiate: forward-only golden for that case.
            dq = dk = dv = None
        pre = name + "/"
        # inputs are exactly bf16-representable: store the 16-bit patterns (upper half of the fp32 word)
        for nm, t in (("q", q), ("k", k), ("v", v), ("do", do)):
            bits = t.detach().numpy().astype(np.float32).view(np.uint32)
            assert not np.any(bits & 0xFFFF)
            out[pre + nm + "_bf16bits"] = (bits >> 16).astype(np.uint16)
        out[pre + "out"] = o.detach().numpy().astype(np.float32)
        if dq is not None:
            out[pre + "dq"] = dq.numpy().astype(np.float32)
            out[pre + "dk"] = dk.numpy().astype(np.float32)
            out[pre + "dv"] = dv.numpy().astype(np.float32)
        out[pre + "meta"] = np.array([B, Sq, Sk, H, Hk, D, int(causal), window[0], window[1]], dtype=np.int64)
        out[pre + "softcap"] = np.array([softcap], dtype=np.float64)
        if slopes is not None:
            out[pre + "alibi_slopes"] = slopes.numpy()
        print(name, "out", tuple(o.shape), "max|out|", float(o.detach().abs().max()))
    np.savez_compressed(os.path.join(HERE, "attention_ref_cases.npz"), **out)

    # dropout: the reference oracle with an explicit keep-mask (tests/test_util.py:262-269), seeded mask
    dro = {}
    for (name, B, Sq, Sk, H, Hk, D, causal, window, pdrop) in [
            ("drop_full_d64", 1, 72, 96, 2, 2, 64, False, (-1, -1), 0.17),
            ("drop_causal_gqa_d128", 2, 65, 100, 4, 2, 128, True, (-1, -1), 0.3),
            ("drop_local_d64", 1, 96, 96, 2, 1, 64, False, (30, 10), 0.1)]:
        g = torch.Generator().manual_seed(sum(map(ord, name)))
        q = torch.randn(B, Sq, H, D, generator=g).bfloat16().float().requires_grad_()
        k = torch.randn(B, Sk, Hk, D, generator=g).bfloat16().float().requires_grad_()
        v = torch.randn(B, Sk, Hk, D, generator=g).bfloat16().float().requires_grad_()
        do = torch.randn(B, Sq, H, D, generator=g).bfloat16().float()
        keep = torch.rand(B, H, Sq, Sk, generator=g) >= pdrop
        o, _ = tu.attention_ref(q, k, v, None, None, None, pdrop, keep, causal=causal, window_size=window)
        dq, dk, dv = torch.autograd.grad(o, (q, k, v), do)
        pre = name + "/"
        for nm, t in (("q", q), ("k", k), ("v", v), ("do", do)):
            bits = t.detach().numpy().astype(np.float32).view(np.uint32)
            dro[pre + nm + "_bf16bits"] = (bits >> 16).astype(np.uint16)
        dro[pre + "keep"] = np.packbits(keep.numpy())
        for nm, t in (("out", o), ("dq", dq), ("dk", dk), ("dv", dv)):
            dro[pre + nm] = t.detach().numpy().astype(np.float32)
        dro[pre + "meta"] = np.array([B, Sq, Sk, H, Hk, D, int(causal), window[0], window[1]], dtype=np.int64)
        dro[pre + "p"] = np.array([pdrop], dtype=np.float64)
        print(name, "out", tuple(o.shape))
    np.savez_compressed(os.path.join(HERE, "dropout_ref_cases.npz"), **dro)

    # documented causal mask pictures, flash_attn_interface.py:1176-1185 (1 = keep)
    pics = {
        "mask_2x5": np.array([[1, 1, 1, 1, 0], [1, 1, 1, 1, 1]], dtype=np.int8),
        "mask_5x2": np.array([[0, 0], [0, 0], [0, 0], [1, 0], [1, 1]], dtype=np.int8),
    }
    # cross-check the pictures against the reference's construct_local_mask (True = masked)
    for nm, pic in pics.items():
        sq, sk = pic.shape
        m = tu.construct_local_mask(sq, sk, (-1, 0))
        assert np.array_equal(~m.numpy(), pic.astype(bool)), nm
    # fixed cu_seqlens layouts used by reference regression tests
    pics["cu_bwd_varlen_overflow_q"] = np.array([0, 76, 110, 256], dtype=np.int32)   # test_flash_attn.py:2363
    pics["cu_bwd_varlen_overflow_k"] = np.array([0, 1, 2, 3], dtype=np.int32)
    pics["cu_seqq_zero_q"] = np.array([0, 0, 256, 512], dtype=np.int32)              # test_flash_attn_ck.py:1522-1560
    pics["cu_seqq_zero_k"] = np.array([0, 503, 768, 1536], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "known_answers.npz"), **pics)
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
