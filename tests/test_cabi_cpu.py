"""CPU suite: the C-ABI library loads and exports every symbol include/fa_gfx950.h declares; the
ctypes mirror matches the C structs; the torch extension exposes the reference backend-module API;
argument validation raises before any GPU work.  (No compute calls without a GPU.)"""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "flash-attention_amd")


@pytest.fixture(scope="module")
def built():
    lib = os.path.join(PKG, "libfa_gfx950.so")
    if not os.path.exists(lib):
        subprocess.check_call([sys.executable, os.path.join(PKG, "build.py")])
    return lib


def _declared_functions():
    hdr = open(os.path.join(ROOT, "include", "fa_gfx950.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(fa_[a-z_0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol(built):
    from flash_attn_amd import _cabi
    names = _declared_functions()
    assert set(names) == set(_cabi.EXPORTS), (names, _cabi.EXPORTS)
    lib = ctypes.CDLL(built)
    for n in names:
        assert hasattr(lib, n), n


def test_ctypes_mirror_matches_c_structs(built):
    from flash_attn_amd import _cabi
    lib = _cabi.load()
    assert lib.fa_abi_version() == _cabi.FA_ABI_VERSION == 6
    assert lib.fa_sizeof_kvappend_params() == ctypes.sizeof(_cabi.FaKvAppendParams)
    assert lib.fa_sizeof_fwd_params() == ctypes.sizeof(_cabi.FaFwdParams)
    assert lib.fa_sizeof_bwd_params() == ctypes.sizeof(_cabi.FaBwdParams)


def test_cabi_rejects_bad_arguments_without_touching_the_gpu(built):
    from flash_attn_amd import _cabi
    lib = _cabi.load()
    a = _cabi.FaFwdParams()
    a.b, a.h, a.h_k, a.d, a.dtype = 1, 3, 2, 128, _cabi.FA_DTYPE_BF16
    assert lib.fa_fwd(ctypes.byref(a), None) == _cabi.FA_ERR_INVALID_ARGUMENT
    assert b"heads" in lib.fa_last_error()
    a.h_k, a.d = 1, 72   # the forward takes any multiple of 8 (run-time column bound): rejected later, for its NULL tensors
    assert lib.fa_fwd(ctypes.byref(a), None) == _cabi.FA_ERR_INVALID_ARGUMENT and b"non-NULL" in lib.fa_last_error()
    a.d = 60
    assert lib.fa_fwd(ctypes.byref(a), None) == _cabi.FA_ERR_INVALID_ARGUMENT and b"multiple of 8" in lib.fa_last_error()
    g = _cabi.FaBwdParams()   # the backward takes any multiple of 8 too (run-time column bound of the next built size's kernels)
    g.b, g.h, g.h_k, g.d, g.dtype = 1, 2, 1, 72, _cabi.FA_DTYPE_BF16
    assert lib.fa_bwd(ctypes.byref(g), None) == _cabi.FA_ERR_INVALID_ARGUMENT and b"non-NULL" in lib.fa_last_error()
    g.d = 260
    assert lib.fa_bwd(ctypes.byref(g), None) == _cabi.FA_ERR_INVALID_ARGUMENT and b"multiple of 8" in lib.fa_last_error()
    a.d, a.dtype = 128, 7
    assert lib.fa_fwd(ctypes.byref(a), None) == _cabi.FA_ERR_INVALID_ARGUMENT
    a.dtype = _cabi.FA_DTYPE_FP16
    assert lib.fa_fwd(ctypes.byref(a), None) == _cabi.FA_ERR_INVALID_ARGUMENT  # NULL tensors
    with pytest.raises(RuntimeError):
        _cabi.check(_cabi.FA_ERR_INVALID_ARGUMENT)


def test_cabi_contract_of_the_later_entry_points(built):
    """Dropout, KV-cache, rotary and workspace entry points: argument contract checked on the host, no launch."""
    from flash_attn_amd import _cabi
    lib = _cabi.load()
    assert lib.fa_sizeof_rotary_params() == ctypes.sizeof(_cabi.FaRotaryParams)
    dummy = ctypes.c_void_p(0x1000)  # never dereferenced: every call below is rejected before a launch

    def fwd_params():
        a = _cabi.FaFwdParams()
        a.b, a.h, a.h_k, a.d, a.dtype = 2, 4, 2, 128, _cabi.FA_DTYPE_BF16
        a.q = a.k = a.v = a.o = a.softmax_lse = dummy
        a.seqlen_q, a.seqlen_k, a.total_q = 16, 512, 32
        return a

    a = fwd_params()
    a.p_dropout = 0.1                                   # dropout needs the device rng_state
    assert lib.fa_fwd(ctypes.byref(a), None) == _cabi.FA_ERR_INVALID_ARGUMENT and b"rng_state" in lib.fa_last_error()
    a = fwd_params()
    a.p_dropout = 1.0
    assert lib.fa_fwd(ctypes.byref(a), None) == _cabi.FA_ERR_INVALID_ARGUMENT and b"p_dropout" in lib.fa_last_error()
    a = fwd_params()
    a.randval = dummy                                   # return_softmax payload only with dropout
    assert lib.fa_fwd(ctypes.byref(a), None) == _cabi.FA_ERR_INVALID_ARGUMENT and b"return_softmax" in lib.fa_last_error()
    a = fwd_params()
    a.num_splits = 4                                    # key splits belong to the KV-cache entry point
    assert lib.fa_fwd(ctypes.byref(a), None) == _cabi.FA_ERR_INVALID_ARGUMENT and b"num_splits" in lib.fa_last_error()
    a = fwd_params()
    a.block_table = dummy                               # so do paged caches
    assert lib.fa_fwd(ctypes.byref(a), None) == _cabi.FA_ERR_INVALID_ARGUMENT and b"fa_fwd_kvcache" in lib.fa_last_error()
    a = fwd_params()
    a.block_table, a.page_block_size = dummy, 100       # page size must be a multiple of 256 (flash_api.cpp:1318)
    assert lib.fa_fwd_kvcache(ctypes.byref(a), None) == _cabi.FA_ERR_INVALID_ARGUMENT and b"256" in lib.fa_last_error()
    a = fwd_params()
    a.block_table, a.page_block_size, a.leftpad_k, a.seqused_k = dummy, 256, dummy, dummy
    assert lib.fa_fwd_kvcache(ctypes.byref(a), None) == _cabi.FA_ERR_INVALID_ARGUMENT and b"leftpad_k" in lib.fa_last_error()
    a = fwd_params()
    a.p_dropout, a.rng_state = 0.1, dummy               # inference path: no dropout
    assert lib.fa_fwd_kvcache(ctypes.byref(a), None) == _cabi.FA_ERR_INVALID_ARGUMENT and b"inference" in lib.fa_last_error()
    # workspace sizing is pure host arithmetic: decode with idle CUs wants split-KV scratch, a full grid does not
    a = fwd_params()
    a.b, a.seqlen_q, a.seqlen_k, a.total_q = 1, 1, 32768, 1
    need = lib.fa_fwd_workspace_bytes(ctypes.byref(a))
    assert need > 0 and need % ((a.d + 1) * 4 * a.b * a.h * a.seqlen_q) == 0
    a.b = 512
    assert lib.fa_fwd_workspace_bytes(ctypes.byref(a)) == 0
    a.b, a.num_splits = 1, 8
    assert lib.fa_fwd_workspace_bytes(ctypes.byref(a)) == 8 * (a.d + 1) * 4 * a.h
    # varlen: a work list is requested only when the max_seqlen grid would be mostly empty
    a = fwd_params()
    a.cu_seqlens_q = a.cu_seqlens_k = dummy
    a.b, a.seqlen_q, a.seqlen_k, a.total_q = 160, 16384, 16384, 65536
    assert lib.fa_fwd_workspace_bytes(ctypes.byref(a)) > 0
    a.b, a.seqlen_q, a.seqlen_k = 16, 4096, 4096
    assert lib.fa_fwd_workspace_bytes(ctypes.byref(a)) == 0
    r = _cabi.FaRotaryParams()
    r.x = r.y = r.cos = r.sin = dummy
    r.b, r.s, r.h, r.d, r.rotary_dim, r.seqlen_ro, r.dtype = 1, 1, 1, 128, 24, 8, _cabi.FA_DTYPE_FP16
    assert lib.fa_rotary(ctypes.byref(r), None) == _cabi.FA_ERR_INVALID_ARGUMENT and b"divisible by 16" in lib.fa_last_error()
    r.rotary_dim = 256
    assert lib.fa_rotary(ctypes.byref(r), None) == _cabi.FA_ERR_INVALID_ARGUMENT and b"headdim" in lib.fa_last_error()
    bw = _cabi.FaBwdParams()
    bw.b, bw.h, bw.h_k, bw.d, bw.dtype = 1, 2, 2, 64, _cabi.FA_DTYPE_BF16
    for f in ("dout", "q", "k", "v", "o", "softmax_lse", "dq", "dk", "dv", "softmax_d"):
        setattr(bw, f, dummy)
    bw.p_dropout = 0.2
    assert lib.fa_bwd(ctypes.byref(bw), None) == _cabi.FA_ERR_INVALID_ARGUMENT and b"rng_state" in lib.fa_last_error()
    assert lib.fa_set_rng_state(1, 2, None, None) == _cabi.FA_ERR_INVALID_ARGUMENT


def test_torch_extension_is_the_reference_backend_module(built):
    import flash_attn_2_cuda as m
    for fn in ("fwd", "varlen_fwd", "bwd", "varlen_bwd", "fwd_kvcache"):
        assert callable(getattr(m, fn))
    q = torch.zeros(1, 4, 2, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="CUDA"):
        m.fwd(q, q, q, None, None, 0.0, 0.125, False, -1, -1, 0.0, False, None)
    with pytest.raises(RuntimeError, match="generator"):
        m.fwd(q, q, q, None, None, 0.0, 0.125, False, -1, -1, 0.0, False, torch.Generator())
    with pytest.raises(RuntimeError, match="CUDA"):
        m.fwd_kvcache(q, q, q, None, None, None, None, None, None, None, None, None, None, 0.125, False, -1, -1, 0.0, True, 0)
    assert os.path.dirname(m.__file__) == PKG  # in-tree build, not site-packages


def test_ctypes_backend_validation_matches():
    from flash_attn_amd import backend as be
    q = torch.zeros(1, 4, 2, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="CUDA"):
        be.fwd(q, q, q, None, None, 0.0, 0.125, False, -1, -1, 0.0, False, None)
    with pytest.raises(RuntimeError, match="generator"):
        be.fwd(q, q, q, None, None, 0.0, 0.125, False, -1, -1, 0.0, False, torch.Generator())
    with pytest.raises(RuntimeError, match="p_dropout"):
        be.fwd(q, q, q, None, None, 1.0, 0.125, False, -1, -1, 0.0, False, None)
    with pytest.raises(RuntimeError, match="CUDA"):  # dropout is built: the device check comes first
        be.fwd(q, q, q, None, None, 0.1, 0.125, False, -1, -1, 0.0, False, None)


def test_interface_mirror_exports_reference_names():
    import flash_attn_amd
    from flash_attn_amd import flash_attn_interface as fi
    for n in ("flash_attn_func", "flash_attn_varlen_func", "flash_attn_qkvpacked_func", "flash_attn_kvpacked_func",
              "flash_attn_varlen_qkvpacked_func", "flash_attn_varlen_kvpacked_func", "flash_attn_with_kvcache",
              "_flash_attn_forward", "_flash_attn_backward", "_flash_attn_varlen_forward", "_flash_attn_varlen_backward"):
        assert callable(getattr(fi, n))
    assert fi.flash_attn_gpu.__name__ in ("flash_attn_2_cuda", "flash_attn_amd.backend")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from flash_attn_amd import _cabi
    monkeypatch.setattr(_cabi, "_LIB", None)
    monkeypatch.setenv("FA_GFX950_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU or PyTorch fallback"):
        _cabi.load()
