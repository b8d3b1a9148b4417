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
    assert lib.fa_abi_version() == _cabi.FA_ABI_VERSION == 4
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
    a.h_k, a.d = 1, 72
    assert lib.fa_fwd(ctypes.byref(a), None) == _cabi.FA_ERR_UNSUPPORTED
    a.d, a.dtype = 128, 7
    assert lib.fa_fwd(ctypes.byref(a), None) == _cabi.FA_ERR_INVALID_ARGUMENT
    a.dtype = _cabi.FA_DTYPE_FP16
    assert lib.fa_fwd(ctypes.byref(a), None) == _cabi.FA_ERR_INVALID_ARGUMENT  # NULL tensors
    with pytest.raises(RuntimeError):
        _cabi.check(_cabi.FA_ERR_INVALID_ARGUMENT)


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
