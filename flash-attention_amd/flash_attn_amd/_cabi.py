"""ctypes binding of the C ABI in include/fa_gfx950.h (libfa_gfx950.so).

The structs mirror ``FaFwdParams`` / ``FaBwdParams`` field for field; ``load()`` checks
``fa_sizeof_*_params()`` and ``fa_abi_version()`` against this file so a stale library cannot be
driven with a mismatched layout.  There is no fallback: if the library is missing, ``load()``
raises (the product path must fail loudly without the HIP code).
"""
from __future__ import annotations

import ctypes as C
import os

FA_ABI_VERSION = 6
FA_DTYPE_FP16, FA_DTYPE_BF16 = 0, 1
FA_OK, FA_ERR_INVALID_ARGUMENT, FA_ERR_UNSUPPORTED, FA_ERR_LAUNCH, FA_ERR_WORKSPACE = 0, -1, -2, -3, -4

_i64, _i32, _f32, _vp = C.c_int64, C.c_int32, C.c_float, C.c_void_p


class FaFwdParams(C.Structure):
    _fields_ = [
        ("q", _vp), ("k", _vp), ("v", _vp), ("o", _vp), ("softmax_lse", _vp),
        ("q_batch_stride", _i64), ("q_row_stride", _i64), ("q_head_stride", _i64),
        ("k_batch_stride", _i64), ("k_row_stride", _i64), ("k_head_stride", _i64),
        ("v_batch_stride", _i64), ("v_row_stride", _i64), ("v_head_stride", _i64),
        ("o_batch_stride", _i64), ("o_row_stride", _i64), ("o_head_stride", _i64),
        ("cu_seqlens_q", _vp), ("cu_seqlens_k", _vp), ("seqused_k", _vp),
        ("alibi_slopes", _vp), ("alibi_batch_stride", _i64),
        ("b", _i32), ("h", _i32), ("h_k", _i32), ("d", _i32),
        ("seqlen_q", _i32), ("seqlen_k", _i32), ("total_q", _i32), ("dtype", _i32),
        ("is_causal", _i32), ("window_left", _i32), ("window_right", _i32),
        ("softmax_scale", _f32), ("softcap", _f32), ("seqused_k_add", _i32),
        ("cache_batch_idx", _vp), ("block_table", _vp), ("block_table_batch_stride", _i64),
        ("page_block_size", _i32), ("num_splits", _i32), ("p_dropout", _f32), ("reserved0", _i32),
        ("rng_state", _vp), ("randval", _vp),
        ("randval_batch_stride", _i64), ("randval_head_stride", _i64), ("randval_row_stride", _i64),
        ("workspace", _vp), ("workspace_bytes", _i64), ("leftpad_k", _vp), ("seqused_q", _vp),
    ]


class FaKvAppendParams(C.Structure):
    _fields_ = [
        ("knew", _vp), ("vnew", _vp), ("kcache", _vp), ("vcache", _vp),
        ("knew_batch_stride", _i64), ("knew_row_stride", _i64), ("knew_head_stride", _i64),
        ("vnew_batch_stride", _i64), ("vnew_row_stride", _i64), ("vnew_head_stride", _i64),
        ("kcache_batch_stride", _i64), ("kcache_row_stride", _i64), ("kcache_head_stride", _i64),
        ("vcache_batch_stride", _i64), ("vcache_row_stride", _i64), ("vcache_head_stride", _i64),
        ("seqlens_k", _vp), ("cache_batch_idx", _vp), ("block_table", _vp), ("block_table_batch_stride", _i64),
        ("page_block_size", _i32), ("b", _i32), ("seqlen_new", _i32), ("h_k", _i32), ("d", _i32),
        ("dtype", _i32), ("reserved", _i32 * 2),
    ]


class FaRotaryParams(C.Structure):
    _fields_ = [
        ("x", _vp), ("y", _vp), ("cos", _vp), ("sin", _vp), ("seqlen_offsets", _vp),
        ("x_batch_stride", _i64), ("x_row_stride", _i64), ("x_head_stride", _i64),
        ("y_batch_stride", _i64), ("y_row_stride", _i64), ("y_head_stride", _i64), ("cos_row_stride", _i64),
        ("b", _i32), ("s", _i32), ("h", _i32), ("d", _i32), ("rotary_dim", _i32), ("seqlen_ro", _i32),
        ("interleaved", _i32), ("per_token", _i32), ("dtype", _i32), ("reserved", _i32 * 3),
    ]


class FaBwdParams(C.Structure):
    _fields_ = [
        ("dout", _vp), ("q", _vp), ("k", _vp), ("v", _vp), ("o", _vp), ("softmax_lse", _vp),
        ("dq", _vp), ("dk", _vp), ("dv", _vp), ("softmax_d", _vp),
        ("workspace", _vp), ("workspace_bytes", _i64),
        ("do_batch_stride", _i64), ("do_row_stride", _i64), ("do_head_stride", _i64),
        ("q_batch_stride", _i64), ("q_row_stride", _i64), ("q_head_stride", _i64),
        ("k_batch_stride", _i64), ("k_row_stride", _i64), ("k_head_stride", _i64),
        ("v_batch_stride", _i64), ("v_row_stride", _i64), ("v_head_stride", _i64),
        ("o_batch_stride", _i64), ("o_row_stride", _i64), ("o_head_stride", _i64),
        ("dq_batch_stride", _i64), ("dq_row_stride", _i64), ("dq_head_stride", _i64),
        ("dk_batch_stride", _i64), ("dk_row_stride", _i64), ("dk_head_stride", _i64),
        ("dv_batch_stride", _i64), ("dv_row_stride", _i64), ("dv_head_stride", _i64),
        ("cu_seqlens_q", _vp), ("cu_seqlens_k", _vp),
        ("alibi_slopes", _vp), ("alibi_batch_stride", _i64),
        ("b", _i32), ("h", _i32), ("h_k", _i32), ("d", _i32),
        ("seqlen_q", _i32), ("seqlen_k", _i32), ("total_q", _i32), ("total_k", _i32),
        ("dtype", _i32), ("is_causal", _i32), ("window_left", _i32), ("window_right", _i32),
        ("softmax_scale", _f32), ("softcap", _f32), ("deterministic", _i32),
        ("p_dropout", _f32), ("reserved", _i32 * 3), ("rng_state", _vp), ("seqused_q", _vp), ("seqused_k", _vp),
    ]


EXPORTS = (
    "fa_abi_version", "fa_sizeof_fwd_params", "fa_sizeof_bwd_params", "fa_sizeof_kvappend_params", "fa_sizeof_rotary_params",
    "fa_last_error", "fa_rotary", "fa_knobs_reload", "fa_last_schedule", "fa_last_kernel_name", "fa_fwd_schedule_query", "fa_bwd_dq_schedule_query", "fa_bwd_plan_query",
    "fa_fwd", "fa_varlen_fwd", "fa_fwd_kvcache", "fa_kvcache_append", "fa_set_rng_state", "fa_fwd_workspace_bytes",
    "fa_bwd_workspace_bytes", "fa_bwd", "fa_varlen_bwd", "fa_bwd_fused_status",
)

_LIB = None


def library_path() -> str:
    env = os.environ.get("FA_GFX950_LIB")
    if env:
        return env
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libfa_gfx950.so")


def load():
    """dlopen libfa_gfx950.so and type its entry points (raises if absent or mismatched)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `python flash-attention_amd/build.py` "
            "(the gfx950 attention path has no CPU or PyTorch fallback)")
    lib = C.CDLL(path)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise ImportError(f"{path} does not export {name}")
    lib.fa_abi_version.restype = C.c_int
    lib.fa_sizeof_fwd_params.restype = C.c_int
    lib.fa_sizeof_bwd_params.restype = C.c_int
    lib.fa_last_error.restype = C.c_char_p
    lib.fa_sizeof_kvappend_params.restype = C.c_int
    lib.fa_kvcache_append.argtypes = [C.POINTER(FaKvAppendParams), C.c_void_p]
    lib.fa_kvcache_append.restype = C.c_int
    for fn in (lib.fa_fwd, lib.fa_varlen_fwd, lib.fa_fwd_kvcache):
        fn.argtypes = [C.POINTER(FaFwdParams), C.c_void_p]
        fn.restype = C.c_int
    for fn in (lib.fa_bwd, lib.fa_varlen_bwd):
        fn.argtypes = [C.POINTER(FaBwdParams), C.c_void_p]
        fn.restype = C.c_int
    lib.fa_sizeof_rotary_params.restype = C.c_int
    lib.fa_knobs_reload.argtypes = []
    lib.fa_knobs_reload.restype = None
    lib.fa_last_schedule.argtypes = [C.POINTER(C.c_int32), C.c_int]
    lib.fa_last_schedule.restype = C.c_int
    lib.fa_last_kernel_name.restype = C.c_char_p
    lib.fa_fwd_schedule_query.argtypes = [C.POINTER(FaFwdParams), C.c_int]
    lib.fa_fwd_schedule_query.restype = C.c_int
    lib.fa_bwd_dq_schedule_query.argtypes = [C.POINTER(FaBwdParams)]
    lib.fa_bwd_dq_schedule_query.restype = C.c_int
    lib.fa_bwd_plan_query.argtypes = [C.POINTER(FaBwdParams), C.POINTER(C.c_int32), C.c_int]
    lib.fa_bwd_plan_query.restype = C.c_int
    lib.fa_rotary.argtypes = [C.POINTER(FaRotaryParams), C.c_void_p]
    lib.fa_rotary.restype = C.c_int
    lib.fa_set_rng_state.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.fa_set_rng_state.restype = C.c_int
    lib.fa_fwd_workspace_bytes.argtypes = [C.POINTER(FaFwdParams)]
    lib.fa_fwd_workspace_bytes.restype = C.c_int64
    lib.fa_bwd_fused_status.argtypes = [C.POINTER(FaBwdParams), C.c_void_p]
    lib.fa_bwd_fused_status.restype = C.c_int
    lib.fa_bwd_workspace_bytes.argtypes = [C.POINTER(FaBwdParams)]
    lib.fa_bwd_workspace_bytes.restype = C.c_int64
    if lib.fa_abi_version() != FA_ABI_VERSION:
        raise ImportError(f"{path}: ABI version {lib.fa_abi_version()} != binder {FA_ABI_VERSION}")
    if (lib.fa_sizeof_fwd_params() != C.sizeof(FaFwdParams) or lib.fa_sizeof_bwd_params() != C.sizeof(FaBwdParams)
            or lib.fa_sizeof_kvappend_params() != C.sizeof(FaKvAppendParams)
            or lib.fa_sizeof_rotary_params() != C.sizeof(FaRotaryParams)):
        raise ImportError(f"{path}: parameter-block size mismatch with the ctypes mirror")
    _LIB = lib
    return lib


def check(rc: int) -> None:
    """Map a negative return code to the exception class the reference host layer raises (RuntimeError)."""
    if rc != FA_OK:
        msg = load().fa_last_error().decode("utf-8", "replace")
        raise RuntimeError(msg or f"libfa_gfx950 error {rc}")
