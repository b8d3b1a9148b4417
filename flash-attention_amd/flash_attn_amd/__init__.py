"""MI355X-native fused attention (gfx950 HIP kernels) behind the flash_attn Python interface."""
__version__ = "0.1.0"

from .flash_attn_interface import (  # noqa: F401
    flash_attn_func,
    flash_attn_kvpacked_func,
    flash_attn_padded_func,
    flash_attn_qkvpacked_func,
    flash_attn_varlen_func,
    flash_attn_varlen_kvpacked_func,
    flash_attn_varlen_qkvpacked_func,
    flash_attn_with_kvcache,
)
