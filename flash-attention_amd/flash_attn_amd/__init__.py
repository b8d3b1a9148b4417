"""MI355X-native fused attention (gfx950 HIP kernels) behind the flash_attn Python interface."""
__version__ = "0.1.0"
