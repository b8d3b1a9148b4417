import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be
B, S = int(sys.argv[1]), int(sys.argv[2]); H, Hk, D = 32, 8, 128
q = torch.randn(B, 1, H, D, device="cuda", dtype=torch.bfloat16)
kc = torch.randn(B, S, Hk, D, device="cuda", dtype=torch.bfloat16); vc = torch.randn_like(kc)
lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
for _ in range(10):
    be.fwd_kvcache(q, kc, vc, None, None, lens, None, None, None, None, None, None, None, D ** -0.5, False, -1, -1, 0.0, True, 0)
torch.cuda.synchronize()
