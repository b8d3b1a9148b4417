import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be
B, S, H, D = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
causal = bool(int(sys.argv[5])); reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
torch.manual_seed(0)
q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
for _ in range(reps):
    be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None)
torch.cuda.synchronize()
