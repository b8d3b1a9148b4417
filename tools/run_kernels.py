"""Launch the hot-path kernels a few times (for rocprofv3 wrapping).  usage: run_kernels.py [fwd|bwd|all] [causal 0/1] [S] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be

what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
causal = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
S = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
B, H, D = 4, 32, 128
torch.manual_seed(0)
q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
sc = D ** -0.5
out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, sc, causal, -1, -1, 0.0, False, None)
do = torch.randn_like(out); dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
for _ in range(reps):
    if what in ("fwd", "all"):
        be.fwd(q, k, v, None, None, 0.0, sc, causal, -1, -1, 0.0, False, None)
    if what in ("bwd", "all"):
        be.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, sc, causal, -1, -1, 0.0, False, None, None)
torch.cuda.synchronize()
