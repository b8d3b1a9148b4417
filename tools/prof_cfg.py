"""Run one BASELINE config's forward and backward a few times (for rocprofv3 wrapping).  usage: prof_cfg.py B S H Hk D causal wl wr [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be
B, S, H, Hk, D, causal, wl, wr = [int(x) for x in sys.argv[1:9]]
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 4
torch.manual_seed(0)
q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn(B, S, Hk, D, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k)
sc = D ** -0.5
out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, sc, bool(causal), wl, wr, 0.0, False, None)
do = torch.randn_like(out); dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
for _ in range(reps):
    be.fwd(q, k, v, None, None, 0.0, sc, bool(causal), wl, wr, 0.0, False, None)
    be.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, sc, bool(causal), wl, wr, 0.0, False, None, None)
torch.cuda.synchronize()
print(be.last_schedule())
