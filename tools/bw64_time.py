"""Runs the backward a few times per shape (for rocprofv3 --kernel-trace: tools/ablate_bw64.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be
torch.manual_seed(0)
shapes = os.environ.get("BW_SHAPES", "4,4096,32,1;4,4096,32,0")
for sh in shapes.split(";"):
    B, S, H, causal = [int(x) for x in sh.split(",")]
    q = torch.randn(B, S, H, 128, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q); do = torch.randn_like(q)
    out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, 128 ** -0.5, bool(causal), -1, -1, 0.0, False, None)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    for _ in range(6):
        be.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, 128 ** -0.5, bool(causal), -1, -1, 0.0, False, None, None)
    torch.cuda.synchronize()
