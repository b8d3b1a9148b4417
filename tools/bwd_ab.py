import os, sys, statistics
sys.path.insert(0, "flash-attention_amd")
import torch
from flash_attn_amd import backend as be
def t(fn, reps=5):
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
for (B, S, H, D, c) in ((4, 4096, 32, 128, True), (4, 4096, 32, 128, False), (8, 2048, 16, 64, False), (16, 1024, 16, 128, True)):
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
    o, l, _, _ = be.fwd(q, k, v, None, None, 0.0, D ** -0.5, c, -1, -1, 0.0, False, None)
    do = torch.randn_like(o); dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    g = lambda: be.bwd(do, q, k, v, o, l, dq, dk, dv, None, 0.0, D ** -0.5, c, -1, -1, 0.0, False, None, None)
    g(); ms = statistics.median([t(g) for _ in range(5)])
    print(f"bwd B={B} S={S} H={H} D={D} causal={int(c)}: {ms:.3f} ms {2.5 * 4 * B * H * S * S * D / (2 if c else 1) / ms / 1e9:.0f} TF")
