"""ALiBi backward at config 3 (and D = 64): per-kernel times through the C ABI.  Run once per library to A/B two builds on one box:
FA_GFX950_LIB=gpurun_abl/libfa_head.so python tools/alibi_bwd_ab.py; python tools/alibi_bwd_ab.py"""
import os, sys, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(5):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return statistics.median(ts)
tag = os.path.basename(os.environ.get("FA_GFX950_LIB", "in-tree"))
for (B, S, H, D) in ((4, 4096, 32, 128), (4, 4096, 32, 64)):
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q); do = torch.randn_like(q)
    for name, al in (("alibi", torch.tensor([2.0 ** (-8.0 * (i + 1) / H) for i in range(H)], device="cuda")), ("plain", None)):
        out, lse = be.fwd(q, k, v, None, al, 0.0, D ** -0.5, True, -1, -1, 0.0, False, None)[:2]
        m = t(lambda: be.bwd(do, q, k, v, out, lse, None, None, None, al, 0.0, D ** -0.5, True, -1, -1, 0.0, False, None, None))
        print(f"[{tag}] bwd causal B={B} S={S} H={H} D={D} {name}: {m:.3f} ms {2.5 * 4 * B * H * S * S * D / 2 / m / 1e9:.0f} TF", flush=True)
