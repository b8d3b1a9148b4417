"""Timings of the 5-contraction backward against the recomputing pair over caps / mixes, in one process.  usage: c5_time.py [shape ...] with shape = B,S,H,Hk,D,causal"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be
def t_ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(5):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return statistics.median(ts)
shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(4, 4096, 32, 32, 128, 1)]
variants = [v.split(":") for v in os.environ.get("C5_VARIANTS", "-1:0:1,5:1024:0,5:2048:0,5:512:0,-1:0:1").split(",")]
for (B, S, H, Hk, D, causal) in shapes:
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn(B, S, Hk, D, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k); do = torch.randn_like(q)
    out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, D ** -0.5, bool(causal), -1, -1, 0.0, False, None)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    fl = 10 * B * H * S * S * D / (2 if causal else 1)
    line = f"bwd B{B} S{S} H{H}/{Hk} D{D} c{causal}:"
    for mode, cap, mix in variants:
        os.environ["FA_BWD_MODE"] = mode; os.environ["FA_BWD_C5_CAP_MB"] = cap if int(cap) else "1024"; os.environ["FA_BWD_C5_MIX"] = mix
        be.reload_knobs()
        ms = t_ms(lambda: be.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, D ** -0.5, bool(causal), -1, -1, 0.0, False, None, None))
        line += f"  [{mode}/{cap}/{mix}] {ms:.3f} ms {fl / ms / 1e9:.0f} TF"
    print(line, flush=True)
