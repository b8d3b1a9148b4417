"""Forward schedules against each other on given shapes (FA_FWD_NW: 0 = the heuristic, 64 = 64 rows per wave, 34 / 38 = pipelined 4 / 8 waves), one process.
usage: fwd_nw_ab.py B,S,H,Hk,D,causal ..."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be
def t_ms(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(5):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return statistics.median(ts)
for a in sys.argv[1:]:
    B, S, H, Hk, D, causal = (int(x) for x in a.split(","))
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn(B, S, Hk, D, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k)
    fl = 4 * B * H * S * S * D / (2 if causal else 1)
    line = f"fwd B{B} S{S} H{H}/{Hk} D{D} c{causal}:"
    for nw in os.environ.get("NWS", "0,64,34,38,0").split(","):
        os.environ["FA_FWD_NW"] = nw; be.reload_knobs()
        try:
            ms = t_ms(lambda: be.fwd(q, k, v, None, None, 0.0, D ** -0.5, bool(causal), -1, -1, 0.0, False, None))
            line += f"  [nw {nw} -> {be.last_schedule()['fwd_nw']}] {fl / ms / 1e9:.0f}"
        except Exception as e:
            line += f"  [nw {nw}] n/a"
    print(line, flush=True)
