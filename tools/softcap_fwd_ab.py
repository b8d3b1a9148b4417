"""Softcap / dropout forward through the C ABI: the 64-rows-per-wave variants (default where plain attention takes that kernel) against the lock-step kernel (FA_FWD_NW=8) on one box.
usage: softcap_fwd_ab.py [softcap|dropout ...]   (default: both)"""
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


for (B, S, H, D, causal) in ((4, 4096, 32, 128, True), (4, 4096, 32, 128, False), (1, 16384, 16, 128, False), (8, 2048, 16, 64, False), (4, 4096, 32, 64, True)):
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
    row = []
    feats = [("plain", 0.0, 0.0)] + [f for f in (("softcap", 30.0, 0.0), ("dropout", 0.0, 0.1)) if len(sys.argv) < 2 or f[0] in sys.argv[1:]]
    for fname, cap, pd in feats:
        for nw in (None, "8"):
            if nw: os.environ["FA_FWD_NW"] = nw
            else: os.environ.pop("FA_FWD_NW", None)
            be.reload_knobs()
            m = t(lambda: be.fwd(q, k, v, None, None, pd, D ** -0.5, causal, -1, -1, cap, False, None))
            fl = 4 * B * H * S * S * D / (2 if causal else 1)
            row.append(f"{fname} {be.last_schedule()['name'].split('::')[-1]}: {m:.3f} ms {fl / m / 1e9:.0f} TF")
    print(f"B={B} S={S} H={H} D={D} causal={int(causal)} | " + " | ".join(row), flush=True)
