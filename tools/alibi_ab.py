"""Causal ALiBi forward: the 64-rows-per-wave variant (descending key walk) against the lock-step kernel that served it before (FA_FWD_NW = 64 | 8) and plain."""
import os, sys, statistics
sys.path.insert(0, "flash-attention_amd")
import torch
from flash_attn_amd import backend as be
def t(fn, reps=10):
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
for (B, S, H, D) in ((4, 4096, 32, 128), (8, 2048, 16, 128), (1, 16384, 16, 128), (4, 4096, 32, 64)):
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
    sl = torch.tensor([2.0 ** (-8.0 * (i + 1) / H) for i in range(H)], device="cuda", dtype=torch.float32)   # the standard ALiBi slopes
    line = f"causal ALiBi fwd B={B} S={S} H={H} D={D}:"
    for nw, al in (("64", sl), ("8", sl), ("64", None)):
        os.environ["FA_FWD_NW"] = nw; be.reload_knobs()
        f = lambda: be.fwd(q, k, v, None, al, 0.0, D ** -0.5, True, -1, -1, 0.0, False, None)
        f(); ms = statistics.median([t(f) for _ in range(5)])
        line += f"  [{'alibi' if al is not None else 'plain'} nw={nw}: {be.last_schedule()['name'][4:]}] {ms:.3f} ms {4 * B * H * S * S * D / 2 / ms / 1e9:.0f} TF"
    print(line, flush=True)
