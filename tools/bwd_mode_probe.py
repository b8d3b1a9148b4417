"""Does the dS-spill (5-contraction) backward win when its dS fits the 256 MB Infinity Cache?  FA_BWD_MODE=0 (7 contractions) vs 2 at growing dS footprints
(experiments/build_experiments.py's library, FA_GFX950_LIB).  dS touched under a causal mask = B*H*S*S bytes (bf16, half of the square)."""
import os, sys, statistics
sys.path.insert(0, "flash-attention_amd")
import torch
from flash_attn_amd import backend as be
def t(fn, reps=5):
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
for (B, S, H, D, c) in ((2, 2048, 16, 128, True), (4, 2048, 16, 128, True), (8, 2048, 16, 128, True), (16, 2048, 16, 128, True), (1, 4096, 8, 128, True), (1, 4096, 16, 128, True), (1, 4096, 32, 128, True), (4, 4096, 32, 128, True)):
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
    o, l, _, _ = be.fwd(q, k, v, None, None, 0.0, D ** -0.5, c, -1, -1, 0.0, False, None)
    do = torch.randn_like(o); dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    g = lambda: be.bwd(do, q, k, v, o, l, dq, dk, dv, None, 0.0, D ** -0.5, c, -1, -1, 0.0, False, None, None)
    line = f"B={B} S={S} H={H} causal={int(c)} dS touched {B * H * S * S / (2 if c else 1) * 2 / 2 ** 20 / 2 * 2:.0f} MiB:"
    for mode in ("0", "2"):
        os.environ["FA_BWD_MODE"] = mode; be.reload_knobs()
        g(); ms = statistics.median([t(g) for _ in range(5)])
        line += f"  [mode {mode}] {ms:.3f} ms {2.5 * 4 * B * H * S * S * D / (2 if c else 1) / ms / 1e9:.0f} TF (spill={be.last_schedule().get('bwd_spill')})"
    print(line, flush=True)
