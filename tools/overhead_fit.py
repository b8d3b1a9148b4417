"""Per-block fixed cost of the forward: time vs number of key tiles at fixed query count (non-causal), linear fit."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be

def t_ms(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return statistics.median(ts)

B, H, Sq, D = 4, 32, 4096, 128
q = torch.randn(B, Sq, H, D, device="cuda", dtype=torch.bfloat16)
for nw in ("34", "38"):
    os.environ["FA_FWD_NW"] = nw; be.reload_knobs()
    xs, ys = [], []
    for Sk in (64, 128, 256, 512, 1024, 2048, 4096):
        k = torch.randn(B, Sk, H, D, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k)
        ms = t_ms(lambda: be.fwd(q, k, v, None, None, 0.0, D ** -0.5, False, -1, -1, 0.0, False, None))
        xs.append(Sk / 64); ys.append(ms * 1e3)
        print(f"nw={nw} Sk={Sk:5d} tiles={Sk // 64:3d}: {ms * 1e3:8.1f} us  ({4 * B * H * Sq * Sk * D / ms / 1e9:7.1f} TF)")
    n = len(xs); mx, my = sum(xs[2:]) / (n - 2), sum(ys[2:]) / (n - 2)
    b = sum((x - mx) * (y - my) for x, y in zip(xs[2:], ys[2:])) / sum((x - mx) ** 2 for x in xs[2:])
    print(f"nw={nw}: fit over tiles>=4: {my - b * mx:.1f} us fixed + {b:.2f} us per tile  (blocks per CU: {B * H * Sq / (32 * (int(nw) - 30)) / 256:.0f})")
