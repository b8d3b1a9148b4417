import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be
os.environ["FA_FWD_NW"] = os.environ.get("FA_FWD_NW", "64"); be.reload_knobs()
def t_ms(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return statistics.median(ts)
torch.manual_seed(0)
for (B, S, H, D, causal) in ((1, 16384, 16, 128, False), (4, 4096, 32, 128, True), (4, 4096, 32, 128, False)):
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
    ms = t_ms(lambda: be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None))
    fl = 4 * B * H * S * S * D / (2 if causal else 1)
    lse = be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None)[1]
    l4 = lse.reshape(B, H, S // 4, 4).float()
    pro, loop, epi = float(l4[..., 1].mean()), float(l4[..., 2].mean()), float(l4[..., 3].mean())
    print(f"S={S} c={int(causal)}: {ms:.3f} ms {fl / ms / 1e9:6.0f} TF clk/MFMA {float(l4[..., 0].mean()):.1f} prologue {pro:.0f} loop {loop:.0f} epilogue {epi:.0f} clk |")
