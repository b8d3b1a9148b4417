import os, sys, statistics
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/flash-attention_amd")
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
torch.manual_seed(0)
for (B, S, H, Hk, D, wl) in ((2, 8192, 32, 8, 128, 1024), (2, 8192, 32, 8, 128, 4096), (2, 8192, 32, 32, 128, 1024), (4, 4096, 32, 32, 128, 512), (2, 8192, 32, 8, 128, 256)):
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn(B, S, Hk, D, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k)
    # visible keys per row: min(i + 1, wl + 1)
    vis = sum(min(i + 1, wl + 1) for i in range(S))
    fl = 4 * B * H * vis * D
    line = f"B{B} S{S} H{H}/{Hk} wl{wl}:"
    for nw in ("0", "64", "34", "38", "0", "64"):
        os.environ["FA_FWD_NW"] = nw; be.reload_knobs()
        ms = t_ms(lambda: be.fwd(q, k, v, None, None, 0.0, D ** -0.5, True, wl, 0, 0.0, False, None))
        line += f"  [nw {nw}: k{be.last_schedule()['fwd_kernel']}/{be.last_schedule()['fwd_nw']}] {ms:.3f} ms {fl / ms / 1e9:.0f} TF"
    print(line, flush=True)
