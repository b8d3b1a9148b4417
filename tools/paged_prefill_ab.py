"""Prefill over a PAGED cache (varlen_fwd with block_table, shuffled pages): the 64-rows-per-wave kernel's paged variant (default where plain attention takes that
kernel) against the lock-step kernel (FA_FWD_NW=8) and against the same shapes on a contiguous cache."""
import os, sys, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be


def t(fn, reps=5):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(5):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return statistics.median(ts)


for (B, S, H, Hk, D, causal, page) in ((4, 4096, 32, 32, 128, True, 256), (4, 4096, 32, 8, 128, True, 256), (2, 8192, 32, 8, 128, True, 512), (8, 2048, 16, 16, 64, False, 256), (1, 16384, 16, 16, 128, False, 256)):
    per = S // page
    kp = torch.randn(B * per, page, Hk, D, device="cuda", dtype=torch.bfloat16); vp = torch.randn_like(kp)
    table = torch.randperm(B * per, device="cuda").reshape(B, per).to(torch.int32)
    cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device="cuda")
    q = torch.randn(B * S, H, D, device="cuda", dtype=torch.bfloat16)
    kc, vc = kp.reshape(B * S, Hk, D), vp.reshape(B * S, Hk, D)
    fl = 4 * B * H * S * S * D / (2 if causal else 1)
    row = []
    for name, fn in (("paged", lambda: be.varlen_fwd(q, kp, vp, None, cu, cu, None, None, table, None, S, S, 0.0, D ** -0.5, False, causal, -1, -1, 0.0, False, None)),
                     ("contiguous", lambda: be.varlen_fwd(q, kc, vc, None, cu, cu, None, None, None, None, S, S, 0.0, D ** -0.5, False, causal, -1, -1, 0.0, False, None))):
        for nw in (None, "8"):
            if nw: os.environ["FA_FWD_NW"] = nw
            else: os.environ.pop("FA_FWD_NW", None)
            be.reload_knobs()
            m = t(fn)
            row.append(f"{name} {be.last_schedule()['name'].split('::')[-1]}: {m:.3f} ms {fl / m / 1e9:.0f} TF")
    print(f"B={B} S={S} H={H}/{Hk} D={D} causal={int(causal)} page={page} | " + " | ".join(row), flush=True)
