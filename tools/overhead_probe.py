import os, sys, statistics
sys.path.insert(0, "/root/repo/flash-attention_amd")
import torch
from flash_attn_amd import backend as be
def t_ms(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts=[]
    for _ in range(5):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/reps)
    return statistics.median(ts)
H, D = 16, 128
for nw in ("34", "38", "4"):
    os.environ["FA_FWD_NW"] = nw; be.reload_knobs()
    for S in (256, 512, 1024, 2048, 4096):
        B = 262144 // S  # constant total rows = 256k per head
        B = max(1, B // 16)
        q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
        ms = t_ms(lambda: be.fwd(q, k, v, None, None, 0.0, D ** -0.5, False, -1, -1, 0.0, False, None))
        bm = 128 if nw in ("34", "4") else 256
        nblk = B * H * ((S + bm - 1) // bm); tiles = S // 64
        per_cu = nblk / 256.0
        print(f"NW={nw} S={S} B={B}: {ms*1e3:8.1f} us  blocks={nblk} ({per_cu:.1f}/CU) tiles/block={tiles}  us per block-slot={ms*1e3/per_cu:6.2f}  TF={4*B*H*S*S*D/ms/1e9:7.1f}", flush=True)
