import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be
from tests._util import attention_torch
os.environ["FA_FWD_NW"] = "64"; be.reload_knobs()
for (d, dt, sq, sk, window) in ((64, torch.float16, 512, 256, (132, 95)), (64, torch.float16, 257, 129, (45, 127)), (64, torch.float16, 256, 256, (69, 93))):
    torch.manual_seed(sq * 31 + sk)
    B, H, Hk = 2, 4, 2
    q = torch.randn(B, sq, H, d, device="cuda", dtype=dt); k = torch.randn(B, sk, Hk, d, device="cuda", dtype=dt); v = torch.randn_like(k)
    out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, d ** -0.5, False, window[0], window[1], 0.0, False, None)
    ref, lse_ref = attention_torch(q.float(), k.float(), v.float(), False, window, upcast=True)
    err = (out.float() - ref).abs().amax(dim=(0, 2, 3))
    rb = [float(err[i:i + 32].max()) for i in range(0, sq, 32)]
    print(f"d={d} sq={sq} sk={sk} w={window}: per 32-row block: {['%.0e' % x for x in rb]}", flush=True)
