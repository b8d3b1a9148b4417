import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be
from tests._util import attention_torch
os.environ["FA_FWD_NW"] = "64"; be.reload_knobs()
d = 128
for dt in (torch.bfloat16,):
  for sq in (64, 256):
    for sk in (64, 128, 192, 256, 320, 1024):
        torch.manual_seed(0)
        q = torch.randn(1, sq, 1, d, device="cuda", dtype=dt); k = torch.randn(1, sk, 1, d, device="cuda", dtype=dt); v = torch.randn_like(k)
        outs = []
        for rep in range(3):
            out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, d ** -0.5, False, -1, -1, 0.0, False, None)
            outs.append((out.clone(), lse.clone()))
        ref, lse_ref = attention_torch(q.float(), k.float(), v.float(), False, (-1, -1), upcast=True)
        err = (outs[0][0].float() - ref).abs().amax(dim=(0, 2, 3))  # per row
        det = all(torch.equal(outs[0][0], o[0]) for o in outs[1:])
        el = (outs[0][1] - lse_ref).abs()[0, 0]
        rb = [float(err[i:i + 32].max()) for i in range(0, sq, 32)]
        lb = [float(el[i:i + 32].max()) for i in range(0, sq, 32)]
        print(f"sq={sq} sk={sk} det={det} out-err per 32-row block: {['%.1e' % x for x in rb]} lse-err: {['%.1e' % x for x in lb]}", flush=True)
