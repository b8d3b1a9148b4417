"""Causal-ALiBi variant of the 64-rows-per-wave forward through the varlen entry: per-sequence, per-head, per-64-row error against the fp64 oracle,
next to the lock-step kernel's (FA_FWD_NW = 64 | 8).  Diagnostic for a failing tests/test_fwd_gpu.py::test_w64_causal_alibi_varlen."""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "flash-attention_amd")
import numpy as np, torch
from flash_attn_amd import backend as be
from oracle import attention_oracle as orc
torch.manual_seed(4)
H, D = 8, 128
lens = [1300, 70, 2048, 513, 900]
cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
tot = int(cu[-1])
q = torch.randn(tot, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
sl = (torch.rand(H, device="cuda") * 0.3 + 0.02).float()
res = {}
for nw in ("64", "8"):
    os.environ["FA_FWD_NW"] = nw; be.reload_knobs()
    res[nw] = be.varlen_fwd(q, k, v, None, cu, cu, None, None, None, sl, max(lens), max(lens), 0.0, D ** -0.5, False, True, -1, -1, 0.0, False, None)[:2]
    print(nw, be.last_schedule()["name"])
print("slopes", [f"{float(x):.3f}" for x in sl])
def chunks(err, n=64): return " ".join(f"{float(err[i:i + n].max()):.0e}" for i in range(0, err.shape[0], n))
for i in range(len(lens)):
    a, b_ = int(cu[i]), int(cu[i + 1])
    ref, lref = orc.attention_fwd(q[a:b_][None], k[a:b_][None], v[a:b_][None], D ** -0.5, True, (-1, -1), 0.0, sl.cpu().numpy())
    ref, lref = torch.from_numpy(ref[0]).float(), torch.from_numpy(lref[0]).float()
    for nw in ("64", "8"):
        out, lse = res[nw]
        e = (out[a:b_].float().cpu() - ref).abs().amax(-1)          # (rows, H)
        el = (lse[:, a:b_].cpu() - lref).abs()                       # (H, rows)
        print(f"seq {i} len {lens[i]} nw={nw}: out max {float(e.max()):.2e} (head {int(e.amax(0).argmax())}) lse max {float(el.max()):.2e} (head {int(el.amax(1).argmax())})")
        if nw == "64" and (float(e.max()) > 1.2e-2 or float(el.max()) > 8e-3):
            h = int(e.amax(0).argmax()); print("   out per 64 rows, worst head:", chunks(e[:, h]))
            h = int(el.amax(1).argmax()); print("   lse per 64 rows, worst head:", chunks(el[h]))
