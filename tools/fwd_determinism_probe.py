"""Run-to-run probe of the 64-rows-per-wave forward: the same call N times, bitwise comparison of out / lse with the first run, contiguous and paged keys."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import numpy as np, torch
from flash_attn_amd import backend as be
os.environ["FA_FWD_NW"] = os.environ.get("FA_FWD_NW", "64"); be.reload_knobs()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(0)
H, hk, d, page = 8, 2, 128, 256
lens_q = [700, 1300, 64, 2048, 1]; lens_k = [700, 1377, 1024, 2348, 513]
per = (max(lens_k) + page - 1) // page; B = len(lens_q)
kp = torch.randn(B * per + 3, page, hk, d, device="cuda", dtype=torch.bfloat16); vp = torch.randn_like(kp)
table = torch.randperm(B * per + 3, device="cuda")[: B * per].reshape(B, per).to(torch.int32)
cu_q = torch.tensor([0] + list(np.cumsum(lens_q)), dtype=torch.int32, device="cuda"); cu_k = torch.tensor([0] + list(np.cumsum(lens_k)), dtype=torch.int32, device="cuda")
q = torch.randn(sum(lens_q), H, d, device="cuda", dtype=torch.bfloat16)
kc = torch.cat([kp[table[b].long()].reshape(per * page, hk, d)[: lens_k[b]] for b in range(B)]); vc = torch.cat([vp[table[b].long()].reshape(per * page, hk, d)[: lens_k[b]] for b in range(B)])
for name, (causal, wl, wr) in (("full", (False, -1, -1)), ("causal", (True, -1, -1)), ("local_causal", (True, 300, 0)), ("local", (False, 100, 200))):
    for kind in ("paged", "contiguous"):
        if kind == "paged":
            f = lambda: be.varlen_fwd(q, kp, vp, None, cu_q, cu_k, None, None, table, None, max(lens_q), max(lens_k), 0.0, d ** -0.5, False, causal, wl, wr, 0.0, False, None)[:2]
        else:
            f = lambda: be.varlen_fwd(q, kc, vc, None, cu_q, cu_k, None, None, None, None, max(lens_q), max(lens_k), 0.0, d ** -0.5, False, causal, wl, wr, 0.0, False, None)[:2]
        o0, l0 = f(); name_k = be.last_schedule()["name"]
        bad, worst = 0, 0.0
        for _ in range(N):
            o, l = f()
            if not (torch.equal(o, o0) and torch.equal(l, l0)):
                bad += 1; worst = max(worst, float((o.float() - o0.float()).abs().max()))
        print(f"{name:13s} {kind:10s} {name_k.split('::')[-1]:38s}: {bad} of {N} runs differ (max |diff| {worst:.3g})", flush=True)
