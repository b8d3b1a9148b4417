"""Quick parity of the 64-rows-per-wave forward (FA_FWD_NW=64) against the fp32 torch reference, small and odd shapes first."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be
from tests._util import attention_torch, max_abs

os.environ["FA_FWD_NW"] = sys.argv[1] if len(sys.argv) > 1 else "64"
be.reload_knobs()
bad = 0
cases = []
for d in (128, 64):
    for dt in (torch.bfloat16, torch.float16):
        for (sq, sk) in ((64, 64), (256, 256), (113, 203), (300, 1000), (1024, 1024), (1023, 1024), (512, 256), (2048, 2048), (1, 300), (257, 129)):
            for mode in ("full", "causal", "local"):
                cases.append((d, dt, sq, sk, mode))
for (d, dt, sq, sk, mode) in cases:
    torch.manual_seed(sq * 31 + sk)
    B, H, Hk = 2, 4, 2
    q = torch.randn(B, sq, H, d, device="cuda", dtype=dt); k = torch.randn(B, sk, Hk, d, device="cuda", dtype=dt); v = torch.randn_like(k)
    causal = mode == "causal"
    window = (-1, -1)
    if mode == "local":
        g = torch.Generator().manual_seed(sq * 7 + sk); window = tuple(int(x) for x in torch.randint(0, sk, (2,), generator=g))
    out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, d ** -0.5, causal, window[0], window[1], 0.0, False, None)
    s = be.last_schedule()
    ref, lse_ref = attention_torch(q.float(), k.float(), v.float(), causal, window, upcast=True)
    pt, _ = attention_torch(q, k, v, causal, window, upcast=False, reorder=True)
    err, err_pt = max_abs(out.float(), ref), max_abs(pt.float(), ref)
    fin = torch.isfinite(lse_ref)
    okinf = torch.equal(torch.isposinf(lse), ~fin)
    el = max_abs(lse[fin], lse_ref[fin]) if fin.any() else 0.0
    ok = err <= 2 * err_pt + 1e-5 and okinf and el < 2e-3 and not torch.isnan(out).any()
    if not ok:
        bad += 1
    print(("ok  " if ok else "FAIL"), s["name"], d, str(dt)[6:], sq, sk, mode, window, f"err {err:.2e} pt {err_pt:.2e} lse {el:.2e} inf {okinf}", flush=True)
print("bad:", bad)
