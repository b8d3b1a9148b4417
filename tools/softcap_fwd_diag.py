"""Diagnostic: softcap forward on the 64-rows-per-wave kernel against the lock-step kernel and the fp64 oracle on a few shapes: max errors and where they sit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be
from oracle import attention_oracle as orc
def setk(v):
    if v is None: os.environ.pop("FA_FWD_NW", None)
    else: os.environ["FA_FWD_NW"] = v
    be.reload_knobs()
for (B, Sq, Sk, H, Hk, D, cap, scale_in, causal) in ((2, 1024, 1024, 4, 4, 128, 30.0, 6.0, False), (2, 1024, 1024, 4, 4, 128, 30.0, 6.0, True), (2, 2048, 2048, 4, 2, 64, 30.0, 6.0, False), (2, 1024, 1024, 4, 4, 128, 30.0, 3.0, False), (1, 256, 256, 1, 1, 128, 30.0, 6.0, False)):
    torch.manual_seed(B * Sq + D + int(cap))
    q = torch.randn(B, Sq, H, D, device="cuda", dtype=torch.bfloat16) * scale_in
    k = torch.randn(B, Sk, Hk, D, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k)
    res = {}
    for nw in ("64", "8"):
        setk(nw); o, l = be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, -1, -1, cap, False, None)[:2]; res[nw] = (o.float().cpu(), l.cpu(), be.last_schedule()["name"])
    ref, lse_ref = orc.attention_fwd(q, k, v, D ** -0.5, causal, (-1, -1), cap, None)
    ref, lse_ref = torch.from_numpy(ref).float(), torch.from_numpy(lse_ref).float()
    print(f"== B{B} Sq{Sq} Sk{Sk} H{H}/{Hk} D{D} cap{cap} x{scale_in} causal{int(causal)}")
    for nw in ("64", "8"):
        o, l, name = res[nw]
        eo = (o - ref).abs(); el = (l - lse_ref).abs()
        io = torch.nonzero(eo == eo.max())[0].tolist(); il = torch.nonzero(el == el.max())[0].tolist()
        print(f"  {name}: out err {float(eo.max()):.4f} at {io} (ref {float(ref[tuple(io)]):.3f}) | lse err {float(el.max()):.4f} at {il} (ref {float(lse_ref[tuple(il)]):.3f}); rows with lse err > 0.02: {int((el > 0.02).sum())} of {el.numel()}; nan {int(torch.isnan(o).sum())}")
