"""Tensors whose batch slices sit more than 2^32 BYTES apart (views into 6.5 GB buffers: batch stride 819 M elements, batch 3 starts 4.9 GB in): forward and backward
on the strided views against the same call on small contiguous copies, bit for bit.  Every kernel forms a (batch, head) slice's base in 64 bits and addresses inside it
with 32-bit offsets / buffer descriptors; this checks the 64-bit half.  usage: big_offset_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be
B, SBIG, H, D = 4, 200000, 32, 128
bufs = {n: torch.empty(B, SBIG, H, D, device="cuda", dtype=torch.bfloat16) for n in ("q", "k", "v", "do", "dq", "dk", "dv")}
ok_all = True
for (S, causal, mode) in ((1024, True, "0"), (1024, True, "-1"), (4096, True, "0"), (2048, False, "0"), (640, True, "0")):
    os.environ["FA_BWD_MODE"] = mode; be.reload_knobs()
    torch.manual_seed(S)
    small = {n: torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16) for n in ("q", "k", "v", "do")}
    view = {n: bufs[n][:, :S] for n in bufs}
    for n in small: view[n].copy_(small[n])
    assert view["q"].stride(0) * 3 * 2 > 2 ** 32
    def run(t, dq, dk, dv):
        out, lse, _, _ = be.fwd(t["q"], t["k"], t["v"], None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None)
        name = be.last_schedule()["name"]
        be.bwd(t["do"], t["q"], t["k"], t["v"], out, lse, dq, dk, dv, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None, None)
        return (out, lse, dq, dk, dv), name, dict(be.last_schedule())
    r_small, name, sch = run(small, *[torch.empty_like(small["q"]) for _ in range(3)])
    for n in ("dq", "dk", "dv"): view[n].fill_(float("nan"))
    r_view, _, sch_v = run(view, view["dq"], view["dk"], view["dv"])
    same = [bool(torch.equal(a, b)) for a, b in zip(r_small, r_view)]
    ok_all &= all(same) and sch["bwd_spill"] == sch_v["bwd_spill"]
    print(f"S{S} causal{int(causal)} FA_BWD_MODE={mode}: {name} bwd_spill {sch_v['bwd_spill']} dq/dkdv {sch_v['bwd_dq_nw']}/{sch_v['bwd_dkdv_nw']}  out,lse,dq,dk,dv equal: {same}", flush=True)
del bufs, small, view, r_small, r_view
torch.cuda.empty_cache()
# ... and a packed batch whose LAST sequences start more than 2^32 bytes into the tensors: 150 x 4096 tokens, H = 32 -> 2.5 G elements per tensor; the last four
# sequences against the same four computed as a batch of their own
os.environ["FA_BWD_MODE"] = "0"; be.reload_knobs()
NS, SL, TAIL = 150, 4096, 4
T = NS * SL
torch.manual_seed(7)
big = {n: torch.randn(T, H, D, device="cuda", dtype=torch.bfloat16) for n in ("q", "k", "v", "do")}
assert (T - TAIL * SL) * H * D * 2 > 2 ** 32
cu = torch.arange(0, T + 1, SL, device="cuda", dtype=torch.int32)
def vrun(t, cu_):
    out, lse, _, _ = be.varlen_fwd(t["q"], t["k"], t["v"], None, cu_, cu_, None, None, None, None, SL, SL, 0.0, D ** -0.5, False, True, -1, -1, 0.0, False, None)
    name = be.last_schedule()["name"]
    dq, dk, dv = torch.empty_like(t["q"]), torch.empty_like(t["k"]), torch.empty_like(t["v"])
    be.varlen_bwd(t["do"], t["q"], t["k"], t["v"], out, lse, dq, dk, dv, cu_, cu_, None, SL, SL, 0.0, D ** -0.5, False, True, -1, -1, 0.0, False, None, None)
    return (out, dq, dk, dv), name, dict(be.last_schedule())
r_big, name, sch = vrun(big, cu)
tail = {n: big[n][-TAIL * SL:].clone() for n in big}
r_tail, _, _ = vrun(tail, cu[: TAIL + 1].clone())
same = [bool(torch.equal(a[-TAIL * SL:], b)) for a, b in zip(r_big, r_tail)]
ok_all &= all(same)
print(f"varlen {NS} x {SL}: {name} dq/dkdv {sch['bwd_dq_nw']}/{sch['bwd_dkdv_nw']} list {sch['bwd_list']}  last {TAIL} sequences' out,dq,dk,dv equal to their own batch: {same}", flush=True)
print("OK" if ok_all else "FAILED")
