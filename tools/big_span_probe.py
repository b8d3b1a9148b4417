"""Views whose ROWS are 2 MB apart (head 0..1 of a (1, S, 8192, 128) buffer): one (batch, head) slice spans 8.6 GB, past what the 64-per-wave kernels' 32-bit
offsets reach -- the dispatch must fall back (forward: pipelined kernel; backward: lock-step kernels, which form a tile's base in 64 bits) and the results must
equal the same call on contiguous copies bit for bit ... when pinned to the same kernels.  usage: big_span_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be
S, HB, H, D = 4096, 8192, 2, 128
ok_all = True
bufs = {n: torch.empty(1, S, HB, D, device="cuda", dtype=torch.bfloat16) for n in ("q", "k", "v", "do", "dq", "dk", "dv")}
for causal in (True, False):
    torch.manual_seed(1)
    small = {n: torch.randn(1, S, H, D, device="cuda", dtype=torch.bfloat16) for n in ("q", "k", "v", "do")}
    view = {n: bufs[n][:, :, :H] for n in bufs}
    for n in small: view[n].copy_(small[n])
    assert view["k"].stride(1) * S * 2 > 2 ** 32
    def run(t, dq, dk, dv):
        out, lse, _, _ = be.fwd(t["q"], t["k"], t["v"], None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None)
        name = be.last_schedule()["name"]
        be.bwd(t["do"], t["q"], t["k"], t["v"], out, lse, dq, dk, dv, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None, None)
        return (out, lse, dq, dk, dv), name, dict(be.last_schedule())
    for k_ in ("FA_FWD_NW", "FA_BWD_DQ_NW", "FA_BWD_DKDV"): os.environ.pop(k_, None)
    be.reload_knobs()
    for n in ("dq", "dk", "dv"): view[n].fill_(float("nan"))
    r_view, name_v, sch_v = run(view, view["dq"], view["dk"], view["dv"])
    # the same kernels on contiguous copies (where the default would be the 64-per-wave ones)
    os.environ["FA_FWD_NW"] = "38"; os.environ["FA_BWD_DQ_NW"] = str(sch_v["bwd_dq_nw"]); os.environ["FA_BWD_DKDV"] = str(sch_v["bwd_dkdv_nw"]); be.reload_knobs()
    r_small, name_s, sch_s = run(small, *[torch.empty_like(small["q"]) for _ in range(3)])
    same = [bool(torch.equal(a, b)) for a, b in zip(r_small, r_view)]
    fell_back = ("w64" not in name_v) and sch_v["bwd_dq_nw"] != 64 and sch_v["bwd_dkdv_nw"] != 64
    ok_all &= all(same) and fell_back
    print(f"causal{int(causal)}: view -> {name_v}, dq/dkdv {sch_v['bwd_dq_nw']}/{sch_v['bwd_dkdv_nw']} (fell back: {fell_back}); copies -> {name_s}; out,lse,dq,dk,dv equal: {same}", flush=True)
print("OK" if ok_all else "FAILED")
