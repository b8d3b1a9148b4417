"""Backward with ONE feature (softcap 30 / dropout 0.1 / causal ALiBi) at the config 3 shape and at D = 64: whole backward through the C ABI under knob settings, one
process, interleaved.  usage: feature_bwd_ab.py ["name=ENV=v,ENV=v;name2=..."]   (default: the tree's dispatch against the feature kernels of fa_bwd.hip)"""
import os, sys, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be


def t(fn, reps=4):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(5):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return statistics.median(ts)


spec = sys.argv[1] if len(sys.argv) > 1 else "default=;old=FA_BWD_DQ_NW=4,FA_BWD_DKDV=8;dq64=FA_BWD_DQ_NW=64,FA_BWD_DKDV=8"
settings = [(it.partition("=")[0], dict(x.split("=") for x in it.partition("=")[2].split(",") if x)) for it in spec.split(";")]
keys = sorted({k for _, e in settings for k in e})
for (B, S, H, D) in ((4, 4096, 32, 128), (4, 4096, 32, 64)):
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q); do = torch.randn_like(q)
    al = torch.tensor([2.0 ** (-8.0 * (i + 1) / H) for i in range(H)], device="cuda")
    for name, kw in (("plain", {}), ("softcap", dict(cap=30.0)), ("dropout", dict(p=0.1)), ("alibi", dict(al=al))):
        for k_ in keys: os.environ.pop(k_, None)
        be.reload_knobs()
        torch.manual_seed(1)
        out, lse, _, rng = be.fwd(q, k, v, None, kw.get("al"), kw.get("p", 0.0), D ** -0.5, True, -1, -1, kw.get("cap", 0.0), False, None)
        row = []
        for sname, env in settings:
            for k_ in keys: os.environ.pop(k_, None)
            os.environ.update(env); be.reload_knobs()
            m = t(lambda: be.bwd(do, q, k, v, out, lse, None, None, None, kw.get("al"), kw.get("p", 0.0), D ** -0.5, True, -1, -1, kw.get("cap", 0.0), False, None, rng))
            s = be.last_schedule()
            row.append(f"[{sname}] {m:.3f} ms {2.5 * 4 * B * H * S * S * D / 2 / m / 1e9:.0f} TF (dq{s['bwd_dq_nw']}/dk{s['bwd_dkdv_nw']})")
        print(f"bwd causal B={B} S={S} H={H} D={D} {name}: " + "  ".join(row), flush=True)
