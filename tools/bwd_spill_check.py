"""5-contraction backward (dS spilled by the dK/dV kernel, FA_BWD_MODE=2) against the recomputing dQ kernels (FA_BWD_MODE=1)
and an fp32 reference; then timings of both on the headline shapes."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be

def run(mode, q, k, v, do, causal, wl, wr, softcap=0.0, alibi=None, p_drop=0.0, nw="4"):
    os.environ["FA_BWD_MODE"] = str(mode); os.environ["FA_BWD_DQ_NW"] = nw; be.reload_knobs()
    D = q.shape[-1]
    torch.cuda.manual_seed(7)
    out, lse, _, rng = be.fwd(q, k, v, None, alibi, p_drop, D ** -0.5, causal, wl, wr, softcap, False, None)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    be.bwd(do, q, k, v, out, lse, dq, dk, dv, alibi, p_drop, D ** -0.5, causal, wl, wr, softcap, False, None, rng)
    return dq, dk, dv, be.last_schedule()

torch.manual_seed(0)
bad = 0
cases = []
for D in (128, 64):
    cases += [(1, 256, 256, 2, 2, D, False, -1, -1), (1, 512, 512, 2, 1, D, True, -1, -1), (2, 1024, 1024, 4, 4, D, True, -1, -1),
              (1, 300, 333, 2, 2, D, False, -1, -1), (1, 300, 333, 2, 2, D, True, -1, -1), (1, 777, 1000, 3, 1, D, False, 100, 50),
              (1, 64, 64, 1, 1, D, True, -1, -1), (1, 1, 500, 2, 2, D, False, -1, -1), (2, 2048, 2048, 4, 2, D, True, -1, -1),
              (1, 1000, 200, 2, 2, D, True, -1, -1), (1, 513, 1025, 2, 2, D, False, 64, 0), (1, 33, 97, 2, 2, D, False, -1, -1),
              (1, 1025, 1025, 1, 1, D, True, -1, -1), (1, 200, 1000, 4, 1, D, True, -1, -1), (1, 2000, 2000, 1, 1, D, False, 0, 0),
              (1, 640, 640, 1, 1, D, False, 300, -1), (1, 640, 640, 1, 1, D, False, -1, 300)]
for dt in (torch.bfloat16, torch.float16):
    for ci, (B, Sq, Sk, H, Hk, D, causal, wl, wr) in enumerate(cases):
        q = torch.randn(B, Sq, H, D, device="cuda", dtype=dt); k = torch.randn(B, Sk, Hk, D, device="cuda", dtype=dt)
        v = torch.randn(B, Sk, Hk, D, device="cuda", dtype=dt); do = torch.randn(B, Sq, H, D, device="cuda", dtype=dt)
        feats = [dict()]
        if ci % 4 == 0: feats.append(dict(softcap=20.0))
        if ci % 4 == 1: feats.append(dict(alibi=torch.rand(B, H, device="cuda") * 0.3))
        if ci % 4 == 2: feats.append(dict(p_drop=0.2))
        for ft in feats:
            a = run(1, q, k, v, do, causal, wl, wr, **ft)
            s = run(2, q, k, v, do, causal, wl, wr, **ft)
            eq = [bool(torch.equal(x, y)) for x, y in zip(a[:3], s[:3])]
            dmax = [float((x.float() - y.float()).abs().max()) for x, y in zip(a[:3], s[:3])]
            ok = s[3]["bwd_spill"] == 1 and a[3]["bwd_spill"] == 0 and all(torch.isfinite(x.float()).all() for x in s[:3]) and max(dmax) <= 1e-2 * max(1.0, float(a[0].float().abs().max()))
            bad += not ok
            print(f"{'ok ' if ok else 'BAD'} {str(dt)[6:]} B{B} Sq{Sq} Sk{Sk} H{H}/{Hk} D{D} c{int(causal)} w({wl},{wr}) {list(ft)}: bitwise dq/dk/dv {eq} maxdiff {dmax}", flush=True)
print("FAILURES", bad)

def t_ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(5):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return statistics.median(ts)
for (B, S, H, D, causal) in ((4, 4096, 32, 128, True), (4, 4096, 32, 128, False), (1, 16384, 16, 128, True), (8, 2048, 16, 128, True), (16, 1024, 16, 128, True),
                             (32, 512, 16, 128, True), (4, 4096, 32, 64, True)):
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q); do = torch.randn_like(q)
    out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    for mode, nw in ((1, "4"), (1, "64"), (2, "4")):
        os.environ["FA_BWD_MODE"] = str(mode); os.environ["FA_BWD_DQ_NW"] = nw; be.reload_knobs()
        ms = t_ms(lambda: be.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None, None))
        fl = 10 * B * H * S * S * D / (2 if causal else 1)
        print(f"bwd B{B} S{S} H{H} D{D} c{int(causal)} mode={mode} dq_nw={nw}: {ms:.3f} ms {fl / ms / 1e9:.0f} TF", flush=True)
