"""dQ of the 64-rows-per-wave backward kernel (FA_BWD_DQ_NW=64) against the 32-rows-per-wave kernel and an fp32 reference."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be

def run(nw, q, k, v, do, causal, wl=-1, wr=-1):
    os.environ["FA_BWD_DQ_NW"] = str(nw); be.reload_knobs()
    D = q.shape[-1]
    out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, wl, wr, 0.0, False, None)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    be.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, D ** -0.5, causal, wl, wr, 0.0, False, None, None)
    return dq, dk, dv, be.last_schedule()

def ref_dq(q, k, v, do, causal, wl, wr):
    qf, kf, vf = [x.float().transpose(1, 2).detach().requires_grad_(True) for x in (q, k, v)]
    Hq, Hk = qf.shape[1], kf.shape[1]
    kk = kf.repeat_interleave(Hq // Hk, 1); vv = vf.repeat_interleave(Hq // Hk, 1)
    s = qf @ kk.transpose(-1, -2) * q.shape[-1] ** -0.5
    Sq, Sk = s.shape[-2:]
    i = torch.arange(Sq, device=q.device)[:, None] + (Sk - Sq); j = torch.arange(Sk, device=q.device)[None]
    m = torch.zeros(Sq, Sk, dtype=torch.bool, device=q.device)
    if causal: wr = 0
    if wr >= 0: m |= j > i + wr
    if wl >= 0: m |= j < i - wl
    s = s.masked_fill(m, float("-inf"))
    p = torch.softmax(s, -1).nan_to_num(0.0)
    o = p @ vv
    o.backward(do.float().transpose(1, 2))
    return qf.grad.transpose(1, 2)

torch.manual_seed(0)
bad = 0
cases = [(1, 256, 256, 2, 2, 128, False, -1, -1), (1, 512, 512, 2, 1, 128, True, -1, -1), (2, 1024, 1024, 4, 4, 128, True, -1, -1),
         (1, 300, 333, 2, 2, 128, False, -1, -1), (1, 300, 333, 2, 2, 128, True, -1, -1), (1, 777, 1000, 3, 1, 128, False, 100, 50),
         (1, 64, 64, 1, 1, 128, True, -1, -1), (1, 1, 500, 2, 2, 128, False, -1, -1), (2, 2048, 2048, 4, 2, 128, True, -1, -1),
         (1, 1000, 200, 2, 2, 128, True, -1, -1), (1, 513, 1025, 2, 2, 128, False, 64, 0)]
for dt in (torch.bfloat16, torch.float16):
    for (B, Sq, Sk, H, Hk, D, causal, wl, wr) in cases:
        q = torch.randn(B, Sq, H, D, device="cuda", dtype=dt); k = torch.randn(B, Sk, Hk, D, device="cuda", dtype=dt)
        v = torch.randn(B, Sk, Hk, D, device="cuda", dtype=dt); do = torch.randn(B, Sq, H, D, device="cuda", dtype=dt)
        d64, _, _, s64 = run(64, q, k, v, do, causal, wl, wr)
        d4, _, _, s4 = run(4, q, k, v, do, causal, wl, wr)
        r = ref_dq(q, k, v, do, causal, wl, wr)
        e64 = float((d64.float() - r).abs().max()); e4 = float((d4.float() - r).abs().max())
        x = float((d64.float() - d4.float()).abs().max())
        ok = e64 <= max(2 * e4, 1e-2) and s64["bwd_dq_nw"] == 64 and torch.isfinite(d64.float()).all()
        bad += not ok
        print(f"{'ok ' if ok else 'BAD'} {str(dt)[6:]} B{B} Sq{Sq} Sk{Sk} H{H}/{Hk} c{int(causal)} w({wl},{wr}): err64 {e64:.4f} err4 {e4:.4f} diff {x:.4f} nw {s64['bwd_dq_nw']}", flush=True)
print("FAILURES", bad)

def t_ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(5):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return statistics.median(ts)
for (B, S, H, causal) in ((4, 4096, 32, True), (4, 4096, 32, False), (1, 16384, 16, True)):
    q = torch.randn(B, S, H, 128, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q); do = torch.randn_like(q)
    out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, 128 ** -0.5, causal, -1, -1, 0.0, False, None)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    for nw in (4, 8, 64):
        os.environ["FA_BWD_DQ_NW"] = str(nw); be.reload_knobs()
        ms = t_ms(lambda: be.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, 128 ** -0.5, causal, -1, -1, 0.0, False, None, None))
        fl = 10 * B * H * S * S * 128 / (2 if causal else 1)
        print(f"bwd B{B} S{S} H{H} c{int(causal)} dq_nw={nw}: {ms:.3f} ms {fl / ms / 1e9:.0f} TF", flush=True)
