"""5-contraction backward (FA_BWD_MODE=5: the 64-keys-per-wave dK/dV items hand dS over through a bounded two-slot workspace, dQ = dS.K in the next launch's
tail) against the recomputing pair: dK / dV must be bitwise equal (same kernel text, softmax_d from the pre-pass in both), dQ within the rounding of the other
contraction order and against an fp32 PyTorch backward, everything bitwise reproducible, with the workspace poisoned (every word a NaN) and the chunk size forced
small (FA_BWD_C5_CAP_MB) so that several launches alternate the slots; then timings.  Usage: python tools/bwd_c5_check.py [--time-only | --check-only | --short]"""
import os, sys, statistics
os.environ["FA_BWD_GSPLIT"] = "0"   # (cross-path bitwise comparisons hold between UNSPLIT GQA groups: tests/conftest.py _UNSPLIT_MODULES)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be

def run(mode, q, k, v, do, causal, wl=-1, wr=-1, cap=None):
    os.environ["FA_BWD_MODE"] = str(mode); os.environ["FA_BWD_FUSE_DELTA"] = "0"; os.environ["FA_BWD_DKDV"] = "64"
    if cap: os.environ["FA_BWD_C5_CAP_MB"] = str(cap)
    else: os.environ.pop("FA_BWD_C5_CAP_MB", None)
    be.reload_knobs()
    D = q.shape[-1]
    out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, wl, wr, 0.0, False, None)
    dq, dk, dv = torch.full_like(q, float("nan")), torch.full_like(k, float("nan")), torch.full_like(v, float("nan"))
    be.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, D ** -0.5, causal, wl, wr, 0.0, False, None, None)
    torch.cuda.synchronize()
    return dq, dk, dv, dict(be.last_schedule())

def ref_fp32(q, k, v, do, causal, wr):
    qf, kf, vf = (x.float().requires_grad_(True) for x in (q, k, v))
    B, Sq, H, D = q.shape; Sk, Hk = k.shape[1], k.shape[2]
    kk = kf.repeat_interleave(H // Hk, dim=2); vv = vf.repeat_interleave(H // Hk, dim=2)
    s = torch.einsum("bqhd,bkhd->bhqk", qf, kk) * D ** -0.5
    if causal or wr >= 0:
        i = torch.arange(Sq, device=q.device)[:, None]; j = torch.arange(Sk, device=q.device)[None, :]
        s = s.masked_fill(j > i + (Sk - Sq) + (0 if causal else wr), float("-inf"))
    o = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), vv)
    o.backward(do.float())
    return qf.grad, kf.grad, vf.grad

def check(short):
    torch.manual_seed(0)
    os.environ["FA_DEBUG_POISON_WS"] = "1"
    bad = 0
    cases = []
    for D in (128, 64):
        cases += [(1, 256, 256, 2, 2, D, False, -1, 0), (1, 512, 512, 2, 1, D, True, -1, 0), (2, 1024, 1024, 4, 4, D, True, -1, 0), (1, 300, 333, 2, 2, D, False, -1, 0), (1, 300, 333, 2, 2, D, True, -1, 0),
                  (1, 64, 64, 1, 1, D, True, -1, 0), (1, 1, 500, 2, 2, D, False, -1, 0), (2, 2048, 2048, 4, 2, D, True, -1, 0), (1, 1025, 1025, 1, 1, D, True, -1, 0), (1, 200, 1000, 4, 1, D, True, -1, 0),
                  (1, 777, 1000, 3, 1, D, False, -1, 0), (1, 640, 900, 2, 2, D, False, 100, 0), (1, 640, 640, 2, 2, D, False, 37, 0)]
        if not short:
            cases += [(3, 1536, 1536, 8, 8, D, True, -1, 0), (1, 4096, 4096, 8, 2, D, True, -1, 0), (1, 4096, 4096, 4, 4, D, False, -1, 0),
                      # several chunks (small cap): many units per XCD, GQA groups as units, a ragged last round of units
                      (3, 512, 512, 32, 32, D, True, -1, 16), (5, 300, 333, 8, 8, D, False, -1, 16), (3, 768, 1024, 32, 16, D, True, -1, 16), (2, 1024, 1024, 32, 8, D, True, -1, 64), (7, 640, 640, 6, 6, D, True, -1, 16),
                      (16, 1024, 1024, 32, 32, D, True, -1, 64)]
    for dt in (torch.bfloat16, torch.float16):
        for (B, Sq, Sk, H, Hk, D, causal, wr, cap) in cases:
            q = torch.randn(B, Sq, H, D, device="cuda", dtype=dt); k = torch.randn(B, Sk, Hk, D, device="cuda", dtype=dt)
            v = torch.randn(B, Sk, Hk, D, device="cuda", dtype=dt); do = torch.randn(B, Sq, H, D, device="cuda", dtype=dt)
            a = run(-1, q, k, v, do, causal, -1, wr)
            s = run(5, q, k, v, do, causal, -1, wr, cap)
            s2 = run(5, q, k, v, do, causal, -1, wr, cap)
            eq = [bool(torch.equal(x, y)) for x, y in zip(a[:3], s[:3])]
            rep = all(torch.equal(x, y) for x, y in zip(s[:3], s2[:3]))
            dmax = [float((x.float() - y.float()).abs().max()) for x, y in zip(a[:3], s[:3])]
            r = ref_fp32(q, k, v, do, causal, wr) if B * H * Sq * Sk <= 2 ** 27 else None
            e5 = float((s[0].float() - r[0]).abs().max()) if r else float("nan")
            e7 = float((a[0].float() - r[0]).abs().max()) if r else float("nan")
            fin = all(bool(torch.isfinite(x.float()).all()) for x in s[:3])
            ok = s[3]["bwd_spill"] == 5 and a[3]["bwd_spill"] == 0 and eq[1] and eq[2] and rep and fin and (r is None or e5 <= max(2 * e7, 2e-3))
            bad += not ok
            print(f"{'ok ' if ok else 'BAD'} {str(dt)[6:]} B{B} Sq{Sq} Sk{Sk} H{H}/{Hk} D{D} c{int(causal)} wr{wr} cap{cap}: bitwise dq/dk/dv {eq} reproducible {rep} finite {fin} maxdiff {[f'{x:.2e}' for x in dmax]} "
                  f"dq err vs fp32: c5 {e5:.2e} pair {e7:.2e} spill {s[3]['bwd_spill']}", flush=True)
    # calls that do not qualify keep working on the default path
    for (B, Sq, Sk, H, D, causal, wl, wr) in ((1, 640, 640, 2, 128, False, 300, -1), (1, 1000, 200, 2, 128, True, -1, -1)):
        q = torch.randn(B, Sq, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn(B, Sk, H, D, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k); do = torch.randn_like(q)
        a = run(-1, q, k, v, do, causal, wl, wr); s = run(5, q, k, v, do, causal, wl, wr)
        ok = s[3]["bwd_spill"] == 0 and all(torch.equal(x, y) for x, y in zip(a[:3], s[:3]))
        bad += not ok
        print(f"{'ok ' if ok else 'BAD'} fallback Sq{Sq} Sk{Sk} w({wl},{wr}) c{int(causal)} spill {s[3]['bwd_spill']}", flush=True)
    os.environ.pop("FA_DEBUG_POISON_WS", None)
    print("FAILURES", bad, flush=True)
    return bad

def t_ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(5):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return statistics.median(ts)

def timings(short):
    shapes = ((4, 4096, 32, 32, 128, True),) if short else ((4, 4096, 32, 32, 128, True), (4, 4096, 32, 32, 128, False), (2, 8192, 32, 32, 128, True), (1, 16384, 32, 32, 128, True), (8, 2048, 32, 32, 128, True), (16, 1024, 32, 32, 128, True),
              (32, 512, 32, 32, 128, True), (8, 2048, 16, 16, 128, False), (2, 8192, 32, 8, 128, True), (4, 4096, 32, 32, 64, True), (8, 2048, 32, 32, 64, False), (8, 2048, 16, 16, 64, False))
    caps = (1024, 2048, 4096) if not short else (1024, 4096)
    for (B, S, H, Hk, D, causal) in shapes:
        q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn(B, S, Hk, D, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k); do = torch.randn_like(q)
        out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        line = f"bwd B{B} S{S} H{H}/{Hk} D{D} c{int(causal)}:"
        fl = 10 * B * H * S * S * D / (2 if causal else 1)
        for mode, cap in ((-1, None),) + tuple((5, c) for c in caps) + ((-1, None),):
            os.environ["FA_BWD_MODE"] = str(mode); os.environ.pop("FA_BWD_FUSE_DELTA", None); os.environ.pop("FA_BWD_DKDV", None)
            if cap: os.environ["FA_BWD_C5_CAP_MB"] = str(cap)
            be.reload_knobs()
            ms = t_ms(lambda: be.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None, None))
            line += f"  [mode {mode}{'/' + str(cap) if cap else ''}] {ms:.3f} ms {fl / ms / 1e9:.0f} TF (spill {be.last_schedule()['bwd_spill']})"
        print(line, flush=True)

if __name__ == "__main__":
    short = "--short" in sys.argv
    bad = 0
    if "--time-only" not in sys.argv: bad = check(short)
    if "--check-only" not in sys.argv and bad == 0: timings(short)
    sys.exit(1 if bad else 0)
