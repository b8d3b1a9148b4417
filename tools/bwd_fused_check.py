"""Fused backward (FA_BWD_MODE=3: dK / dV and dQ = dS.K in one launch, dS through the Infinity Cache) against the default three-contraction dQ path:
dK / dV must be bitwise equal (same dK/dV arithmetic), dQ within the rounding of the other contraction order, every result bitwise reproducible and correct
again when the same workspace is reused with other inputs; then timings.  Usage: python tools/bwd_fused_check.py [--time-only | --check-only]"""
import os, sys, statistics
os.environ["FA_BWD_GSPLIT"] = "0"   # (cross-path bitwise comparisons hold between UNSPLIT GQA groups: tests/conftest.py _UNSPLIT_MODULES)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be

def run(mode, q, k, v, do, causal, wl=-1, wr=-1):
    # (softmax_d from the pre-pass in both, the eight-wave dK/dV kernel in both: bitwise dK / dV.  Mode 0 = the recomputing pair: round 6's table would send some of
    # these shapes to the fused launch by itself, so the reference run pins -1)
    os.environ["FA_BWD_MODE"] = str(-1 if mode == 0 else mode); os.environ["FA_BWD_FUSE_DELTA"] = "0"; os.environ["FA_BWD_DKDV"] = "8"; be.reload_knobs()
    D = q.shape[-1]
    out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, wl, wr, 0.0, False, None)
    dq, dk, dv = torch.full_like(q, float("nan")), torch.full_like(k, float("nan")), torch.full_like(v, float("nan"))
    be.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, D ** -0.5, causal, wl, wr, 0.0, False, None, None)
    torch.cuda.synchronize()
    return dq, dk, dv, dict(be.last_schedule())

def check():
    torch.manual_seed(0)
    bad = 0
    cases = []
    for D in (128, 64):
        cases += [(1, 256, 256, 2, 2, D, False), (1, 512, 512, 2, 1, D, True), (2, 1024, 1024, 4, 4, D, True), (1, 300, 333, 2, 2, D, False), (1, 300, 333, 2, 2, D, True),
                  (1, 64, 64, 1, 1, D, True), (1, 1, 500, 2, 2, D, False), (2, 2048, 2048, 4, 2, D, True), (1, 1025, 1025, 1, 1, D, True), (1, 200, 1000, 4, 1, D, True),
                  (1, 777, 1000, 3, 1, D, False), (3, 1536, 1536, 8, 8, D, True), (1, 4096, 4096, 8, 2, D, True), (1, 4096, 4096, 4, 4, D, False),
                  # many units per XCD, GQA groups as units, a ragged last round of units
                  (3, 512, 512, 32, 32, D, True), (5, 300, 333, 8, 8, D, False), (3, 768, 1024, 32, 16, D, True), (2, 1024, 1024, 32, 8, D, True), (7, 640, 640, 6, 6, D, True),
                  (16, 1024, 1024, 32, 32, D, True)]
    for dt in (torch.bfloat16, torch.float16):
        for (B, Sq, Sk, H, Hk, D, causal) in cases:
            q = torch.randn(B, Sq, H, D, device="cuda", dtype=dt); k = torch.randn(B, Sk, Hk, D, device="cuda", dtype=dt)
            v = torch.randn(B, Sk, Hk, D, device="cuda", dtype=dt); do = torch.randn(B, Sq, H, D, device="cuda", dtype=dt)
            a = run(0, q, k, v, do, causal)
            s = run(3, q, k, v, do, causal)
            s2 = run(3, q, k, v, do, causal)
            eq = [bool(torch.equal(x, y)) for x, y in zip(a[:3], s[:3])]
            rep = all(torch.equal(x, y) for x, y in zip(s[:3], s2[:3]))
            dmax = [float((x.float() - y.float()).abs().max()) for x, y in zip(a[:3], s[:3])]
            scale = max(1.0, float(a[0].float().abs().max()))
            ok = s[3]["bwd_spill"] in (0, 3) and a[3]["bwd_spill"] == 0 and eq[1] and eq[2] and rep and all(torch.isfinite(x.float()).all() for x in s[:3]) and dmax[0] <= 1e-2 * scale
            bad += not ok
            print(f"{'ok ' if ok else 'BAD'} {str(dt)[6:]} B{B} Sq{Sq} Sk{Sk} H{H}/{Hk} D{D} c{int(causal)}: bitwise dq/dk/dv {eq} reproducible {rep} maxdiff {[f'{x:.2e}' for x in dmax]} spill {s[3]['bwd_spill']}", flush=True)
    # calls that do not qualify keep working on the default path
    for (B, Sq, Sk, H, D, causal, wl, wr) in ((1, 640, 640, 2, 128, False, 300, -1), (1, 1000, 200, 2, 128, True, -1, -1)):
        q = torch.randn(B, Sq, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn(B, Sk, H, D, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k); do = torch.randn_like(q)
        a = run(0, q, k, v, do, causal, wl, wr); s = run(3, q, k, v, do, causal, wl, wr)
        ok = s[3]["bwd_spill"] == 0 and all(torch.equal(x, y) for x, y in zip(a[:3], s[:3]))
        bad += not ok
        print(f"{'ok ' if ok else 'BAD'} fallback Sq{Sq} Sk{Sk} w({wl},{wr}) c{int(causal)} spill {s[3]['bwd_spill']}", flush=True)
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

def timings():
    shapes = ((4, 4096, 32, 128, True),) if "--short" in sys.argv else ((4, 4096, 32, 128, True), (4, 4096, 32, 128, False), (2, 8192, 32, 128, True), (1, 16384, 32, 128, True), (8, 2048, 32, 128, True), (16, 1024, 32, 128, True),
              (4, 4096, 32, 64, True), (8, 2048, 32, 64, False))
    for (B, S, H, D, causal) in shapes:
        q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q); do = torch.randn_like(q)
        out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        line = f"bwd B{B} S{S} H{H} D{D} c{int(causal)}:"
        for mode in (0, 3, 0, 3):
            os.environ["FA_BWD_MODE"] = str(-1 if mode == 0 else mode); os.environ.pop("FA_BWD_FUSE_DELTA", None); os.environ.pop("FA_BWD_DKDV", None); be.reload_knobs()
            ms = t_ms(lambda: be.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None, None))
            fl = 10 * B * H * S * S * D / (2 if causal else 1)
            line += f"  [mode {mode}] {ms:.3f} ms {fl / ms / 1e9:.0f} TF (spill {be.last_schedule()['bwd_spill']})"
        print(line, flush=True)

def stats():
    """Library built with EXTRA=-DFA_FZ_STATS=1 tools/ablate_fused.sh (experiments/ablations/fa_bwd.patch): where the fused launch spends its workgroups' cycles."""
    last = {}
    orig = be._run_bwd
    def keep(a, device, varlen):
        last["ws"] = orig(a, device, varlen); return last["ws"]
    be._run_bwd = keep
    os.environ["FA_BWD_MODE"] = "3"; os.environ.pop("FA_BWD_FUSE_DELTA", None); be.reload_knobs()
    for (B, S, H, D, causal) in ((4, 4096, 32, 128, True), (8, 2048, 16, 128, True), (16, 1024, 16, 128, True), (32, 512, 16, 128, True), (16, 1024, 16, 128, False)):
        q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q); do = torch.randn_like(q)
        out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        for _ in range(3): be.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None, None)
        torch.cuda.synchronize()
        n32, np64 = (S + 31) // 32, (S + 63) // 64   # the dS workspace (rows packed in 64-key pairs: the causal triangle); the sync area sits behind it
        fz = (B * H * sum(2 * min(np64, ((i if causal else 10 ** 9) // 2) + 1) for i in range(n32)) * 2048 + 255) & ~255
        w = last["ws"][fz + 32 * 4: fz + 32 * 4 + 64].view(torch.int64).tolist()
        print(f"   error flag {int(last['ws'][fz: fz + 4].view(torch.int32)[0])}  workspace {last['ws'].numel() / 2 ** 20:.0f} MiB", flush=True)
        n = max(1, w[6])
        print(f"stats B{B} S{S} c{int(causal)}: workgroups {w[6]}  per workgroup (kcycles): dK/dV part {w[0] / n / 1e3:.1f}  pops {w[1] / n / 1e3:.1f} (max {w[7] / 1e3:.1f})  dQ items {w[2] / n / 1e3:.1f}"
              f"  | CAS retries {w[3]}  publication polls {w[4]}  items taken {w[5]}", flush=True)

if __name__ == "__main__":
    if "--stats" in sys.argv: stats(); sys.exit(0)
    bad = 0
    if "--time-only" not in sys.argv: bad = check()
    if "--check-only" not in sys.argv and bad == 0: timings()
    sys.exit(1 if bad else 0)
