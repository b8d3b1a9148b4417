"""Random shapes inside the region where round 6's dispatch table gives the backward to the fused launch BY DEFAULT (head dim 128, Sq = Sk, >= 32 (batch, kv head) units, causal 512 .. 2048 rows or
from 256 rows on large grids with or without a mask, dS workspace <= 1.25 GiB, larger batches in chunks): the default against the recomputing pair pinned onto the same dK/dV kernel text (dK / dV bitwise, dQ within rounding and within the
reference's rule against fp32), twice (bitwise), through the autograd interface as well.  usage: python tools/bwd_table_stress.py [cases] [seed]"""
import os, sys, random
os.environ["FA_BWD_GSPLIT"] = "0"   # (cross-path bitwise comparisons hold between UNSPLIT GQA groups: tests/conftest.py _UNSPLIT_MODULES)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be
from tests.test_bwd_schedules_gpu import ref_grads, _plan_of

def run(env, q, k, v, do, causal=True):
    for kk in ("FA_BWD_MODE", "FA_BWD_FUSE_DELTA", "FA_BWD_DKDV"): os.environ.pop(kk, None)
    os.environ.update(env); be.reload_knobs()
    D = q.shape[-1]
    out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None)
    dq, dk, dv = torch.full_like(q, float("nan")), torch.full_like(k, float("nan")), torch.full_like(v, float("nan"))
    be.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None, None)
    torch.cuda.synchronize()
    return dq, dk, dv, dict(be.last_schedule())

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
os.environ["FA_DEBUG_POISON_WS"] = "1"
bad = 0
for i in range(n):
    causal = rng.random() < 0.7
    S = rng.choice([512, 1024, 2048, 4096, rng.randint(512, 4096), rng.randint(512, 2048), rng.randint(256, 511)]) if causal else rng.choice([512, 1024, 1536, rng.randint(512, 1536), rng.randint(256, 511)])
    Hk = rng.choice([1, 2, 3, 4, 8]); g = rng.choice([1, 1, 2, 4]); H = Hk * g
    Bmax = max(1, (1 << 30) // (H * ((S + 31) // 32) ** 2 * 2048))
    if -(-32 // Hk) > Bmax:   # (32 units of this shape do not fit 1 GiB: outside the table's region -- draw again)
        S = rng.choice([512, 1024]); Bmax = max(1, (1 << 30) // (H * ((S + 31) // 32) ** 2 * 2048))
    B = rng.randint(max(1, -(-32 // Hk)), max(-(-32 // Hk), min(40, Bmax)))
    if S < 512: B = max(B, -(-(196608 if causal else 786432) // (S * Hk)))   # (below 512 rows the table asks for a large grid: units x rows)
    if S <= 2048 and rng.random() < 0.3:   # (late round 6: a batch over the 1.25 GiB bound -> chunks of whole batch entries, one launch each)
        B = rng.randint(Bmax + Bmax // 4 + 1, 3 * Bmax)
    dt = rng.choice([torch.bfloat16, torch.float16])
    torch.manual_seed(i)
    q = torch.randn(B, S, H, 128, device="cuda", dtype=dt); k = torch.randn(B, S, Hk, 128, device="cuda", dtype=dt); v = torch.randn_like(k); do = torch.randn_like(q)
    d = run({}, q, k, v, do, causal); d2 = run({}, q, k, v, do, causal)
    plan = _plan_of(B, S, H, Hk, 128, causal)   # (under the default knobs run() has just restored)
    p = run({"FA_BWD_MODE": "-1", "FA_BWD_FUSE_DELTA": "0", "FA_BWD_DKDV": "8"}, q, k, v, do, causal)
    ok = d[3]["bwd_spill"] == plan[0] and p[3]["bwd_spill"] == 0 and all(torch.equal(a, b) for a, b in zip(d[:3], d2[:3])) and torch.equal(d[1], p[1]) and torch.equal(d[2], p[2])
    ok = ok and all(bool(torch.isfinite(x.float()).all()) for x in d[:3])
    if plan[0] != 3: ok = d[3]["bwd_spill"] == 0 and all(torch.equal(a, b) for a, b in zip(d[:3], d2[:3]))   # (outside the table after all, e.g. too few units per chunk: the pair, on its own kernels)
    e = ""
    if B * H * S * S <= 2 ** 27:
        r = ref_grads(q, k, v, do, causal, -1, -1); pt = ref_grads(q, k, v, do, causal, -1, -1, upcast=False)
        ed, ep, et = (float((x.float() - r[0]).abs().max()) for x in (d[0], p[0], pt[0]))
        ok = ok and ed <= 3 * et + 1e-5 and ed <= 2 * ep + 1e-5
        e = f" dq err default {ed:.2e} pair {ep:.2e} torch-in-dtype {et:.2e}"
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} {str(dt)[6:]} B{B} S{S} H{H}/{Hk} c{int(causal)}: default spill {d[3]['bwd_spill']} launches {plan[1]}{e}", flush=True)
print("FAILURES", bad)
sys.exit(1 if bad else 0)
