"""Headline benchmark: attention forward (and forward+backward) TFLOP/s on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--windows 5] [--preroll-ms 400] [--no-sweep] [--no-cpu] [--no-traffic] [--no-parity]

Workload (BASELINE.json): config 3 -- B=4 H=32 S=4096 D=128 bf16 causal, synthetic N(0,1) inputs
resident in HBM.  One "step" = one forward pass of the hot path (fa_fwd through the C ABI); the
forward+backward rate on the same config, the reference's seqlen sweep (S = 512 .. 16k at D = 128 and D = 64, causal and not), BASELINE's
configs 2 / 4 / 5 at their real shapes (`configs`) and the error table of the timed workload against PyTorch fp32 / bf16 / an fp64 sample
(`parity`, default and FA_STRICT numerics) and the opt-in fused backward next to the default one (`bwd_fused`, timed in a child process under a timeout) are reported as extra keys, `roofline.kernel` is what fa_last_schedule() says the C ABI launched, `roofline.traffic` comes from
two rocprofv3 --pmc passes run from here (outside the timed region).  FLOP convention = the reference's (benchmarks/benchmark_flash_attention.py:
27-30): fwd = 4*B*H*S^2*D (/2 causal), bwd = 2.5x, fwd+bwd = 3.5x.
Timing: an untimed clock ramp (>= --preroll-ms of the same launch), W counted warm-up launches, then --windows regions of EXACTLY K
launches each, every region bracketed by barrier + synchronize; `ms_per_step` / `value` come from the median region (max over ranks),
min / max are kept under `windows`.
N>1: independent replicas, one process per GPU (the path has no exchange step): under a launcher (WORLD_SIZE set) this
process is one rank; without one, `--gpus N` starts the N ranks itself.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "flash-attention_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def fwd_flops(B, H, S, D, causal):
    return 4.0 * B * H * S * S * D / (2.0 if causal else 1.0)


def time_kernel(fn, steps, warmup, sync):
    for _ in range(warmup):
        fn()
    sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    sync()
    wall = time.perf_counter() - t0
    return wall, e0.elapsed_time(e1) / steps  # host seconds for all steps, device ms per step


def cpu_baseline(D, S, causal, heads=8, reps=2):
    """Reference CPU SDPA (torch F.scaled_dot_product_attention, bf16) on a bounded sample of the same
    workload: `heads` (batch, head) units of config 3 (units are independent, so the rate is per-unit
    exact); all host cores.  Forward, and forward+backward through autograd (one rep)."""
    import torch.nn.functional as F
    n = os.cpu_count() or 1
    torch.set_num_threads(n)
    g = torch.Generator().manual_seed(0)
    q = torch.randn(1, heads, S, D, generator=g).bfloat16()
    k = torch.randn(1, heads, S, D, generator=g).bfloat16()
    v = torch.randn(1, heads, S, D, generator=g).bfloat16()
    F.scaled_dot_product_attention(q[:, :1], k[:, :1], v[:, :1], is_causal=causal)  # warm-up
    t0 = time.perf_counter()
    for _ in range(reps):
        F.scaled_dot_product_attention(q, k, v, is_causal=causal)
    dt = (time.perf_counter() - t0) / reps
    qg, kg, vg = (t.clone().requires_grad_() for t in (q, k, v))
    t0 = time.perf_counter()
    o = F.scaled_dot_product_attention(qg, kg, vg, is_causal=causal)
    o.backward(torch.ones_like(o))
    dt_fb = time.perf_counter() - t0
    fl = fwd_flops(1, heads, S, D, causal)
    return {"value": fl / dt / 1e12, "unit": "TFLOP/s", "cores": n, "kind": "reference",
            "fwd_bwd_value": 3.5 * fl / dt_fb / 1e12,
            "sample": f"torch CPU SDPA bf16, {heads} of 128 (batch,head) units of config 3 (S={S}, D={D}, causal): fwd {reps} reps, "
                      f"fwd+bwd (autograd) 1 rep"}


def measure_traffic(kernel_substr, timeout=240):
    """HBM bytes per launch of the dominant kernel from the PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
    WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (kernel-trace only) over `bench.py --one-launch`; FETCH_SIZE is doubled
    (gfx950 reports half the bytes of wide coalesced reads), both are KiB.  Returns (bytes, detail) or (None, reason)."""
    import shutil, sqlite3, subprocess, tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="fa_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--one-launch"]
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=timeout, check=True)
            db = None
            for root, _, files in os.walk(d):
                for f in files:
                    if f.endswith(".db"):
                        db = os.path.join(root, f)
            if db is None:
                return None, "no rocpd database written"
            c = sqlite3.connect(db)
            ccols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
            kn = "kernel_name" if "kernel_name" in ccols else "name"
            rows = c.execute(f"select {kn}, avg(value), count(*) from counters_collection where counter_name = ? group by {kn}", (ctr,)).fetchall()
            rows = [r for r in rows if kernel_substr in r[0]]
            if not rows:
                return None, f"kernel {kernel_substr} not in the {ctr} pass"
            vals[ctr] = max(rows, key=lambda r: r[2])[1]
        except Exception as e:  # profiler missing / refused / format change: report, never fail the bench
            return None, f"{ctr} pass failed: {type(e).__name__}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    nbytes = int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
    return nbytes, {"fetch_size_kib_raw": vals["FETCH_SIZE"], "fetch_correction": 2.0, "write_size_kib_raw": vals["WRITE_SIZE"]}


def dist_setup(backend=None, device=None):
    """One process per GPU (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the launcher).  Returns (rank, world)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
            dist.init_process_group(backend=backend or "nccl", **kw)
    return rank, world


def replica_aggregate(wall_seconds, units_per_rank, world, device="cpu"):
    """Replicas-only scaling (no data-path collective): whole-job rate = all ranks' units / max-over-ranks time."""
    t = torch.tensor([wall_seconds], dtype=torch.float64, device=device)
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max = float(t.item())
    return world * units_per_rank / wall_max, wall_max


def sweep(be, dev, sync, D=128):
    """The reference's headline sweep (benchmarks/benchmark_flash_attention.py:70-78): hidden dim 2048 (H = 16 at D = 128, 32 at D = 64 -- it
    sweeps both head dims), B = 16384 / S, bf16, dropout 0; TFLOP/s forward and forward+backward, non-causal and causal."""
    rows = []
    H = 2048 // D
    for causal in (False, True):
        for S in (512, 1024, 2048, 4096, 8192, 16384):
            B = 16384 // S
            q = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
            k, v = torch.randn_like(q), torch.randn_like(q)
            sc = D ** -0.5
            f = lambda: be.fwd(q, k, v, None, None, 0.0, sc, causal, -1, -1, 0.0, False, None)
            _, ms = time_kernel(f, 20, 5, sync)
            name = be.last_schedule()["name"]
            o, l, _, _ = f()
            g = torch.randn_like(o)
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            h = lambda: be.bwd(g, q, k, v, o, l, dq, dk, dv, None, 0.0, sc, causal, -1, -1, 0.0, False, None, None)
            _, mb = time_kernel(h, 8, 2, sync)
            fl = fwd_flops(B, H, S, D, causal)
            rows.append({"seqlen": S, "batch": B, "causal": causal, "fwd_tflops": round(fl / ms / 1e9, 1), "bwd_tflops": round(2.5 * fl / mb / 1e9, 1),
                         "fwd_bwd_tflops": round(3.5 * fl / (ms + mb) / 1e9, 1), "fwd_kernel": name})
    return {"head_dim": D, "heads": H, "dtype": "bf16", "rows": rows}


def compact_sweep(sw):
    """Numbers-only form of a sweep for the contract line: TFLOP/s by seqlen, non-causal / causal."""
    out = {"seqlen": sorted({r["seqlen"] for r in sw["rows"]}), "heads": sw["heads"]}
    for causal, tag in ((False, "noncausal"), (True, "causal")):
        rows = sorted((r for r in sw["rows"] if r["causal"] == causal), key=lambda r: r["seqlen"])
        out["fwd_" + tag] = [r["fwd_tflops"] for r in rows]
        out["bwd_" + tag] = [r["bwd_tflops"] for r in rows]
        out["fwd_bwd_" + tag] = [r["fwd_bwd_tflops"] for r in rows]
    return out


def write_extras(obj):
    """Full tables (sweeps with kernel names, configs, parity, fused-backward probe) -> gpurun_out/bench_extras.json; returns the path or the reason."""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "bench_extras.json")
        with open(path, "w") as f:
            json.dump(obj, f, indent=1)
        return os.path.relpath(path, ROOT)
    except OSError as e:
        return f"not written: {e}"


def fused_bwd_rows(be, dev, sync):
    """Extra key `bwd_fused`: the 5-contraction backwards (the fused launch FA_BWD_MODE=3, fa_bwd.hip fa_bwd_fused_kernel, profiles/r04_bwd_fused.txt; the chunked
    one FA_BWD_MODE=5, profiles/r06_bwd_c5.txt) next to the recomputing pair and to what the dispatch picks, timed alternately in this process: the headline shape
    and two shorter causal rows of the sweep.  Never fatal: a failure is
    reported in the key; the knob is restored whatever happens."""
    rows = []
    try:
        for (B, H, S, D, causal) in ((4, 32, 4096, 128, True), (8, 16, 2048, 128, True), (16, 16, 1024, 128, True)):
            q = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
            k, v = torch.randn_like(q), torch.randn_like(q)
            sc = D ** -0.5
            o, l = be.fwd(q, k, v, None, None, 0.0, sc, causal, -1, -1, 0.0, False, None)[:2]
            g = torch.randn_like(o)
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            h = lambda: be.bwd(g, q, k, v, o, l, dq, dk, dv, None, 0.0, sc, causal, -1, -1, 0.0, False, None, None)
            fl = 2.5 * fwd_flops(B, H, S, D, causal)
            row = {"batch": B, "heads": H, "seqlen": S, "head_dim": D, "causal": causal}
            # (round 6: "pair" = the recomputing 7-contraction pair pinned with FA_BWD_MODE=-1, "fused" = FA_BWD_MODE=3, "chunked5" = the chunked 5-contraction
            # backward FA_BWD_MODE=5 within its default 1 GiB of workspace, "default" = what the dispatch table of fa_api.cpp picks by itself)
            for mode, key in (("-1", "pair"), ("3", "fused"), ("5", "chunked5"), ("0", "default"), ("-1", "pair_again"), ("3", "fused_again")):
                os.environ["FA_BWD_MODE"] = mode
                be.reload_knobs()
                _, mb = time_kernel(h, 8, 2, sync)
                row[key + "_bwd_tflops"] = round(fl / mb / 1e9, 1)
                if mode == "3":
                    row["fused_launched"] = be.last_schedule().get("bwd_spill") == 3
                if mode in ("0", "5"):
                    row[key + "_handoff"] = be.last_schedule().get("bwd_spill")   # 0 = recomputing pair, 3 = fused launch, 5 = chunked launches
            row["workspace_gb"] = round(B * H * (S // 32) ** 2 * 2048 / 2 ** 30, 2)
            rows.append(row)
        return {"rows": rows, "note": "default = the table of fa_api.cpp bwd_fused_by_table: the fused launch at head dim 128, Sq = Sk, >= 32 units, causal 512-4096 rows within 1.25 GiB (up to 2048 rows: in chunks of batch entries), else the pair"}
    except Exception as e:  # noqa: BLE001
        return {"rows": rows, "error": repr(e)[:300]}
    finally:
        os.environ.pop("FA_BWD_MODE", None)
        be.reload_knobs()


def fused_bwd_probe(timeout=120):
    """Runs fused_bwd_rows in a child process under a timeout: an opt-in path must not be able to take the contract line down with it."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--fused-probe"], capture_output=True, text=True, timeout=timeout)
        for line in r.stdout.splitlines():
            if line.startswith("FUSED_PROBE "):
                return json.loads(line[len("FUSED_PROBE "):])
        return {"rows": [], "error": "no result (exit code %d): %s" % (r.returncode, (r.stderr or "")[-200:])}
    except Exception as e:  # noqa: BLE001  (TimeoutExpired included: subprocess.run has killed the child by then)
        return {"rows": [], "error": repr(e)[:300]}


def feature_rows(be, dev, sync, B, H, S, D):
    """Extra key `features`: the headline config with causal ALiBi (standard slopes 2^(-8(h+1)/H)) -- forward and backward on the ALiBi variants of the
    64-per-wave kernels -- and with softcap, next to the plain numbers of the main line.  Never fatal:
    any failure is reported in the key instead of taking the contract line down."""
    try:
        q = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
        k, v = torch.randn_like(q), torch.randn_like(q)
        al = torch.tensor([2.0 ** (-8.0 * (i + 1) / H) for i in range(H)], device=dev, dtype=torch.float32)
        sc = D ** -0.5
        f = lambda: be.fwd(q, k, v, None, al, 0.0, sc, True, -1, -1, 0.0, False, None)
        _, ms = time_kernel(f, 20, 5, sync)
        name = be.last_schedule()["name"]
        o, l = f()[:2]
        g = torch.randn_like(o)
        h = lambda: be.bwd(g, q, k, v, o, l, None, None, None, al, 0.0, sc, True, -1, -1, 0.0, False, None, None)
        _, mb = time_kernel(h, 8, 2, sync)
        sb = be.last_schedule()
        fl = fwd_flops(B, H, S, D, True)
        out = {"causal_alibi": {"fwd_tflops": round(fl / ms / 1e9, 1), "bwd_tflops": round(2.5 * fl / mb / 1e9, 1),
                                "fwd_bwd_tflops": round(3.5 * fl / (ms + mb) / 1e9, 1), "fwd_kernel": name,
                                "bwd_dq_waves_x_rows": sb["bwd_dq_nw"], "bwd_dkdv_waves": sb["bwd_dkdv_nw"]}}
        # ... and with softcap 30 (round 5: forward on the 64-rows-per-wave kernel's softcap variant; at head dim 128 the backward on the softcap variants of both 64-per-wave kernels)
        f = lambda: be.fwd(q, k, v, None, None, 0.0, sc, True, -1, -1, 30.0, False, None)
        _, ms = time_kernel(f, 20, 5, sync)
        name = be.last_schedule()["name"]
        o, l = f()[:2]
        h = lambda: be.bwd(g, q, k, v, o, l, None, None, None, None, 0.0, sc, True, -1, -1, 30.0, False, None, None)
        _, mb = time_kernel(h, 8, 2, sync)
        sb = be.last_schedule()
        out["causal_softcap"] = {"fwd_tflops": round(fl / ms / 1e9, 1), "bwd_tflops": round(2.5 * fl / mb / 1e9, 1),
                                 "fwd_bwd_tflops": round(3.5 * fl / (ms + mb) / 1e9, 1), "fwd_kernel": name,
                                 "bwd_dq_waves_x_rows": sb["bwd_dq_nw"], "bwd_dkdv_waves": sb["bwd_dkdv_nw"]}
        # ... and with dropout 0.1 (round 5: forward on the 64-rows-per-wave kernel's dropout variant; at head dim 128 dQ on the 64-rows-per-wave kernel's dropout variant, dK/dV on the eight-wave feature kernel)
        f = lambda: be.fwd(q, k, v, None, None, 0.1, sc, True, -1, -1, 0.0, False, None)
        _, ms = time_kernel(f, 20, 5, sync)
        name = be.last_schedule()["name"]
        o, l, _, rng = f()
        h = lambda: be.bwd(g, q, k, v, o, l, None, None, None, None, 0.1, sc, True, -1, -1, 0.0, False, None, rng)
        _, mb = time_kernel(h, 8, 2, sync)
        sb = be.last_schedule()
        out["causal_dropout"] = {"fwd_tflops": round(fl / ms / 1e9, 1), "bwd_tflops": round(2.5 * fl / mb / 1e9, 1),
                                 "fwd_bwd_tflops": round(3.5 * fl / (ms + mb) / 1e9, 1), "fwd_kernel": name,
                                 "bwd_dq_waves_x_rows": sb["bwd_dq_nw"], "bwd_dkdv_waves": sb["bwd_dkdv_nw"]}
        return out
    except Exception as e:   # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}


def _visible_pairs(S, causal, window):
    """Number of visible (query, key) pairs of an S x S problem under a causal / sliding-window mask (bottom-right aligned; here Sq = Sk)."""
    wl, wr = window
    if causal:
        wr = 0
    i = torch.arange(S, dtype=torch.int64)
    hi = torch.clamp(i + wr, max=S - 1) if wr >= 0 else torch.full_like(i, S - 1)
    lo = torch.clamp(i - wl, min=0) if wl >= 0 else torch.zeros_like(i)
    return int(torch.clamp(hi - lo + 1, min=0).sum())


def long_tail_lengths(total=65536, seed=0):
    """Long-tail sequence lengths in the spirit of benchmarks/benchmark_varlen_sched.py:76-84 (generator seed 0), trimmed to `total` tokens."""
    g = torch.Generator().manual_seed(seed)
    lens = []
    while sum(lens) < total:
        x = float(torch.rand(1, generator=g))
        lens.append(max(16, min(int(64 * (1.0 / max(x, 1e-3)) ** 0.9), 16384)))
    lens[-1] -= sum(lens) - total
    if lens[-1] <= 0:
        lens.pop()
        lens[-1] += total - sum(lens)
    return lens


def config_rows(be, dev, sync):
    """Extra key `configs`: BASELINE.json configs 2, 4 (i: 16 x 4096, ii: long-tail lengths, both 64k tokens) and 5 at their real shapes -- forward,
    backward and forward+backward TFLOP/s on the visible (query, key) pairs, and the kernels fa_last_schedule() names.  Never fatal."""
    rows = []

    def fixed(name, B, S, H, Hk, D, causal, window):
        q = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
        k = torch.randn(B, S, Hk, D, device=dev, dtype=torch.bfloat16)
        v = torch.randn_like(k)
        sc = D ** -0.5
        f = lambda: be.fwd(q, k, v, None, None, 0.0, sc, causal, window[0], window[1], 0.0, False, None)
        _, ms = time_kernel(f, 20, 5, sync)
        sched = be.last_schedule()
        o, l = f()[:2]
        g = torch.randn_like(o)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        h = lambda: be.bwd(g, q, k, v, o, l, dq, dk, dv, None, 0.0, sc, causal, window[0], window[1], 0.0, False, None, None)
        _, mb = time_kernel(h, 8, 2, sync)
        sb = be.last_schedule()
        fl = 4.0 * B * H * D * _visible_pairs(S, causal, window)
        rows.append({"config": name, "fwd_tflops": round(fl / ms / 1e9, 1), "bwd_tflops": round(2.5 * fl / mb / 1e9, 1),
                     "fwd_bwd_tflops": round(3.5 * fl / (ms + mb) / 1e9, 1), "fwd_ms": round(ms, 4), "bwd_ms": round(mb, 4),
                     "fwd_kernel": sched["name"], "bwd_dq_waves_x_rows": sb["bwd_dq_nw"], "bwd_dkdv_waves": sb["bwd_dkdv_nw"]})

    def varlen(name, lens, H, D):
        import itertools
        cu = torch.tensor([0] + list(itertools.accumulate(lens)), dtype=torch.int32, device=dev)
        tot, mx = sum(lens), max(lens)
        q = torch.randn(tot, H, D, device=dev, dtype=torch.bfloat16)
        k, v = torch.randn_like(q), torch.randn_like(q)
        sc = D ** -0.5
        f = lambda: be.varlen_fwd(q, k, v, None, cu, cu, None, None, None, None, mx, mx, 0.0, sc, False, True, -1, -1, 0.0, False, None)
        _, ms = time_kernel(f, 20, 5, sync)
        sched = be.last_schedule()
        o, l = f()[:2]
        g = torch.randn_like(o)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        h = lambda: be.varlen_bwd(g, q, k, v, o, l, dq, dk, dv, cu, cu, None, mx, mx, 0.0, sc, False, True, -1, -1, 0.0, False, None, None)
        _, mb = time_kernel(h, 8, 2, sync)
        sb = be.last_schedule()
        fl = sum(4.0 * H * D * (n * (n + 1) // 2) for n in lens)
        rows.append({"config": name, "sequences": len(lens), "total_tokens": tot, "max_seqlen": mx, "fwd_tflops": round(fl / ms / 1e9, 1),
                     "bwd_tflops": round(2.5 * fl / mb / 1e9, 1), "fwd_bwd_tflops": round(3.5 * fl / (ms + mb) / 1e9, 1), "fwd_ms": round(ms, 4),
                     "bwd_ms": round(mb, 4), "fwd_kernel": sched["name"], "work_list": bool(sched["fwd_list"]),
                     "bwd_dq_waves_x_rows": sb["bwd_dq_nw"], "bwd_dkdv_waves": sb["bwd_dkdv_nw"]})

    for fn, args in ((fixed, ("2: B=8 H=16 S=2048 D=64 non-causal", 8, 2048, 16, 16, 64, False, (-1, -1))),
                     (varlen, ("4-i: varlen 16 x 4096, H=16 D=128 causal", [4096] * 16, 16, 128)),
                     (varlen, ("4-ii: varlen long-tail lengths, 64k tokens, H=16 D=128 causal", long_tail_lengths(), 16, 128)),
                     (fixed, ("5: B=2 S=8192 H=32 / 8 KV heads D=128 causal, sliding window 1024", 2, 8192, 32, 8, 128, True, (1024, 0)))):
        try:
            fn(*args)
        except Exception as e:   # noqa: BLE001
            rows.append({"config": args[0], "error": f"{type(e).__name__}: {e}"})
    return rows


def _torch_attention(q, k, v, causal, dtype):
    """Plain PyTorch attention of ONE (batch, head) -- q, k, v (S, D) -- in `dtype` (fp32 / fp64: the reference point; bf16: the same-dtype
    baseline whose error calibrates the tolerance, tests/test_flash_attn.py:1121).  Returns out (S, D), lse (S,)."""
    q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    s = (q * (q.shape[-1] ** -0.5)) @ k.T
    if causal:
        S = q.shape[0]
        s = s.masked_fill(torch.ones(S, S, dtype=torch.bool, device=q.device).triu(1), float("-inf"))
    lse = torch.logsumexp(s.float() if dtype != torch.float64 else s, dim=-1)
    return torch.softmax(s, dim=-1).to(v.dtype) @ v, lse


def _ref_all_heads(q, k, v, do, causal, dtype):
    B, S, H, D = q.shape
    out = torch.empty(B, S, H, D, device=q.device, dtype=dtype)
    lse = torch.empty(B, H, S, device=q.device, dtype=torch.float32)
    dq, dk, dv = torch.empty_like(out), torch.empty_like(out), torch.empty_like(out)
    for b in range(B):
        for h in range(H):
            qs, ks, vs = (t[b, :, h].detach().to(dtype).requires_grad_(True) for t in (q, k, v))
            o, l = _torch_attention(qs, ks, vs, causal, dtype)
            gq, gk, gv = torch.autograd.grad(o, (qs, ks, vs), do[b, :, h].to(dtype))
            out[b, :, h], lse[b, h], dq[b, :, h], dk[b, :, h], dv[b, :, h] = o.detach(), l.detach().float(), gq, gk, gv
    return out, lse, dq, dk, dv


def parity_report(be, dev, q, k, v, causal):
    """Extra key `parity` (outside every timed region): max / mean |error| of out, LSE, dQ, dK, dV of the timed workload against plain PyTorch in fp32
    (all (batch, head) units, computed on the GPU), next to the error of the same computation done by PyTorch in bf16 -- the reference's own
    acceptance yardstick -- for the DEFAULT numerics and for FA_STRICT=1 (softmax_scale applied in fp32 to every score; the default 64-rows-per-wave
    forward rounds q*scale*log2e to bf16 once); plus one sampled (batch, head) unit in fp64, gradients included, to tie the fp32 reference itself down."""
    def err(a, b):
        d = (a.double() - b.double()).abs()
        return {"max": float(d.max()), "mean": float(d.mean())}

    def ours():
        D = q.shape[-1]
        out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None)
        name = be.last_schedule()["name"]
        dq, dk, dv, _ = be.bwd(do, q, k, v, out, lse, None, None, None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None, None)
        return (out, lse, dq, dk, dv), name

    try:
        torch.manual_seed(1234)
        do = torch.randn_like(q)
        names = ("out", "lse", "dq", "dk", "dv")
        ref = _ref_all_heads(q, k, v, do, causal, torch.float32)
        pt = _ref_all_heads(q, k, v, do, causal, torch.bfloat16)
        rep = {"reference": "PyTorch fp32 on the GPU, all (batch, head) units of the timed workload", "tensors": {}}
        got, kname = ours()
        rep["default_forward_kernel"] = kname
        os.environ["FA_STRICT"] = "1"
        be.reload_knobs()
        try:
            got_s, sname = ours()
        finally:
            os.environ.pop("FA_STRICT", None)
            be.reload_knobs()
        rep["strict_forward_kernel"] = sname
        for i, nm in enumerate(names):
            rep["tensors"][nm] = {"default": err(got[i], ref[i]), "strict": err(got_s[i], ref[i]), "pytorch_bf16": err(pt[i], ref[i])}
        b, h = q.shape[0] - 1, q.shape[2] // 2 + 1
        qs, ks, vs = (t[b, :, h].detach().double().requires_grad_(True) for t in (q, k, v))
        o64, l64 = _torch_attention(qs, ks, vs, causal, torch.float64)
        g64 = torch.autograd.grad(o64, (qs, ks, vs), do[b, :, h].double())
        r64 = (o64.detach(), l64.detach(), g64[0], g64[1], g64[2])
        sel = lambda t, i: t[b, h] if i == 1 else t[b, :, h]
        rep["fp64_sample"] = {"unit": [b, h], "tensors": {nm: {"default": err(sel(got[i], i), r64[i]), "fp32_reference": err(sel(ref[i], i), r64[i])}
                                                           for i, nm in enumerate(names)}}
        return rep
    except Exception as e:   # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}


def preroll(fn, sync, min_ms):
    """Clock ramp: the same launch, back to back, for at least `min_ms` of wall time (outside every timed region).  A count-based
    warm-up of a 0.6 ms kernel ends ~3 ms after process start, before the chip has left its idle clocks (round 2: headline 9 %
    under the same kernel's sweep row in the same process)."""
    t0 = time.perf_counter()
    sync()
    while (time.perf_counter() - t0) * 1e3 < min_ms:
        for _ in range(16):
            fn()
        sync()
    return (time.perf_counter() - t0) * 1e3


def timed_windows(fn, steps, windows, sync, world, dev, use_events=True):
    """`windows` timed regions of EXACTLY `steps` launches each, every one bracketed by barrier + synchronize on both sides (sync());
    per window: host seconds (max over ranks) and device ms per step (HIP events on the launch stream)."""
    out = []
    for _ in range(windows):
        sync()
        if use_events:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if use_events:
            e0.record()
        for _ in range(steps):
            fn()
        if use_events:
            e1.record()
        sync()
        wall = time.perf_counter() - t0
        ms_dev = e0.elapsed_time(e1) / steps if use_events else wall / steps * 1e3
        _, wall_max = replica_aggregate(wall, 0.0, world, dev)
        out.append((wall_max, ms_dev))
    return out


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawned_rank(rank, argv, world, port):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    main(argv)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--windows", type=int, default=5, help="timed windows of --steps launches; ms_per_step / value = the median window")
    ap.add_argument("--preroll-ms", type=float, default=400.0, help="clock ramp before the counted warm-up (same launch, untimed)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the seqlen sweep (extra key `sweep`)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the error table of the timed workload (extra key `parity`)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc passes behind roofline.traffic")
    ap.add_argument("--one-launch", action="store_true", help="(internal) a few forward launches of the workload, for the PMC passes")
    ap.add_argument("--fused-probe", action="store_true", help="(internal) time the opt-in fused backward next to the default one and print the rows as JSON")
    ap.add_argument("--stub-workload", action="store_true",
                    help="(tests only) CPU + gloo run of the launcher / timing / aggregation code with a sleep in place of the kernel; the line says so")
    return ap.parse_args(argv)


def main(argv=None):
    a = parse_args(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the N ranks here, one process per device (replicas only: the ranks share nothing but the barriers and
        # the max-over-ranks reduction of the window times)
        import torch.multiprocessing as mp
        mp.spawn(_spawned_rank, args=(list(sys.argv[1:] if argv is None else argv), a.gpus, _free_port()), nprocs=a.gpus, join=True)
        return

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    B, H, S, D, causal = 4, 32, 4096, 128, True
    scale = D ** -0.5
    if a.stub_workload:
        dev = torch.device("cpu")
        rank, world = dist_setup("gloo")
        be = None
        fwd = lambda: time.sleep(1e-3)
        dev_sync = lambda: None
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        from flash_attn_amd import backend as be
        if a.one_launch:
            torch.manual_seed(0)
            q = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
            k, v = torch.randn_like(q), torch.randn_like(q)
            for _ in range(3):
                be.fwd(q, k, v, None, None, 0.0, scale, causal, -1, -1, 0.0, False, None)
            torch.cuda.synchronize(dev)
            return
        if a.fused_probe:
            print("FUSED_PROBE " + json.dumps(fused_bwd_rows(be, dev, lambda: torch.cuda.synchronize(dev))), flush=True)
            return
        rank, world = dist_setup("nccl", dev)
        torch.manual_seed(rank)
        q = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
        k = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
        v = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
        fwd = lambda: be.fwd(q, k, v, None, None, 0.0, scale, causal, -1, -1, 0.0, False, None)
        dev_sync = lambda: torch.cuda.synchronize(dev)
    dist_on = world > 1
    if dist_on:
        import torch.distributed as dist

    def sync():
        dev_sync()
        if dist_on:
            dist.barrier()
            dev_sync()

    sync()
    preroll_ms = preroll(fwd, dev_sync, a.preroll_ms)
    for _ in range(a.warmup):
        fwd()
    wins = timed_windows(fwd, a.steps, max(1, a.windows), sync, world, dev, use_events=not a.stub_workload)
    wins_sorted = sorted(wins)
    wall_max, ms = wins_sorted[len(wins_sorted) // 2]      # the median window (host clock, max over ranks) and ITS device time per step
    flops = fwd_flops(B, H, S, D, causal)
    value = world * flops * a.steps / wall_max / 1e12      # whole-job aggregate over replicas, host-clocked
    sched = be.last_schedule() if be else {"name": "stub (sleep 1 ms)", "fwd_nw": 0}

    if a.stub_workload:
        if rank == 0:
            print(json.dumps({"metric": "attention_fwd_tflops", "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": world, "steps": a.steps,
                              "warmup": a.warmup, "ms_per_step": round(wall_max / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                              "vs_baseline": None, "dtype": "none", "data": "STUB: no kernel ran (launcher / aggregation self-test on CPU)",
                              "config": {"workload": "stub", "parallelism": f"replicas x{world}"}, "preroll_ms": round(preroll_ms, 1)}), flush=True)
        if dist_on:
            dist.barrier()
            dist.destroy_process_group()
        return

    # forward + backward on the same config (extra keys)
    out, lse, _, _ = fwd()
    do = torch.randn_like(out)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    bwd = lambda: be.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, scale, causal, -1, -1, 0.0, False, None, None)
    nb = max(5, a.steps // 5)
    for _ in range(3):
        bwd()
    bw = sorted(timed_windows(bwd, nb, 3, sync, world, dev))
    ms_bwd = bw[len(bw) // 2][1]

    if rank == 0:
        algo_bytes = 2 * (q.numel() + k.numel() + v.numel() + out.numel()) + 4 * lse.numel()
        res = {
            "metric": "attention_fwd_tflops", "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(wall_max / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE config 3: B=4 H=32 S=4096 D=128 bf16 causal, forward (flash_attn_func path, fa_fwd via C ABI)",
                       "batch": B, "heads": H, "seqlen": S, "head_dim": D, "causal": causal, "parallelism": f"replicas x{world}"},
            "preroll_ms": round(preroll_ms, 1),
            "windows": {"n": len(wins), "steps_each": a.steps, "reported": "median", "ms_per_step_min": round(wins_sorted[0][0] / a.steps * 1e3, 4),
                        "ms_per_step_max": round(wins_sorted[-1][0] / a.steps * 1e3, 4)},
            "roofline": {"bound": "mfma", "achieved": round(flops / ms / 1e9, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(flops / ms / 1e9 / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                         "kernel": sched["name"], "kernel_waves_per_workgroup": sched["fwd_nw"],
                         "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms": round(ms, 4)},
            "fwd_bwd": {"tflops": round(3.5 * flops / (ms + ms_bwd) / 1e9, 2), "bwd_tflops": round(2.5 * flops / ms_bwd / 1e9, 2),
                        "bwd_ms": round(ms_bwd, 4), "frac_of_peak": round(3.5 * flops / (ms + ms_bwd) / 1e9 / MFMA_BF16_PEAK_TFLOPS, 4)},
        }
        # HBM traffic of the dominant kernel, measured now (two rocprofv3 --pmc passes over `bench.py --one-launch`, outside
        # the timed region); null (with the reason) when the profiler is unavailable.
        if world == 1 and not a.no_traffic:
            nbytes, detail = measure_traffic(sched["name"].split("::")[-1].split("<")[0])
            res["roofline"]["traffic"] = nbytes
            res["roofline"]["traffic_detail"] = detail
        # Everything beyond the contract keys goes to a side file (round 4's 13.7 KB line was cut by the driver's tail); the line keeps a compact
        # summary of it (numbers only) and the file's path.
        extras = {}
        if world == 1 and not a.no_sweep:
            extras["sweep"] = sweep(be, dev, dev_sync)
            extras["sweep_d64"] = sweep(be, dev, dev_sync, 64)
            extras["configs"] = config_rows(be, dev, dev_sync)
            extras["features"] = feature_rows(be, dev, dev_sync, B, H, S, D)
            extras["bwd_fused"] = fused_bwd_probe()
            res["sweep_d128"] = compact_sweep(extras["sweep"])
            res["sweep_d64"] = compact_sweep(extras["sweep_d64"])
            res["configs"] = {r["config"].split(":")[0]: ([r.get("fwd_tflops"), r.get("bwd_tflops"), r.get("fwd_bwd_tflops")] if "error" not in r else "error")
                              for r in extras["configs"]}
            res["configs"]["columns"] = "fwd, bwd, fwd+bwd TFLOP/s on the visible pairs"
            ca = extras["features"].get("causal_alibi")
            if ca:
                res["causal_alibi"] = [ca["fwd_tflops"], ca["bwd_tflops"], ca["fwd_bwd_tflops"]]
            cs = extras["features"].get("causal_softcap")
            if cs:
                res["causal_softcap"] = [cs["fwd_tflops"], cs["bwd_tflops"], cs["fwd_bwd_tflops"]]
            cd = extras["features"].get("causal_dropout")
            if cd:
                res["causal_dropout"] = [cd["fwd_tflops"], cd["bwd_tflops"], cd["fwd_bwd_tflops"]]
        if world == 1 and not a.no_parity:
            extras["parity"] = parity_report(be, dev, q, k, v, causal)
            t = extras["parity"].get("tensors")
            if t:
                res["parity_max_abs_err"] = {nm: [round(t[nm][c]["max"], 6) for c in ("default", "strict", "pytorch_bf16")] for nm in t}
                res["parity_max_abs_err"]["columns"] = "default, FA_STRICT, PyTorch bf16 -- against PyTorch fp32, all 128 units"
        if world == 1 and not a.no_cpu:
            res["cpu_baseline"] = cpu_baseline(D, S, causal)
        if extras:
            res["extras_file"] = write_extras(dict(res, **extras))
        print(json.dumps(res), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
