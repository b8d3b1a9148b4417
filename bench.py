"""Headline benchmark: attention forward (and forward+backward) TFLOP/s on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--sweep]

Workload (BASELINE.json): config 3 -- B=4 H=32 S=4096 D=128 bf16 causal, synthetic N(0,1) inputs
resident in HBM.  One "step" = one forward pass of the hot path (fa_fwd through the C ABI); the
forward+backward rate on the same config and (with --sweep) the reference's seqlen sweep are
reported as extra keys.  FLOP convention = the reference's (benchmarks/benchmark_flash_attention.py:
27-30): fwd = 4*B*H*S^2*D (/2 causal), bwd = 2.5x, fwd+bwd = 3.5x.
N>1: independent replicas, one process per GPU (the path has no exchange step).
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
    exact); all host cores."""
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
    return {"value": fwd_flops(1, heads, S, D, causal) / dt / 1e12, "unit": "TFLOP/s", "cores": n, "kind": "reference",
            "sample": f"torch CPU SDPA bf16 fwd, {heads} of 128 (batch,head) units of config 3 (S={S}, D={D}, causal), {reps} reps"}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--sweep", action="store_true", help="also print the reference's seqlen sweep (stderr)")
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rank, world = dist_setup("nccl", dev)
    dist_on = world > 1
    if dist_on:
        import torch.distributed as dist

    from flash_attn_amd import backend as be

    B, H, S, D, causal = 4, 32, 4096, 128, True
    torch.manual_seed(rank)
    q = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
    k = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
    v = torch.randn(B, S, H, D, device=dev, dtype=torch.bfloat16)
    scale = D ** -0.5
    fwd = lambda: be.fwd(q, k, v, None, None, 0.0, scale, causal, -1, -1, 0.0, False, None)

    def sync():
        torch.cuda.synchronize(dev)
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize(dev)

    sync()
    wall, ms = time_kernel(fwd, a.steps, a.warmup, sync)
    flops = fwd_flops(B, H, S, D, causal)
    rate, wall_max = replica_aggregate(wall, flops * a.steps, world, dev)
    value = rate / 1e12  # whole-job aggregate over replicas, host-clocked

    # forward + backward on the same config (extra keys)
    out, lse, _, _ = fwd()
    do = torch.randn_like(out)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    bwd = lambda: be.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, scale, causal, -1, -1, 0.0, False, None, None)
    _, ms_bwd = time_kernel(bwd, max(5, a.steps // 5), 3, sync)

    if rank == 0:
        res = {
            "metric": "attention_fwd_tflops", "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(wall_max / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE config 3: B=4 H=32 S=4096 D=128 bf16 causal, forward (flash_attn_func path, fa_fwd via C ABI)",
                       "batch": B, "heads": H, "seqlen": S, "head_dim": D, "causal": causal, "parallelism": f"replicas x{world}"},
            "roofline": {"bound": "mfma", "achieved": round(flops / ms / 1e9, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(flops / ms / 1e9 / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                         "kernel": "fa::fa_fwd_il_kernel<bf16,128,4,3> (software-pipelined forward, 4-wave workgroups; schedule chosen in fa_api.cpp)", "algorithmic_flops_per_launch": flops, "kernel_ms": round(ms, 4)},
            "fwd_bwd": {"tflops": round(3.5 * flops / (ms + ms_bwd) / 1e9, 2), "bwd_tflops": round(2.5 * flops / ms_bwd / 1e9, 2),
                        "bwd_ms": round(ms_bwd, 4), "frac_of_peak": round(3.5 * flops / (ms + ms_bwd) / 1e9 / MFMA_BF16_PEAK_TFLOPS, 4)},
        }
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the committed
        # rocprofv3 measurement of the same kernel/config (separate --pmc passes, gfx950 FETCH_SIZE x2 correction)
        # is reported when present.
        tpath = os.path.join(ROOT, "profiles", "r01_fwd_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                res["roofline"]["traffic"] = json.load(f)["hbm_bytes_per_launch"]
        if world == 1 and not a.no_cpu:
            res["cpu_baseline"] = cpu_baseline(D, S, causal)
        if a.sweep:
            for d_, H_ in ((128, 16), (64, 32)):
                for c_ in (False, True):
                    for S_ in (512, 1024, 2048, 4096, 8192, 16384):
                        B_ = 16384 // S_
                        q_ = torch.randn(B_, S_, H_, d_, device=dev, dtype=torch.bfloat16)
                        k_, v_ = torch.randn_like(q_), torch.randn_like(q_)
                        f_ = lambda: be.fwd(q_, k_, v_, None, None, 0.0, d_ ** -0.5, c_, -1, -1, 0.0, False, None)
                        _, m_ = time_kernel(f_, 20, 5, lambda: torch.cuda.synchronize(dev))
                        o_, l_, _, _ = f_()
                        g_ = torch.randn_like(o_)
                        a_, b_, c2 = torch.empty_like(q_), torch.empty_like(k_), torch.empty_like(v_)
                        h_ = lambda: be.bwd(g_, q_, k_, v_, o_, l_, a_, b_, c2, None, 0.0, d_ ** -0.5, c_, -1, -1, 0.0, False, None, None)
                        _, mb_ = time_kernel(h_, 10, 3, lambda: torch.cuda.synchronize(dev))
                        fl = fwd_flops(B_, H_, S_, d_, c_)
                        print(f"sweep d={d_} causal={int(c_)} S={S_:6d} B={B_:3d}: fwd {fl / m_ / 1e9:7.1f} TF  bwd {2.5 * fl / mb_ / 1e9:7.1f} TF  "
                              f"fwd+bwd {3.5 * fl / (m_ + mb_) / 1e9:7.1f} TF", file=sys.stderr, flush=True)
        print(json.dumps(res), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
