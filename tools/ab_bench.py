"""A/B timing of kernel variants in ONE process, interleaved rounds (median of rounds)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be


def t_ms(fn, reps=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    variants = sys.argv[1].split(",") if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else ["4", "8", "16"]
    if "--sweep-shapes" in sys.argv:
        shapes_override = [(16384 // S, S, 16, 128, c) for c in (False, True) for S in (512, 1024, 2048, 4096, 8192)] + [(8, 2048, 32, 128, True), (4, 4096, 32, 128, True), (2, 8192, 32, 128, True)]
    elif "--d64-shapes" in sys.argv:
        shapes_override = [(8, 2048, 16, 64, False), (32, 512, 16, 64, False), (16, 1024, 16, 64, False), (4, 4096, 16, 64, False), (2, 8192, 16, 64, False),
                           (1, 16384, 16, 64, False), (16, 1024, 16, 64, True), (8, 2048, 16, 64, True), (4, 4096, 32, 64, True), (2, 8192, 32, 64, True), (1, 16384, 16, 64, True)]
    else:
        shapes_override = None
    shapes = [(4, 4096, 32, 128, True), (4, 4096, 32, 128, False), (1, 16384, 16, 128, True), (1, 16384, 16, 128, False),
              (8, 2048, 16, 64, False), (4, 4096, 32, 64, True), (16, 1024, 16, 128, True), (2, 8192, 32, 128, True)]
    if shapes_override: shapes = shapes_override
    torch.manual_seed(0)
    for (B, S, H, D, causal) in shapes:
        q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
        fl = 4 * B * H * S * S * D / (2 if causal else 1)
        res = {x: [] for x in variants}
        f = lambda: be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None)
        def setv(x):
            parts = x.split(":") + ["", "", ""]
            if parts[3]:
                os.environ["FA_IL_LDS_PAD"] = parts[3]
            else:
                os.environ.pop("FA_IL_LDS_PAD", None)
            os.environ["FA_FWD_NW"] = parts[0]
            os.environ["FA_RESCALE_THR"] = parts[1] or "0"
            if parts[2]:
                os.environ["FA_IL_SCHED"] = parts[2]
            else:
                os.environ.pop("FA_IL_SCHED", None)
            be.reload_knobs()
        for x in variants:
            setv(x); f()
        for _ in range(5):
            for x in variants:
                setv(x)
                res[x].append(t_ms(f))
        line = f"fwd B={B} S={S} H={H} D={D} causal={int(causal)}: " + "  ".join(
            f"[{x}] {statistics.median(res[x]):.3f} ms {fl / statistics.median(res[x]) / 1e9:7.1f} TF" for x in variants)
        print(line, flush=True)
    os.environ.pop("FA_FWD_NW", None); os.environ.pop("FA_RESCALE_THR", None); be.reload_knobs()
    if "--bwd" in sys.argv:
        for (B, S, H, D, causal) in shapes:
            q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
            out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None)
            do = torch.randn_like(out); dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            g = lambda: be.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None, None)
            g()
            ms = statistics.median([t_ms(g, 5) for _ in range(5)])
            fl = 2.5 * 4 * B * H * S * S * D / (2 if causal else 1)
            print(f"bwd B={B} S={S} H={H} D={D} causal={int(causal)}: {ms:.3f} ms {fl / ms / 1e9:7.1f} TF", flush=True)


if __name__ == "__main__":
    main()
