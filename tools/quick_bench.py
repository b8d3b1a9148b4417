"""Quick TFLOPS sweep of the forward (and backward when built) on one GPU; prints a table."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
from flash_attn_amd import backend as be  # noqa: E402


def bench(fn, warmup=3, reps=10):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    torch.manual_seed(0)
    do_bwd = "--bwd" in sys.argv
    rows = []
    for d, H in ((128, 16), (64, 32)):
        for causal in (False, True):
            for S in (512, 1024, 2048, 4096, 8192, 16384):
                B = 16384 // S
                q = torch.randn(B, S, H, d, device="cuda", dtype=torch.bfloat16)
                k = torch.randn_like(q); v = torch.randn_like(q)
                sc = d ** -0.5
                f = lambda: be.fwd(q, k, v, None, None, 0.0, sc, causal, -1, -1, 0.0, False, None)
                ms = bench(f)
                flops = 4 * B * H * S * S * d / (2 if causal else 1)
                line = f"fwd d={d} causal={int(causal)} S={S:6d} B={B:3d}: {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TFLOPS"
                if do_bwd:
                    out, lse, _, _ = f()
                    do = torch.randn_like(out)
                    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
                    g = lambda: be.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, sc, causal, -1, -1, 0.0, False, None, None)
                    msb = bench(g)
                    line += f" | bwd {msb:8.3f} ms {2.5 * flops / msb / 1e9:8.1f} TFLOPS"
                print(line, flush=True)
    # BASELINE config 3
    B, S, H, d = 4, 4096, 32, 128
    q = torch.randn(B, S, H, d, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
    for nw in ("4", "8"):
        os.environ["FA_FWD_NW"] = nw; be.reload_knobs()
        ms = bench(lambda: be.fwd(q, k, v, None, None, 0.0, d ** -0.5, True, -1, -1, 0.0, False, None))
        print(f"cfg3 fwd NW={nw}: {ms:.3f} ms {4 * B * H * S * S * d / 2 / ms / 1e9:.1f} TFLOPS", flush=True)
    os.environ.pop("FA_FWD_NW"); be.reload_knobs()


if __name__ == "__main__":
    main()
