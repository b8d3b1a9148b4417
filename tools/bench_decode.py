"""Decode (KV-cache, one query token) bandwidth: bytes of K and V actually attended / time, vs the HBM roofline.
usage: bench_decode.py [num_splits ...]   (0 = heuristic, 1 = unsplit)"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be

HBM_PEAK = 8000.0  # GB/s


def t_ms(fn, reps=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    splits = [int(x) for x in sys.argv[1:]] or [1, 0]
    H, Hk, D = 32, 8, 128
    for B, S in ((1, 8192), (1, 32768), (1, 131072), (8, 8192), (8, 32768), (64, 4096), (64, 16384), (256, 4096)):
        q = torch.randn(B, 1, H, D, device="cuda", dtype=torch.bfloat16)
        kc = torch.randn(B, S, Hk, D, device="cuda", dtype=torch.bfloat16)
        vc = torch.randn_like(kc)
        lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
        gb = 2 * B * S * Hk * D * 2 / 1e9
        line = f"decode B={B:3d} Sk={S:6d} H={H}/{Hk} D={D} ({gb * 1e3:7.1f} MB of K,V):"
        for ns in splits:
            f = lambda: be.fwd_kvcache(q, kc, vc, None, None, lens, None, None, None, None, None, None, None, D ** -0.5, False, -1, -1, 0.0, True, ns)
            f()
            ms = statistics.median([t_ms(f) for _ in range(5)])
            line += f"  [splits={ns}] {ms * 1e3:8.1f} us {gb / ms * 1e3:7.0f} GB/s ({gb / ms * 1e3 / HBM_PEAK * 100:4.1f}% of HBM peak)"
        print(line, flush=True)


if __name__ == "__main__":
    main()
