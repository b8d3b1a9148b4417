"""Short query chunks against a KV cache with grouped heads: query heads packed into the rows of a block (default) vs one block per
query head (FA_PACK_GQA=0).  Whole fwd_kvcache calls (binder included), HIP-event timed; GB/s = K/V bytes of the batch / time."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
from flash_attn_amd import backend as be  # noqa: E402


def bench(fn, warmup=5, reps=30):
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
    d = 128
    for B, H, Hk, Sk in ((64, 32, 4, 8192), (8, 32, 8, 32768), (1, 64, 8, 131072)):
        kc = torch.randn(B, Sk, Hk, d, device="cuda", dtype=torch.bfloat16)
        vc = torch.randn_like(kc)
        lens = torch.full((B,), Sk, dtype=torch.int32, device="cuda")
        kv_bytes = 2 * B * Sk * Hk * d * 2
        for sq in (2, 4, 8, 16):
            if sq * (H // Hk) > 128:
                continue
            q = torch.randn(B, sq, H, d, device="cuda", dtype=torch.bfloat16)
            f = lambda: be.fwd_kvcache(q, kc, vc, None, None, lens, None, None, None, None, None, None, None, d ** -0.5, True, -1, -1, 0.0, True, 0)
            res = {}
            for pack in (1, 0):
                os.environ["FA_PACK_GQA"] = str(pack)
                be.reload_knobs()
                ms = bench(f)
                s = be.last_schedule()
                res[pack] = (ms, s["fwd_pack"], s["fwd_splits"])
            print(f"B={B:3d} H={H} Hk={Hk} Sk={Sk:6d} Sq={sq:2d}: packed {res[1][0] * 1e3:8.1f} us ({kv_bytes / res[1][0] / 1e6:6.0f} GB/s, g={res[1][1]}, splits {res[1][2]}) | "
                  f"per-head {res[0][0] * 1e3:8.1f} us ({kv_bytes / res[0][0] / 1e6:6.0f} GB/s, splits {res[0][2]})", flush=True)
    os.environ.pop("FA_PACK_GQA", None)
    be.reload_knobs()


if __name__ == "__main__":
    main()
