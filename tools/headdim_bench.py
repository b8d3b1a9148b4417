"""Head dims 32 / 96 / 192: the trimmed kernels on the tensors as given, against what round 1 did for them -- zero-padded copies
of q/k/v (and dO/O; padded gradients copied back) run through the 64 / 128 / 256 kernels.  Whole-call times (HIP events around the
binder call, copies included for the padded path), TFLOP/s counted on the true head dim.  Usage: python tools/headdim_bench.py"""
import os
import sys

import torch
import torch.nn.functional as F

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
    for d, dp in ((32, 64), (96, 128), (192, 256)):
        H = 2048 // d if d != 192 else 8
        for causal in (False, True):
            for S in (1024, 4096, 16384):
                B = max(1, 16384 // S)
                q = torch.randn(B, S, H, d, device="cuda", dtype=torch.bfloat16)
                k = torch.randn_like(q); v = torch.randn_like(q); do = torch.randn_like(q)
                sc = d ** -0.5
                flops = 4 * B * H * S * S * d / (2 if causal else 1)

                def fwd_native():
                    return be.fwd(q, k, v, None, None, 0.0, sc, causal, -1, -1, 0.0, False, None)

                def fwd_padded():
                    qp, kp, vp = (F.pad(t, (0, dp - d)) for t in (q, k, v))
                    o, lse, _, _ = be.fwd(qp, kp, vp, None, None, 0.0, sc, causal, -1, -1, 0.0, False, None)
                    return o[..., :d].contiguous(), lse

                out, lse, _, _ = fwd_native()
                name = be.last_schedule()["name"]

                def bwd_native():
                    return be.bwd(do, q, k, v, out, lse, None, None, None, None, 0.0, sc, causal, -1, -1, 0.0, False, None, None)

                def bwd_padded():
                    qp, kp, vp, dop, op = (F.pad(t, (0, dp - d)) for t in (q, k, v, do, out))
                    g = be.bwd(dop, qp, kp, vp, op, lse, None, None, None, None, 0.0, sc, causal, -1, -1, 0.0, False, None, None)
                    return [t[..., :d].contiguous() for t in g[:3]]

                tn, tp = bench(fwd_native), bench(fwd_padded)
                bn, bp = bench(bwd_native), bench(bwd_padded)
                print(f"d={d:3d} causal={int(causal)} B={B:2d} S={S:5d} H={H:2d}: fwd native {tn:7.3f} ms {flops / tn / 1e9:6.0f} TF | padded-to-{dp} {tp:7.3f} ms "
                      f"{flops / tp / 1e9:6.0f} TF || bwd native {bn:7.3f} ms {2.5 * flops / bn / 1e9:6.0f} TF | padded {bp:7.3f} ms {2.5 * flops / bp / 1e9:6.0f} TF   [{name}]",
                      flush=True)


if __name__ == "__main__":
    main()
