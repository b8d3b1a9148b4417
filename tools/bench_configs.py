"""Time the five BASELINE.json configurations (forward, backward, forward+backward) on one GPU."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import numpy as np
import torch
from flash_attn_amd import backend as be
from oracle.attention_oracle import visible_keys_per_row


def t_ms(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return statistics.median(ts)


def fixed(name, B, S, H, Hk, D, causal, window, do_bwd=True):
    torch.manual_seed(0)
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, S, Hk, D, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k)
    sc = D ** -0.5
    f = lambda: be.fwd(q, k, v, None, None, 0.0, sc, causal, window[0], window[1], 0.0, False, None)
    fl = 4.0 * B * H * D * float(visible_keys_per_row(S, S, causal, window).sum())
    ms = t_ms(f)
    line = f"{name}: fwd {ms:.3f} ms {fl / ms / 1e9:7.1f} TF"
    if do_bwd:
        out, lse, _, _ = f(); do = torch.randn_like(out)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        g = lambda: be.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, sc, causal, window[0], window[1], 0.0, False, None, None)
        mb = t_ms(g, 5)
        line += f" | bwd {mb:.3f} ms {2.5 * fl / mb / 1e9:7.1f} TF | fwd+bwd {3.5 * fl / (ms + mb) / 1e9:7.1f} TF"
    print(line, flush=True)


def varlen(name, lens, H, D, causal=True):
    torch.manual_seed(0)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device="cuda")
    tot = int(cu[-1])
    q = torch.randn(tot, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
    sc = D ** -0.5; mx = max(lens)
    f = lambda: be.varlen_fwd(q, k, v, None, cu, cu, None, None, None, None, mx, mx, 0.0, sc, False, causal, -1, -1, 0.0, False, None)
    fl = sum(2.0 * H * s * s * D for s in lens) * (1 if causal else 2)
    ms = t_ms(f)
    out, lse, _, _ = f(); do = torch.randn_like(out)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    g = lambda: be.varlen_bwd(do, q, k, v, out, lse, dq, dk, dv, cu, cu, None, mx, mx, 0.0, sc, False, causal, -1, -1, 0.0, False, None, None)
    mb = t_ms(g, 5)
    print(f"{name} (total {tot}, {len(lens)} seqs, max {mx}): fwd {ms:.3f} ms {fl / ms / 1e9:7.1f} TF | bwd {mb:.3f} ms {2.5 * fl / mb / 1e9:7.1f} TF | "
          f"fwd+bwd {3.5 * fl / (ms + mb) / 1e9:7.1f} TF", flush=True)


def long_tail_lengths(total=65536, seed=0):
    """Long-tail lengths in the spirit of benchmarks/benchmark_varlen_sched.py:76-84 (generator seed 0), trimmed to `total`."""
    g = torch.Generator().manual_seed(seed)
    lens = []
    while sum(lens) < total:
        x = float(torch.rand(1, generator=g))
        s = int(64 * (1.0 / max(x, 1e-3)) ** 0.9)
        lens.append(max(16, min(s, 16384)))
    lens[-1] -= sum(lens) - total
    if lens[-1] <= 0: lens.pop(); lens[-1] += total - sum(lens)
    return lens


if __name__ == "__main__":
    fixed("cfg2 B8 H16 S2048 D64 non-causal", 8, 2048, 16, 16, 64, False, (-1, -1), do_bwd=True)
    fixed("cfg3 B4 H32 S4096 D128 causal", 4, 4096, 32, 32, 128, True, (-1, -1))
    varlen("cfg4i varlen 16x4096 H16 D128 causal", [4096] * 16, 16, 128)
    varlen("cfg4ii varlen long-tail H16 D128 causal", long_tail_lengths(), 16, 128)
    fixed("cfg5 B2 S8192 H32/8 D128 causal window(1024,0)", 2, 8192, 32, 8, 128, True, (1024, 0))
    fixed("extra B2 S8192 H32/8 D128 causal GQA", 2, 8192, 32, 8, 128, True, (-1, -1))
