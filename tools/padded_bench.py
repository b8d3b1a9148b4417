"""flash_attn_padded_func (padded batch attended in place) against the reference's chain unpad_input -> flash_attn_varlen_func -> pad_input:
forward and forward+backward time and peak memory, right-padded batches with random lengths."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import flash_attn_padded_func, flash_attn_varlen_func
from flash_attn_amd.bert_padding import pad_input, padded_batch_args, unpad_input


def t_ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return statistics.median(ts)


for (B, S, H, D, causal, lo) in ((16, 4096, 16, 128, True, 0.25), (64, 1024, 16, 128, False, 0.25), (8, 8192, 32, 128, True, 0.5), (32, 2048, 16, 64, False, 0.1)):
    gen = torch.Generator().manual_seed(B + S)
    lens = (torch.rand(B, generator=gen) * (1 - lo) + lo).mul(S).long().clamp(1, S); lens[0] = S
    mask = (torch.arange(S)[None, :] < lens[:, None]).cuda()
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    k = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    v = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    do = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16)

    def chain(bwd):
        qu, idx, cu, mx, _ = unpad_input(q, mask)
        ku = unpad_input(k, mask)[0]; vu = unpad_input(v, mask)[0]
        out = pad_input(flash_attn_varlen_func(qu, ku, vu, cu, cu, mx, mx, causal=causal), idx, B, S)
        if bwd: torch.autograd.grad(out, (q, k, v), do)
        return out

    def fused(bwd):
        ln, st = padded_batch_args(mask)
        out = flash_attn_padded_func(q, k, v, ln, starts_q=st, causal=causal)
        if bwd: torch.autograd.grad(out, (q, k, v), do)
        return out

    res = []
    for name, fn in (("chain", chain), ("fused", fused)):
        f_ms = t_ms(lambda: fn(False)); fb_ms = t_ms(lambda: fn(True))
        torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats(); base = torch.cuda.memory_allocated()
        fn(True); torch.cuda.synchronize()
        res.append((name, f_ms, fb_ms, (torch.cuda.max_memory_allocated() - base) / 2 ** 20))
    fill = float(lens.float().mean()) / S
    print(f"B={B} S={S} H={H} D={D} causal={int(causal)} mean fill {fill:.2f}: " + "  ".join(f"[{n}] fwd {a:.3f} ms, fwd+bwd {b:.3f} ms, peak {m:.0f} MiB" for n, a, b, m in res)
          + f"  -> fused/chain time {res[1][1] / res[0][1]:.2f} fwd, {res[1][2] / res[0][2]:.2f} fwd+bwd", flush=True)
