"""Error of the default kernels vs the FA_STRICT=1 kernels at the real shapes of BASELINE configs 2-5: max / mean |delta| of out, LSE, dQ, dK,
dV against plain PyTorch attention in fp32 (one (batch, head) at a time on the GPU), next to the error of the same PyTorch code computing in
bf16 (the yardstick of the reference's 2x / 3x acceptance rule).  -> profiles/r03_numerics_default_vs_strict.txt
The default path deviates from the reference's arithmetic in three stated places: Q (forward, 64-rows-per-wave kernel)
and K (dK/dV kernel) are multiplied by softmax_scale*log2(e) ONCE and rounded to the input dtype, and O is rescaled only when a row maximum
grows by more than 2^8.  FA_STRICT=1 scales every score in fp32 and rescales on any growth.   usage: python tools/numerics_table.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import numpy as np
import torch
from flash_attn_amd import backend as be
from tests.test_baseline_configs_gpu import ref_fwd_bwd


def stats(a, b):
    d = (a.double() - b.double()).abs()
    return float(d.max()), float(d.mean())


def run_fixed(B, S, H, Hk, D, causal, window, bwd):
    torch.manual_seed(0)
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16)
    k = torch.randn(B, S, Hk, D, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(B, S, Hk, D, device="cuda", dtype=torch.bfloat16)
    do = torch.randn_like(q) if bwd else None
    sc = D ** -0.5
    ref = ref_fwd_bwd(q, k, v, do, causal, window, True)
    pt = ref_fwd_bwd(q, k, v, do, causal, window, False)
    rows = {}
    for strict in ("0", "1"):
        os.environ["FA_STRICT"] = strict; be.reload_knobs()
        out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, sc, causal, window[0], window[1], 0.0, False, None)
        name = be.last_schedule()["name"]
        got = [out, lse]
        if bwd:
            dq, dk, dv, _ = be.bwd(do, q, k, v, out, lse, None, None, None, None, 0.0, sc, causal, window[0], window[1], 0.0, False, None, None)
            got += [dq, dk, dv]
        rows[strict] = (name, got)
    os.environ.pop("FA_STRICT", None); be.reload_knobs()
    names = ["out", "lse", "dq", "dk", "dv"][: 5 if bwd else 2]
    fin = torch.isfinite(ref[1])
    for i, nm in enumerate(names):
        r = ref[i] if nm != "lse" else ref[1][fin]
        cells = []
        for strict in ("0", "1"):
            g = rows[strict][1][i] if nm != "lse" else rows[strict][1][1][fin]
            cells.append("%.3e / %.3e" % stats(g.float(), r.float()))
        p = pt[i] if nm != "lse" else pt[1][fin]
        cells.append("%.3e / %.3e" % stats(p.float(), r.float()))
        print(f"  {nm:4s} | default {cells[0]} | strict {cells[1]} | PyTorch bf16 {cells[2]}")
    print(f"  kernels: default {rows['0'][0]}, strict {rows['1'][0]}")


if __name__ == "__main__":
    print("# max / mean |delta| vs fp32 PyTorch attention; bf16 inputs N(0,1), seed 0")
    for title, args in (("config 2: B=8 H=16 S=2048 D=64 non-causal, forward", (8, 2048, 16, 16, 64, False, (-1, -1), False)),
                        ("config 3: B=4 H=32 S=4096 D=128 causal, forward + backward", (4, 4096, 32, 32, 128, True, (-1, -1), True)),
                        ("config 4-i (one sequence of the packed batch: the varlen kernels are bit-identical per sequence): B=1 H=16 S=4096 D=128 causal", (1, 4096, 16, 16, 128, True, (-1, -1), True)),
                        ("config 5: B=2 H=32/8 S=8192 D=128 causal window 1024, forward + backward", (2, 8192, 32, 8, 128, True, (1024, 0), True))):
        print(title)
        run_fixed(*args)
