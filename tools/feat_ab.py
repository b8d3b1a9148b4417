"""Forward timings of the lock-step kernel's users: feature variants at config 3, head dim 256, decode."""
import os, sys, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be
def t(fn, reps=10):
    fn(); torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); ts = []
    for _ in range(5):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    return statistics.median(ts)
q = torch.randn(4, 4096, 32, 128, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
al = torch.rand(32, device="cuda") * 0.3
for name, kw in (("softcap30", dict(sc=30.0)), ("alibi", dict(al=al)), ("dropout0.1", dict(p=0.1))):
    m = t(lambda: be.fwd(q, k, v, None, kw.get("al"), kw.get("p", 0.0), 128 ** -0.5, True, -1, -1, kw.get("sc", 0.0), False, None))
    print(f"cfg3 fwd {name}: {m:.3f} ms {0.5498 / m * 1e3:.0f} TF")
q = torch.randn(4, 4096, 16, 256, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
m = t(lambda: be.fwd(q, k, v, None, None, 0.0, 256 ** -0.5, True, -1, -1, 0.0, False, None)); print(f"D=256 causal S=4096 fwd: {m:.3f} ms {4*4*16*4096*4096*256/2/m/1e9:.0f} TF")
q = torch.randn(32, 256, 16, 128, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
os.environ["FA_FWD_NW"] = "4"
m = t(lambda: be.fwd(q, k, v, None, None, 0.0, 128 ** -0.5, False, -1, -1, 0.0, False, None)); print(f"short S=256 B=32 fwd (lock-step): {m*1e3:.1f} us")
