"""The fused backward launch (persistent workgroups, spin-waited hand-offs) next to other work on the GPU: two streams run it concurrently on their own tensors while
a third keeps the CUs busy with GEMMs; every result must equal the serial run's bit for bit and no hand-off may time out (FA_BWD_FUSED_CHECK=1 reads the launch's
error flag after every call).  usage: fused_concurrency_probe.py [rounds]"""
import os, sys, threading
os.environ["FA_BWD_FUSED_CHECK"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(0)
D = 128
def make(B, S, H, causal):
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k, v, do = torch.randn_like(q), torch.randn_like(q), torch.randn_like(q)
    out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None)
    return dict(q=q, k=k, v=v, do=do, out=out, lse=lse, causal=causal)
def bwd(t):
    dq, dk, dv = torch.empty_like(t["q"]), torch.empty_like(t["k"]), torch.empty_like(t["v"])
    be.bwd(t["do"], t["q"], t["k"], t["v"], t["out"], t["lse"], dq, dk, dv, None, 0.0, D ** -0.5, t["causal"], -1, -1, 0.0, False, None, None)
    return dq, dk, dv
jobs = [make(8, 1024, 4, True), make(4, 2048, 8, True), make(16, 512, 2, False), make(1, 4096, 32, True)]
serial = [bwd(t) for t in jobs]; torch.cuda.synchronize()
assert be.last_schedule()["bwd_spill"] == 3, be.last_schedule()
a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16); b = torch.randn_like(a)
streams = [torch.cuda.Stream() for _ in range(3)]
bad, errs = [], []
def worker(si, idxs):
    try:
        with torch.cuda.stream(streams[si]):
            for r in range(rounds):
                for i in idxs:
                    g = bwd(jobs[i])
                    streams[si].synchronize()
                    if not all(torch.equal(x, y) for x, y in zip(g, serial[i])): bad.append((si, r, i))
    except Exception as e:   # a timed-out hand-off raises here
        errs.append(repr(e))
def hog():
    with torch.cuda.stream(streams[2]):
        for _ in range(rounds * 6): (a @ b)
ths = [threading.Thread(target=worker, args=(0, [0, 1])), threading.Thread(target=worker, args=(1, [2, 3])), threading.Thread(target=hog)]
[t.start() for t in ths]; [t.join() for t in ths]; torch.cuda.synchronize()
print(f"rounds {rounds}: mismatches {len(bad)} {bad[:5]}  errors {errs[:3]}")
print("OK" if not bad and not errs else "FAILED")
