"""HIP-graph capture of the hot path (torch.cuda.graph): capture once, replay on NEW contents of the same buffers, compare with an eager call bit for bit.
usage: graph_probe.py case [case ...]   cases: fwd_il fwd_w64 bwd_pair_short bwd_pair_w64 bwd_fused_default autograd
Run as its own process (tests/test_hip_graph_gpu.py does): PyTorch-ROCm 2.10 segfaults in capture_end when autograd ran eagerly on the default stream earlier in the
process and the capture contains autograd again -- with its own scaled_dot_product_attention as well (measured round 6) -- which a test suite's process always has."""
import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be, flash_attn_func

SHAPES = {"fwd_il": (2, 512, 4, 128), "fwd_w64": (1, 4096, 4, 128), "bwd_pair_short": (2, 1024, 4, 128), "bwd_pair_w64": (1, 4096, 4, 128),
          "bwd_fused_default": (8, 1024, 4, 128), "autograd": (8, 1024, 4, 128)}


def capture(step):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = step()
    return g, outs


def run(case):
    B, S, H, D = SHAPES[case]
    torch.manual_seed(3)
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k, v, do = torch.randn_like(q), torch.randn_like(q), torch.randn_like(q)
    fwd = lambda: be.fwd(q, k, v, None, None, 0.0, D ** -0.5, True, -1, -1, 0.0, False, None)
    if case == "autograd":
        for t in (q, k, v): t.requires_grad_(True)
        def step():
            o = flash_attn_func(q, k, v, causal=True)
            return (o,) + torch.autograd.grad(o, (q, k, v), do)
    elif case.startswith("fwd"):
        step = lambda: tuple(fwd()[:2])
    else:
        out, lse = fwd()[:2]
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        def step():
            be.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, D ** -0.5, True, -1, -1, 0.0, False, None, None)
            return (dq, dk, dv)
    g, outs = capture(step)
    sch = dict(be.last_schedule())
    with torch.no_grad():
        q.copy_(torch.randn_like(q)); do.copy_(torch.randn_like(do))
        if case.startswith("bwd"):   # (the backward's out / lse belong to the new q)
            o2, l2 = fwd()[:2]; out.copy_(o2); lse.copy_(l2)
    g.replay(); torch.cuda.synchronize()
    got = [t.clone() for t in outs]
    ref = [t.clone() for t in step()]
    same = all(torch.equal(a, b) for a, b in zip(got, ref))
    print(f"{case}: replay == eager: {same}  kernel {sch.get('name')} bwd_spill {sch.get('bwd_spill')} bwd_dkdv_nw {sch.get('bwd_dkdv_nw')}", flush=True)


for c in sys.argv[1:]:
    run(c)
