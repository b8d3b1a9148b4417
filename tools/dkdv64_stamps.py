"""Clock stamps of the 64-keys-per-wave dK/dV kernel (library built with -DFA_DKDV64_ABL=256: tools/ablate_dkdv64.sh): every workgroup writes {start, prologue done,
step loop done, end} (shader clocks), its step count and its key block over dq.  Prints, per key block position, the clocks of prologue / loop / epilogue and the
clocks per step -- what a workgroup costs besides its steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be
os.environ["FA_BWD_DKDV"] = "64"; os.environ["FA_BWD_MODE"] = "-1"; be.reload_knobs()
torch.manual_seed(0)
for (B, S, H, D, causal) in ((4, 4096, 32, 128, True), (4, 4096, 32, 128, False)):
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q); do = torch.randn_like(q)
    out, lse, _, _ = be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    for _ in range(3): be.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None, None)
    torch.cuda.synchronize()
    n_wg = B * H * (S // 256)
    st = dq.view(torch.int64).reshape(-1)[: n_wg * 8].reshape(n_wg, 8).cpu().double()
    pro, loop, epi, steps, nb = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2], st[:, 4], st[:, 5]
    print(f"S={S} causal={int(causal)}: workgroups {n_wg}; mean clocks: prologue {pro.mean():.0f} loop {loop.mean():.0f} epilogue {epi.mean():.0f}; per step {(loop.sum() / steps.sum()):.0f}; "
          f"fixed share {(pro.sum() + epi.sum()) / (pro.sum() + loop.sum() + epi.sum()):.3f}")
    for n in sorted(set(int(x) for x in nb.tolist())):
        m = nb == n
        print(f"  key block {n:2d}: n {int(m.sum()):4d} steps {steps[m].mean():6.1f} | prologue {pro[m].mean():7.0f} loop {loop[m].mean():8.0f} ({loop[m].mean() / max(1.0, steps[m].mean()):6.0f} / step) epilogue {epi[m].mean():7.0f}")
