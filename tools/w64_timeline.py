"""Per-workgroup timeline of the persistent fa_fwd_w64_kernel from its clock stamps (library built with -DFA_W64_ABL=18432 = 2048 + 16384): lane 60 = workgroup,
59 = round, 61 = block start and 63 = block duration on the chip-wide 100 MHz clock (10 ns units).  Prints, for config 3 (causal) and the same shape
without a mask: kernel span, per-workgroup busy time (sum of block durations), gaps between a workgroup's blocks, spread of the finish times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be
os.environ["FA_FWD_NW"] = "64"; be.reload_knobs()
torch.manual_seed(0)
for (B, S, H, D, causal) in ((4, 4096, 32, 128, True), (4, 4096, 32, 128, False), (8, 2048, 16, 128, True)):
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
    f = lambda: be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None)
    for _ in range(10): f()
    torch.cuda.synchronize()
    lse = f()[1].float().cpu().reshape(B, H, S // 256, 4, 64)[:, :, :, 3, :].reshape(-1, 64)   # wave 3 of every block
    wg, rnd, t0, dur = lse[:, 60].long(), lse[:, 59].long(), lse[:, 61].long(), lse[:, 63].long()
    M = 1 << 22
    base = int(t0.min())
    t0 = (t0 - base) % M                      # (the launch is far shorter than the 42 ms wrap)
    end = t0 + dur
    span = int(end.max())
    nwg = int(wg.max()) + 1
    busy = torch.zeros(nwg, dtype=torch.long).index_add_(0, wg, dur)
    first = torch.full((nwg,), M, dtype=torch.long).scatter_reduce(0, wg, t0, "amin")
    last = torch.zeros(nwg, dtype=torch.long).scatter_reduce(0, wg, end, "amax")
    print(f"B={B} S={S} H={H} causal={int(causal)}: {nwg} workgroups, {lse.shape[0]} blocks; span of the launch (first block start -> last block end) {span / 100:.1f} us")
    print(f"  per workgroup: busy {busy.float().mean() / 100:.1f} us (min {busy.min() / 100:.1f}, max {busy.max() / 100:.1f}); first start {first.float().mean() / 100:.2f} us (max {first.max() / 100:.2f}); "
          f"finish {last.float().mean() / 100:.1f} us (min {last.min() / 100:.1f}, max {last.max() / 100:.1f}); gaps between blocks {((last - first) - busy).float().mean() / 100:.2f} us")
    for r in range(int(rnd.max()) + 1):
        m = rnd == r
        print(f"  round {r}: blocks {int(m.sum())}, duration mean {dur[m].float().mean() / 100:.1f} us (min {dur[m].min() / 100:.1f}, max {dur[m].max() / 100:.1f}); start mean {t0[m].float().mean() / 100:.1f} us")
    by_xcd = [busy[x::8].float().mean() / 100 for x in range(8)]
    print("  busy time by XCD (workgroup % 8): " + " ".join(f"{x:.1f}" for x in by_xcd))
