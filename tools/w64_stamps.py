"""Per-iteration clock stamps of fa_fwd_w64_kernel (library built with -DFA_W64_ABL=2048, see tools/ablate_w64.sh): the LSE output carries, for every
wave (64 rows), lane i = stamp i in shader clocks since the block started: 0 prologue barrier passed, 1 Q converted, 2 K_0 landed, 3 tile loop starts,
4+u iteration u done, 62 O stored.  Prints the four waves of a few blocks side by side (deltas between consecutive stamps)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be
os.environ["FA_FWD_NW"] = "64"; be.reload_knobs()
torch.manual_seed(0)
for (B, S, H, D, causal, blocks) in ((4, 4096, 32, 128, True, (15, 8, 2, 0)), (4, 4096, 32, 128, False, (0, 7))):
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
    f = lambda: be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None)
    for _ in range(20): f()
    torch.cuda.synchronize()
    lse = f()[1].float().cpu()   # (B, H, S)
    for (b, h) in ((0, 0), (3, 17)):
        for mb in blocks:
            st = lse[b, h, mb * 256:(mb + 1) * 256].reshape(4, 64)
            n_it = (mb * 256 + 256 + 63) // 64 + 1 if causal else S // 64 + 1
            n_it = min(n_it, 58)
            print(f"S={S} causal={int(causal)} b={b} h={h} m_block={mb}: iterations={n_it}; per wave: prologue stamps 0..3, O stored (62)")
            for w in range(4):
                s = st[w]
                print(f"  wave {w}: bar {int(s[0])} qconv {int(s[1])} k0 {int(s[2])} loop0 {int(s[3])} | loop_end {int(s[3 + n_it])} end {int(s[62])}")
            print("  iteration deltas (rows = iteration u, columns = waves 0..3):")
            for u in range(n_it):
                d = [int(st[w][4 + u] - st[w][3 + u]) for w in range(4)]
                print(f"    u={u:2d}: " + " ".join(f"{x:6d}" for x in d))
