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
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    st_all = lse.reshape(B, H, S // 64, 64)
    print(f"S={S} causal={int(causal)}: kernel {e0.elapsed_time(e1) / 20:.4f} ms; mean clocks per block {float(st_all[..., 3::4, 62].mean()):.0f}, mean block time {float(st_all[..., 3::4, 63].mean()) / 100:.2f} us")
    for (b, h) in ((0, 0), (3, 17)):
        for mb in blocks:
            st = lse[b, h, mb * 256:(mb + 1) * 256].reshape(4, 64)
            n_it = (mb * 256 + 256 + 63) // 64 + 1 if causal else S // 64 + 1
            n_it = min(n_it, 58)
            print(f"S={S} causal={int(causal)} b={b} h={h} m_block={mb}: iterations={n_it}; per wave: prologue stamps 0..3, O stored (62)")
            for w in range(4):
                s = st[w]
                print(f"  wave {w}: bar {int(s[0])} qconv {int(s[1])} k0 {int(s[2])} loop0 {int(s[3])} | loop_end {int(s[3 + n_it])} end {int(s[62])} | {int(s[63]) / 100:.2f} us -> {int(s[62]) / max(1, int(s[63])) / 10:.3f} GHz")
            print("  iteration deltas (rows = iteration u, columns = waves 0..3):")
            for u in range(n_it):
                d = [int(st[w][4 + u] - st[w][3 + u]) for w in range(4)]
                print(f"    u={u:2d}: " + " ".join(f"{x:6d}" for x in d))
    # per query block (all batch entries and heads): where the clocks go, wave 3 (the last to finish under a causal mask)
    st = st_all.reshape(B, H, S // 256, 4, 64)[:, :, :, 3, :]   # (B, H, m_block, 64)
    print("  (58 iteration stamps fit the payload: for a block of more iterations -- n_it printed as 58 -- the last column is the iterations past the 58th PLUS the epilogue)")
    print("  m_block  n_it |  total | to_bar  qconv  to_loop |   u=0   mean u=1..n-5  last5 (sum) | epilogue")
    for mb in range(S // 256):
        n_it = min(((mb * 256 + 256 + 63) // 64 + 1) if causal else S // 64 + 1, 58)
        x = st[:, :, mb, :].reshape(-1, 64).double()
        d = x[:, 4:4 + n_it] - x[:, 3:3 + n_it]
        mid = d[:, 1:max(2, n_it - 5)].mean() if n_it > 6 else float("nan")
        print(f"  {mb:7d} {n_it:5d} | {x[:, 62].mean():6.0f} | {x[:, 0].mean():6.0f} {(x[:, 1] - x[:, 0]).mean():6.0f} {(x[:, 3] - x[:, 1]).mean():8.0f} | {d[:, 0].mean():6.0f} {mid:10.0f} {d[:, max(1, n_it - 5):].sum(1).mean():14.0f} | {(x[:, 62] - x[:, 3 + n_it]).mean():6.0f}")
