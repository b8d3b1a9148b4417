"""Cold vs warm instruction fetch of the once-per-block code of fa_fwd_w64_kernel: library built with -DFA_W64_ABL=$((2048+32768)), which runs the Q
conversion and the epilogue TWICE in a row (both idempotent) and stamps the end of each first pass (lanes 58 / 57).  If the second pass of the same
instructions is much faster than the first, the per-block code is instruction-fetch-bound, not issue-bound."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be
os.environ["FA_FWD_NW"] = "64"; be.reload_knobs()
torch.manual_seed(0)
for (B, S, H, D, causal) in ((4, 4096, 32, 128, True), (16, 1024, 16, 128, True)):
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
    f = lambda: be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None)
    for _ in range(20): f()
    torch.cuda.synchronize()
    st = f()[1].float().cpu().reshape(B, H, S // 64, 64).double()      # per wave: 64 stamps
    nb = S // 256
    print(f"S={S} causal={int(causal)}: per query block position (mean over batch, heads, waves), shader clocks")
    print("  m_block | barrier->K0/V DMA issued  ->Q frags read from LDS | Q conversion: 1st pass  2nd pass | epilogue: 1st pass  2nd pass")
    for mb in range(nb):
        x = st[:, :, 4 * mb:4 * mb + 4, :].reshape(-1, 64)
        n_it = min((mb * 256 + 256 + 63) // 64 + 1 if causal else S // 64 + 1, 53)
        d0, d1 = (x[:, 56] - x[:, 0]).mean(), (x[:, 55] - x[:, 56]).mean()
        q1, q2 = (x[:, 58] - x[:, 55]).mean(), (x[:, 1] - x[:, 58]).mean()
        e1, e2 = (x[:, 57] - x[:, 3 + n_it]).mean(), (x[:, 62] - x[:, 57]).mean()
        print(f"  {mb:7d} | {d0:24.0f} {d1:25.0f} | {q1:22.0f} {q2:9.0f} | {e1:18.0f} {e2:9.0f}")
