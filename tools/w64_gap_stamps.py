"""Where inside a steady iteration of fa_fwd_w64_kernel the clocks go (rocprofv3 --att is not usable in this image).  Libraries built with
-DFA_W64_ABL=67584 -DFA_W64_GAPOFF=g (experiments/ablations/fa_fwd_w64.patch; tools/ablate_w64.sh VARIANTS="g0:...;g1:..."): s_memtime at the head of gaps g and g + 16 of
both steps of every steady, unmasked iteration (no wait inside the steps: the stamps are read behind the tile barrier), summed per wave and block in the LSE payload.
One child process per library; the parent assembles the sixteen two-point profiles into the cumulative clock of all 64 gap heads.
usage: python tools/w64_gap_stamps.py            (parent: every gpurun_abl/libfa_g<k>.so)
       python tools/w64_gap_stamps.py --child     (FA_GFX950_LIB set by the parent)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def child():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
    import torch
    from flash_attn_amd import backend as be
    os.environ["FA_FWD_NW"] = "64"; be.reload_knobs()
    torch.manual_seed(0)
    out = {}
    for (B, S, H, D, causal) in ((4, 4096, 32, 128, True), (4, 4096, 32, 128, False)):
        q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
        f = lambda: be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, -1, -1, 0.0, False, None)
        for _ in range(20): f()
        torch.cuda.synchronize()
        st = f()[1].float().reshape(B, H, S // 64, 64).double()          # (.., wave rows, lane)
        n = st[..., 14].sum()
        out[str(int(causal))] = [float(st[..., i].sum() / n) for i in range(8, 14)] + [float(n)]
    print("GAPSTAMPS " + json.dumps(out))

def parent():
    rows = {}
    for g in range(16):
        lib = os.path.join(ROOT, "gpurun_abl", f"libfa_g{g}.so")
        if not os.path.exists(lib): continue
        r = subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, FA_GFX950_LIB=lib), capture_output=True, text=True, timeout=300)
        for line in r.stdout.splitlines():
            if line.startswith("GAPSTAMPS "): rows[g] = json.loads(line[10:])
        if g not in rows: print(f"g{g}: no result: {r.stderr[-300:]}")
    for c in ("1", "0"):
        print(f"# causal={c}: mean clocks from the head of the iteration's first step (steady, unmasked iterations; every wave of every block)")
        cum = {}
        ends = []
        for g, d in sorted(rows.items()):
            a = d[c]
            cum[(0, g)] = a[0]; cum[(0, g + 16)] = a[1]; cum[(1, g)] = a[2]; cum[(1, g + 16)] = a[3]
            ends.append((a[4], a[5], a[6]))
        if not ends: continue
        e1 = sum(x[0] for x in ends) / len(ends); e2 = sum(x[1] for x in ends) / len(ends)
        print(f"#   end of step 2 at {e1:.0f}, behind the tile barrier at {e2:.0f} (mean over the {len(ends)} builds; iterations sampled per build ~{ends[0][2]:.0f})")
        for step in (0, 1):
            pts = [(x, cum[(step, x)]) for x in range(32) if (step, x) in cum]
            line = f"step {step + 1} gap head:clock  " + " ".join(f"{x}:{t:.0f}" for x, t in pts)
            print(line)
            d = [(pts[i][0], pts[i + 1][1] - pts[i][1]) for i in range(len(pts) - 1) if pts[i + 1][0] == pts[i][0] + 1]
            print(f"step {step + 1} clocks per gap  " + " ".join(f"{x}:{t:.0f}" for x, t in d))
            top = sorted(d, key=lambda z: -z[1])[:5]
            print(f"step {step + 1} slowest gaps    " + " ".join(f"{x}:{t:.0f}" for x, t in top))

if __name__ == "__main__":
    child() if "--child" in sys.argv else parent()
