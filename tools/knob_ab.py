"""A/B of environment knobs in ONE process, interleaved rounds (median of 5): forward and backward of a list of shapes under each knob setting.
usage: knob_ab.py "name=ENV1=v,ENV2=v;name2=..." [--shapes cfg3|cfg2|cfg5|d64|short] [--fwd] [--bwd]        (an empty setting list = the defaults)"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be

SHAPES = {   # B, S, H, Hk, D, causal, window
    "cfg3": [(4, 4096, 32, 32, 128, True, (-1, -1)), (4, 4096, 32, 32, 128, False, (-1, -1))],
    "cfg2": [(8, 2048, 16, 16, 64, False, (-1, -1))],
    "cfg5": [(2, 8192, 32, 8, 128, True, (1024, 0)), (2, 8192, 32, 8, 128, True, (-1, -1))],
    "d64": [(16384 // S, S, 32, 32, 64, c, (-1, -1)) for c in (False, True) for S in (1024, 2048, 4096, 8192, 16384)],
    "short": [(16384 // S, S, 16, 16, 128, True, (-1, -1)) for S in (512, 1024, 2048)],
}


def t_ms(fn, reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def visible_pairs(S, causal, window):
    wl, wr = window
    if causal: wr = 0
    i = torch.arange(S, dtype=torch.int64)
    hi = torch.clamp(i + wr, max=S - 1) if wr >= 0 else torch.full_like(i, S - 1)
    lo = torch.clamp(i - wl, min=0) if wl >= 0 else torch.zeros_like(i)
    return int(torch.clamp(hi - lo + 1, min=0).sum())


def main():
    settings = []
    for item in sys.argv[1].split(";"):
        name, _, kv = item.partition("=")
        env = dict(x.split("=") for x in kv.split(",") if x)
        settings.append((name, env))
    keys = sorted({k for _, e in settings for k in e})
    which = sys.argv[sys.argv.index("--shapes") + 1].split(",") if "--shapes" in sys.argv else ["cfg3"]
    do_f, do_b = "--fwd" in sys.argv or "--bwd" not in sys.argv, "--bwd" in sys.argv

    def apply(env):
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(env)
        be.reload_knobs()

    torch.manual_seed(0)
    for grp in which:
        for (B, S, H, Hk, D, causal, win) in SHAPES[grp]:
            q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16)
            k = torch.randn(B, S, Hk, D, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k)
            fl = 4.0 * B * H * D * visible_pairs(S, causal, win)
            f = lambda: be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, win[0], win[1], 0.0, False, None)
            apply({})
            out, lse, _, _ = f()
            do = torch.randn_like(out); dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            g = lambda: be.bwd(do, q, k, v, out, lse, dq, dk, dv, None, 0.0, D ** -0.5, causal, win[0], win[1], 0.0, False, None, None)
            rf, rb, names = {n: [] for n, _ in settings}, {n: [] for n, _ in settings}, {}
            for n, e in settings:
                apply(e)
                if do_f: f(); names[n] = be.last_schedule()["name"]
                if do_b: g(); names[n] = names.get(n, "") + " dq%d" % be.last_schedule()["bwd_dq_nw"]
            for _ in range(5):
                for n, e in settings:
                    apply(e)
                    if do_f: rf[n].append(t_ms(f, 10))
                    if do_b: rb[n].append(t_ms(g, 4))
            line = f"B={B} S={S} H={H}/{Hk} D={D} c={int(causal)} w={win}:"
            for n, _ in settings:
                line += f"  [{n}]"
                if do_f: line += f" fwd {statistics.median(rf[n]):.3f} ms {fl / statistics.median(rf[n]) / 1e9:6.0f} TF"
                if do_b: line += f" bwd {statistics.median(rb[n]):.3f} ms {2.5 * fl / statistics.median(rb[n]) / 1e9:6.0f} TF"
                line += f" ({names[n].replace('fa::', '')})"
            print(line, flush=True)
    apply({})


if __name__ == "__main__":
    main()
