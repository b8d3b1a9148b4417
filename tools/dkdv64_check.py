"""Parity of the 64-keys-per-wave dK/dV kernel (fa_bwd_dkdv_w64.hip, FA_BWD_DKDV=64) against the eight-wave kernel (FA_BWD_DKDV=8) and an fp32 PyTorch
reference: fixed-length and packed batches, causal / windows / GQA / ragged lengths, bf16 and fp16, head dims 128 and 64.  Prints one line per case and
a final verdict; exit code 1 on a mismatch.  (Different accumulation orders: the two kernels agree to rounding, not bit for bit.)"""
import os, sys, itertools
os.environ["FA_BWD_GSPLIT"] = "0"   # (cross-path bitwise comparisons hold between UNSPLIT GQA groups: tests/conftest.py _UNSPLIT_MODULES)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be


def set_knob(v):
    os.environ["FA_BWD_DKDV"] = str(v)
    be.reload_knobs()


def ref_bwd(q, k, v, do, causal, win):
    B, Sq, H, D = q.shape; Sk, Hk = k.shape[1], k.shape[2]
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    ke = kf.repeat_interleave(H // Hk, dim=2); ve = vf.repeat_interleave(H // Hk, dim=2)
    s = torch.einsum("bqhd,bkhd->bhqk", qf, ke) * D ** -0.5
    i = torch.arange(Sq, device=q.device)[:, None] + (Sk - Sq); j = torch.arange(Sk, device=q.device)[None, :]
    wl, wr = win
    if causal: wr = 0
    m = torch.zeros(Sq, Sk, dtype=torch.bool, device=q.device)
    if wr >= 0: m |= j > i + wr
    if wl >= 0: m |= j < i - wl
    s = s.masked_fill(m, float("-inf"))
    p = torch.softmax(s, dim=-1).nan_to_num(0.0)
    o = torch.einsum("bhqk,bkhd->bqhd", p, ve)
    return torch.autograd.grad(o, (qf, kf, vf), do.float())


def main():
    torch.manual_seed(0)
    bad = 0
    cases = []
    for dt in (torch.bfloat16, torch.float16):
        for D in (128, 64):
            cases += [(dt, D, 2, 512, 512, 4, 4, True, (-1, -1)), (dt, D, 1, 1024, 1024, 4, 2, False, (-1, -1)), (dt, D, 2, 333, 777, 6, 2, True, (-1, -1)),
                      (dt, D, 1, 2048, 2048, 8, 2, True, (256, 0)), (dt, D, 2, 200, 200, 2, 2, False, (64, 32)), (dt, D, 1, 777, 333, 4, 4, True, (-1, -1)),
                      (dt, D, 3, 65, 513, 2, 1, False, (-1, -1)), (dt, D, 1, 4096, 4096, 4, 4, True, (-1, -1)), (dt, D, 1, 31, 31, 1, 1, True, (-1, -1))]
    for (dt, D, B, Sq, Sk, H, Hk, causal, win) in cases:
        q = torch.randn(B, Sq, H, D, device="cuda", dtype=dt)
        k = torch.randn(B, Sk, Hk, D, device="cuda", dtype=dt); v = torch.randn_like(k)
        do = torch.randn_like(q)
        set_knob(8)
        out, lse = be.fwd(q, k, v, None, None, 0.0, D ** -0.5, causal, win[0], win[1], 0.0, False, None)[:2]
        g8 = be.bwd(do, q, k, v, out, lse, None, None, None, None, 0.0, D ** -0.5, causal, win[0], win[1], 0.0, False, None, None)[:3]
        n8 = be.last_schedule()["bwd_dkdv_nw"]
        set_knob(64)
        g64 = be.bwd(do, q, k, v, out, lse, None, None, None, None, 0.0, D ** -0.5, causal, win[0], win[1], 0.0, False, None, None)[:3]
        n64 = be.last_schedule()["bwd_dkdv_nw"]
        g64b = be.bwd(do, q, k, v, out, lse, None, None, None, None, 0.0, D ** -0.5, causal, win[0], win[1], 0.0, False, None, None)[:3]
        rep = all(torch.equal(a, b) for a, b in zip(g64, g64b))
        ref = ref_bwd(q, k, v, do, causal, win)
        e8 = [float((a.float() - r).abs().max()) for a, r in zip(g8, ref)]
        e64 = [float((a.float() - r).abs().max()) for a, r in zip(g64, ref)]
        d = [float((a.float() - b.float()).abs().max()) for a, b in zip(g8, g64)]
        nan = any(bool(torch.isnan(a).any()) for a in g64)
        ok = (not nan) and rep and n64 == 64 and all(x <= 2.0 * y + 2e-3 for x, y in zip(e64[1:], e8[1:]))
        bad += not ok
        print(f"{'ok ' if ok else 'BAD'} {str(dt)[6:]:8s} D={D} B={B} Sq={Sq} Sk={Sk} H={H}/{Hk} c={int(causal)} w={win}: kernels {n8}/{n64} err vs fp32 dk {e8[1]:.2e}/{e64[1]:.2e} dv {e8[2]:.2e}/{e64[2]:.2e}"
              f" |8-64| dk {d[1]:.2e} dv {d[2]:.2e} rerun-equal={rep} nan={nan}", flush=True)
    # packed batch: varlen path with the key-block work list
    for D in (128, 64):
        lens = [700, 33, 1500, 256, 64, 1, 900, 257]
        cu = torch.tensor([0] + list(itertools.accumulate(lens)), dtype=torch.int32, device="cuda")
        tot, mx = sum(lens), max(lens)
        q = torch.randn(tot, 4, D, device="cuda", dtype=torch.bfloat16); k = torch.randn(tot, 2, D, device="cuda", dtype=torch.bfloat16); v = torch.randn_like(k)
        do = torch.randn_like(q)
        res = {}
        for kn in (8, 64):
            set_knob(kn)
            out, lse = be.varlen_fwd(q, k, v, None, cu, cu, None, None, None, None, mx, mx, 0.0, D ** -0.5, False, True, -1, -1, 0.0, False, None)[:2]
            res[kn] = be.varlen_bwd(do, q, k, v, out, lse, None, None, None, cu, cu, None, mx, mx, 0.0, D ** -0.5, False, True, -1, -1, 0.0, False, None, None)[:3]
        d = [float((a.float() - b.float()).abs().max()) for a, b in zip(res[8], res[64])]
        ok = max(d[1:]) < 3e-2 and not any(bool(torch.isnan(a).any()) for a in res[64])
        bad += not ok
        print(f"{'ok ' if ok else 'BAD'} varlen D={D} lens={lens}: |8-64| dq {d[0]:.2e} dk {d[1]:.2e} dv {d[2]:.2e}", flush=True)
    os.environ.pop("FA_BWD_DKDV", None); be.reload_knobs()
    print("VERDICT:", "all cases agree" if bad == 0 else f"{bad} cases BAD")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
