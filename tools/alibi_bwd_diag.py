"""Diagnostic: ALiBi backward on the 64-per-wave kernels vs the feature kernels vs fp32 / input-dtype PyTorch on a few shapes: max errors and where they sit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import torch
from flash_attn_amd import backend as be
from tests.test_bwd_schedules_gpu import run_bwd
from tests.test_bwd_alibi_w64_gpu import ref_grads_alibi, slopes_for, pt_grads_alibi as pt_grads


def setk(**kw):
    for k_, v_ in kw.items(): os.environ[k_] = str(v_)
    be.reload_knobs()


for (B, Sq, Sk, H, Hk, wl, pb, d) in ((1, 1024, 1024, 4, 2, -1, True, 128), (1, 4096, 4096, 2, 2, -1, False, 64), (3, 200, 1000, 8, 8, 64, True, 64), (2, 512, 512, 4, 4, -1, False, 128)):
    torch.manual_seed(0)
    q = torch.randn(B, Sq, H, d, device="cuda", dtype=torch.bfloat16); k = torch.randn(B, Sk, Hk, d, device="cuda", dtype=torch.bfloat16)
    v, do = torch.randn_like(k), torch.randn_like(q)
    sl = slopes_for(B, H, pb)
    wr = 0 if wl >= 0 else -1
    res = {}
    for name, kn in (("old", dict(FA_BWD_DQ_NW=4, FA_BWD_DKDV=8)), ("dq64", dict(FA_BWD_DQ_NW=64, FA_BWD_DKDV=8)), ("dkdv64", dict(FA_BWD_DQ_NW=4, FA_BWD_DKDV=64)), ("new", dict(FA_BWD_DQ_NW=64, FA_BWD_DKDV=64))):
        setk(**kn); res[name] = run_bwd(be, q, k, v, do, True, wl, wr, alibi=sl)
    r = ref_grads_alibi(q, k, v, do, sl, wl); pt = pt_grads(q, k, v, do, sl, wl)
    print(f"== B{B} Sq{Sq} Sk{Sk} H{H}/{Hk} wl{wl} pb{int(pb)} d{d}")
    for i, nm in enumerate(("dq", "dk", "dv")):
        e = {n: float((res[n][i].float() - r[i]).abs().max()) for n in res}
        e_pt = float((pt[i].float() - r[i]).abs().max())
        dmax = (res["new"][i].float() - r[i]).abs()
        idx = torch.nonzero(dmax == dmax.max())[0].tolist()
        print(f"  {nm}: pt {e_pt:.4f} | " + " ".join(f"{n} {e[n]:.4f}" for n in e) + f" | new's max at {idx}, ref value {float(r[i][tuple(idx)]):.3f}, |ref| max {float(r[i].abs().max()):.2f}; sched {res['new'][3]['bwd_dq_nw']}/{res['new'][3]['bwd_dkdv_nw']}")
