"""Randomised parity sweep (GPU): random shapes / head dims / masks / feature SUBSETS through the public interface mirror, forward and
backward against the fp64 oracle.  usage: fuzz_gpu.py [cases] [seed] [--big]

Importable: run(n_cases, seed, big=False) -> (cases, worst error/tolerance per (tensor, feature set), failures); tests/test_fuzz_gpu.py runs
two seeded slices of a FIXED number of cases inside `pytest -m gpu`.  --big mixes in long key loops (Sk up to 8192, small batch / head
counts so that the fp64 oracle stays within seconds per case).

Features are drawn independently -- ALiBi x softcap x dropout x {full, causal, local} x {fixed, varlen} -- so their products occur (the
reference crosses them the same way, tests/test_flash_attn.py:903-1170, :1172-1490); varlen cases run with dropout too (the keep mask comes
from the return_attn_probs payload and is fed to the oracle).

Acceptance = the reference's own rule (tests/test_flash_attn.py:1121,1130): |ours - ref| <= 2x (out) / 3x (gradients) the error of a plain
PyTorch implementation computing in the SAME dtype, where ref = the fp64 oracle; plus one input-dtype ulp of the largest reference value
(the rule has no slack of its own: on a two-key row both errors are a single rounding and a coin flip decides), plus -- for dq / dk only -- the
a-priori size of the one rounding the kernels (ours and the reference's, flash_bwd_preprocess_kernel.h:40-48) cannot avoid and the PyTorch
baseline does not have: delta_i = sum_d dO.O is formed from the 16-bit ROUNDED output, so dS carries 2^-9-relative noise of |dO.O| per query,
which dk sums over the queries (dq over the keys): 4 sigma of a random walk = 4 * 2^-9 * sqrt(D * n) * scale * rms(dO) rms(O) rms(q or k)
(rms(O) of the kept rows under dropout; the largest component instead of the rms where the walk has fewer than eight steps).
"""
import os, sys, time, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flash-attention_amd"))
import numpy as np
import torch


def _rms(x):
    x = np.asarray(x, dtype=np.float64)
    return float(np.sqrt((x * x).mean())) if x.size else 0.0


def torch_baseline(q, k, v, g, causal, window, softcap, alibi, pd, keep):
    """Plain PyTorch attention + autograd with every feature, math in the input dtype (the error yardstick of the acceptance rule).
    q (B,Sq,H,D), k/v (B,Sk,Hk,D); alibi (H,) or None; keep bool (B,H,Sq,Sk) or None.  Returns out, dq, dk, dv."""
    q, k, v = (t.detach().clone().requires_grad_() for t in (q, k, v))
    B, Sq, H, D = q.shape
    Sk, Hk = k.shape[1], k.shape[2]
    kk, vv = k.repeat_interleave(H // Hk, dim=2), v.repeat_interleave(H // Hk, dim=2)
    s = torch.einsum("bthd,bshd->bhts", q * (D ** -0.5), kk)
    if softcap > 0:
        s = softcap * torch.tanh(s / softcap)
    i = torch.arange(Sq, device=q.device)[:, None] + (Sk - Sq)
    j = torch.arange(Sk, device=q.device)[None, :]
    if alibi is not None:
        s = s + (-(i - j).abs().to(s.dtype))[None, None] * alibi.to(s.dtype)[None, :, None, None]
    wl, wr = (window[0], 0) if causal else window
    masked = torch.zeros(Sq, Sk, dtype=torch.bool, device=q.device)
    if wr >= 0 and wr < Sk: masked |= j > i + wr
    if wl >= 0 and wl < Sk: masked |= j < i - wl
    s = s.masked_fill(masked, float("-inf"))
    p = torch.softmax(s, dim=-1).masked_fill(masked.all(-1)[None, None, :, None], 0.0)
    if pd > 0:
        p = p * keep.to(p.dtype) / (1 - pd)
    out = torch.einsum("bhts,bshd->bthd", p, vv)
    dq, dk, dv = torch.autograd.grad(out, (q, k, v), g)
    return out.detach(), dq, dk, dv


def run(n_cases, seed=0, big=False, verbose=True):
  from flash_attn_amd import flash_attn_interface as fi
  from oracle import attention_oracle as orc
  rng = np.random.default_rng(seed)
  worst, failures = {}, []
  for n in range(1, n_cases + 1):
      dtype = [torch.bfloat16, torch.float16][rng.integers(2)]
      d = int(rng.choice([32, 40, 64, 72, 96, 128, 160, 192, 256]))
      hk = int(rng.choice([1, 2, 3, 4])); h = hk * int(rng.choice([1, 2, 4]))
      B = int(rng.integers(1, 4))
      sq = int(rng.choice([1, 7, 33, 64, 65, 127, 128, 129, 200, 256, 300, 513])); sk = int(rng.choice([1, 5, 63, 64, 65, 128, 191, 256, 257, 400, 640]))
      if big and n % 3 == 0:  # long key loops: every schedule the heuristic picks for S >= 3k, fp64 oracle kept to seconds
          d = int(rng.choice([64, 128, 96])); hk = int(rng.choice([1, 2])); h = hk * int(rng.choice([1, 2])); B = 1
          sq = int(rng.choice([513, 1024, 2000, 4096])); sk = int(rng.choice([3000, 4096, 6144, 8192]))
      mode = rng.choice(["full", "causal", "local"])
      window = (-1, -1) if mode != "local" else (int(rng.integers(0, sk + 10)), int(rng.integers(0, sk + 10)))
      use_alibi, use_cap, use_drop = (bool(rng.random() < 0.3) for _ in range(3))   # independent: products of features occur
      feat = "+".join([f for f, on in (("alibi", use_alibi), ("softcap", use_cap), ("dropout", use_drop)) if on]) or "none"
      varlen = bool(rng.integers(2)) and sq > 1
      torch.manual_seed(n)
      causal = mode == "causal"
      kw = dict(causal=causal, window_size=window)
      alibi = torch.rand(h, device="cuda") * 0.3 if use_alibi else None
      if use_alibi: kw["alibi_slopes"] = alibi
      cap = float(rng.choice([5.0, 30.0])) if use_cap else 0.0
      if use_cap: kw["softcap"] = cap
      pd = float(rng.choice([0.1, 0.25])) if use_drop else 0.0
      if use_drop: kw["dropout_p"] = pd; kw["return_attn_probs"] = True
      thr8 = math.floor(255 * (1 - pd))
      al_np = None if alibi is None else alibi.cpu().numpy()
      desc = f"dtype={dtype} d={d} h={h}/{hk} B={B} sq={sq} sk={sk} mode={mode} window={window} feat={feat} varlen={varlen}"
      try:
          if not varlen:
              q = torch.randn(B, sq, h, d, device="cuda", dtype=dtype, requires_grad=True)
              k = torch.randn(B, sk, hk, d, device="cuda", dtype=dtype, requires_grad=True)
              v = torch.randn(B, sk, hk, d, device="cuda", dtype=dtype, requires_grad=True)
              res = fi.flash_attn_func(q, k, v, **kw)
              out = res[0] if pd else res
              keep_t = None if not pd else (res[2].to(torch.int32) <= thr8)
              g = torch.randn_like(out)
              dq, dk, dv = torch.autograd.grad(out, (q, k, v), g)
              keep = None if keep_t is None else keep_t.cpu().numpy()
              o_ref, _ = orc.attention_fwd(q, k, v, None, causal, window, cap, al_np, pd, keep)
              gr = orc.attention_bwd(g, q, k, v, None, None, None, causal, window, cap, al_np, pd, keep)
              pt = torch_baseline(q, k, v, g, causal, window, cap, alibi, pd, keep_t)
              got, ref = (out, dq, dk, dv), (o_ref, gr[0], gr[1], gr[2])
          else:
              lq = [int(x) for x in rng.integers(0, sq + 1, size=B + 1)]; lq[0] = sq
              lk = [max(1, l + int(x)) for l, x in zip(lq, rng.integers(-3, 40, size=B + 1))] if mode != "causal" else lq
              cq = [0] + [int(x) for x in np.cumsum(lq)]; ck = [0] + [int(x) for x in np.cumsum(lk)]
              cu_q = torch.tensor(cq, dtype=torch.int32, device="cuda"); cu_k = torch.tensor(ck, dtype=torch.int32, device="cuda")
              q = torch.randn(sum(lq), h, d, device="cuda", dtype=dtype, requires_grad=True)
              k = torch.randn(sum(lk), hk, d, device="cuda", dtype=dtype, requires_grad=True)
              v = torch.randn(sum(lk), hk, d, device="cuda", dtype=dtype, requires_grad=True)
              res = fi.flash_attn_varlen_func(q, k, v, cu_q, cu_k, max(lq), max(lk), **kw)
              out = res[0] if pd else res
              keep_all = None if not pd else (res[2].to(torch.int32) <= thr8)   # (H, total_q, max_seqlen_k)
              g = torch.randn_like(out)
              dq, dk, dv = torch.autograd.grad(out, (q, k, v), g)
              ref = [np.zeros(t.shape) for t in (out, q, k, v)]
              pt = [torch.zeros_like(t) for t in (out, q, k, v)]
              for b in range(len(lq)):   # sequence by sequence: the oracle and the baseline on (1, len, H, D) slices, this sequence's keep mask
                  q0, q1, k0, k1 = cq[b], cq[b + 1], ck[b], ck[b + 1]
                  if q1 == q0:
                      continue
                  kt = None if keep_all is None else keep_all[:, q0:q1, :k1 - k0][None]
                  kn = None if kt is None else kt.cpu().numpy()
                  qs, ks, vs, gs = q[None, q0:q1], k[None, k0:k1], v[None, k0:k1], g[None, q0:q1]
                  o_b, _ = orc.attention_fwd(qs, ks, vs, None, causal, window, cap, al_np, pd, kn)
                  g_b = orc.attention_bwd(gs, qs, ks, vs, None, None, None, causal, window, cap, al_np, pd, kn)
                  ref[0][q0:q1] = o_b[0]; ref[1][q0:q1] = g_b[0][0]; ref[2][k0:k1] += g_b[1][0]; ref[3][k0:k1] += g_b[2][0]
                  p_b = torch_baseline(qs, ks, vs, gs, causal, window, cap, alibi, pd, kt)
                  pt[0][q0:q1] = p_b[0][0]; pt[1][q0:q1] = p_b[1][0]; pt[2][k0:k1] += p_b[2][0]; pt[3][k0:k1] += p_b[3][0]
              got = (out, dq, dk, dv)
          sc_ = d ** -0.5
          # (under dropout the kept rows carry all of O's energy: their rms is rms(O) / sqrt(1 - p); and a "walk" of fewer than eight steps is not averaged over -- a
          # single key's largest component stands where rms(k) would: seed 81 case 64, sk = 1 with dropout, sat 2.8x above the averaged bound with every kernel)
          noise = 4 * 2.0 ** -9 * math.sqrt(d) * sc_ * _rms(g.float().cpu()) * _rms(out.detach().float().cpu()) / math.sqrt(1.0 - pd)
          mag = lambda t, n: float(t.detach().float().abs().max()) if n < 8 else _rms(t.detach().float().cpu())
          extra = {"dk": noise * math.sqrt(max(sq, 1)) * mag(q, sq), "dq": noise * math.sqrt(max(sk, 1)) * mag(k, sk)}
          ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10   # spacing of the input dtype just above a power of two, relative
          for nm, gt, rf, bl in zip(("out", "dq", "dk", "dv"), got, ref, pt):
              rf = np.asarray(rf)
              if not rf.size:
                  continue
              err = float(np.abs(gt.detach().float().cpu().numpy() - rf).max())
              err_pt = float(np.abs(bl.detach().float().cpu().numpy() - rf).max())
              tol = (2.0 if nm == "out" else 3.0) * err_pt + ulp * float(np.abs(rf).max()) + extra.get(nm, 0.0)
              key = (nm, feat)
              worst[key] = max(worst.get(key, 0.0), err / tol if tol > 0 else (0.0 if err == 0 else float("inf")))
              if not (err <= tol):
                  failures.append((n, nm, err, tol))
                  print(f"FAIL case {n}: {nm} err {err:.3e} tol {tol:.3e} (same-dtype PyTorch {err_pt:.3e})  {desc}", flush=True)
      except Exception as e:
          failures.append((n, "exception", str(e)[:200], 0.0))
          print(f"EXC case {n}: {type(e).__name__}: {str(e)[:200]}  {desc}", flush=True)
  return n_cases, worst, failures


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    cases = int(args[0]) if len(args) > 0 else 200
    t0 = time.time()
    n, worst, failures = run(cases, int(args[1]) if len(args) > 1 else 0, big="--big" in sys.argv)
    print(f"fuzz: {n} cases in {time.time() - t0:.0f} s, {len(failures)} failures; worst error/tolerance per (tensor, feature set):",
          {f"{a}/{b}": round(c, 2) for (a, b), c in sorted(worst.items())})
