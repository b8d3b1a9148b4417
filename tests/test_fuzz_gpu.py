"""Two seeded slices of the randomised parity sweep (tools/fuzz_gpu.py) inside `pytest -m gpu`, a FIXED number of cases each (what a box
checks does not depend on its speed): random dtype x head dim x GQA ratio x lengths (Sk up to 8192 in the `big` third) x {full, causal, local}
x every subset of {ALiBi, softcap, dropout} x {fixed, varlen}, forward and backward through the public interface mirror against the fp64
oracle, judged by the reference's rule -- at most 2x (out) / 3x (gradients) the error of a same-dtype PyTorch implementation
(reference sweep: tests/test_flash_attn.py:903-1170, :1172-1490 cross the same features; rule :1121,1130)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed,big,cases", [(1, False, 120), (2, True, 45)])
def test_fuzz_slice(seed, big, cases):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_gpu
    n, worst, failures = fuzz_gpu.run(cases, seed, big=big)
    print(f"fuzz seed {seed} big {big}: {n} cases;", {f"{a}/{b}": round(c, 2) for (a, b), c in sorted(worst.items())})
    assert n == cases
    assert not failures, failures[:5]
