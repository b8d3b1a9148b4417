"""A seeded slice of the randomised parity sweep (tools/fuzz_gpu.py) inside `pytest -m gpu`: random dtype x head dim x GQA
ratio x lengths (Sk up to 8192 in the `big` third) x {full, causal, local} x {plain, ALiBi, softcap, dropout} x {fixed,
varlen}, forward and backward through the public interface mirror against the fp64 oracle (reference sweep:
tests/test_flash_attn.py:903-1170, :1172-1490 cross the same features)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed,big", [(1, False), (2, True)])
def test_fuzz_slice(seed, big):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_gpu
    n, worst, failures = fuzz_gpu.run(30.0, seed, big=big, max_cases=400)
    print(f"fuzz seed {seed} big {big}: {n} cases;", {f"{a}/{b}": round(c, 2) for (a, b), c in sorted(worst.items())})
    assert n >= 20
    assert not failures, failures[:5]
