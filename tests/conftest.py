import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "flash-attention_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_cases():
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "attention_ref_cases.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    return {n: {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(n + "/")} for n in names}
