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


class _Knobs:
    """Set FA_* environment knobs for one test: the library reads its knobs once per process, so every change is
    followed by fa_knobs_reload(); the previous environment is restored (and reloaded) at teardown."""

    def __init__(self, monkeypatch):
        self._mp = monkeypatch

    def set(self, name, value):
        from flash_attn_amd import backend
        self._mp.setenv(name, str(value))
        backend.reload_knobs()

    def unset(self, name):
        from flash_attn_amd import backend
        self._mp.delenv(name, raising=False)
        backend.reload_knobs()


@pytest.fixture
def knobs(monkeypatch):
    from flash_attn_amd import backend
    yield _Knobs(monkeypatch)
    monkeypatch.undo()
    backend.reload_knobs()


@pytest.fixture(scope="session")
def golden_cases():
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "attention_ref_cases.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    return {n: {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(n + "/")} for n in names}


@pytest.fixture(scope="session")
def dropout_cases():
    """Reference oracle outputs with an explicit keep-mask (tests/golden/make_golden.py)."""
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "dropout_ref_cases.npz"))
    names = sorted({k.split("/")[0] for k in z.files})
    cases = {n: {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(n + "/")} for n in names}
    for c in cases.values():
        B, Sq, Sk, H = [int(x) for x in c["meta"][:4]]
        c["keep"] = np.unpackbits(c["keep"])[: B * H * Sq * Sk].reshape(B, H, Sq, Sk).astype(bool)
    return cases
