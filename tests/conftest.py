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


# Modules whose tests compare two PATHS bit for bit (varlen against per-sequence calls, the fused / chunked launches against the pair, one dK/dV kernel against the other):
# those equalities are statements about kernel texts and hold between UNSPLIT GQA groups -- the group split of the dK/dV kernels (fa_api.cpp bwd_gsplit_plan, late round 6)
# rounds its partial dK / dV to the input dtype before summing them and is chosen per call from the grid's size, so a one-sequence call may split where the packed batch
# does not.  They run with FA_BWD_GSPLIT=0; tests/test_bwd_gsplit_gpu.py covers the split (forced, automatic, off), every other module runs with the default.
_UNSPLIT_MODULES = {"test_bwd_gpu", "test_bwd_schedules_gpu", "test_bwd_c5_gpu", "test_bwd_dkdv_w64_gpu", "test_bwd_alibi_w64_gpu", "test_headdim_trimmed_gpu"}


@pytest.fixture(autouse=True)
def _pin_group_split(request, monkeypatch):
    if request.module.__name__.split(".")[-1] in _UNSPLIT_MODULES:
        from flash_attn_amd import backend
        monkeypatch.setenv("FA_BWD_GSPLIT", "0")
        backend.reload_knobs()
        yield
        monkeypatch.undo()
        backend.reload_knobs()
    else:
        yield
