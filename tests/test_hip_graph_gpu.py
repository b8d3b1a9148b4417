"""HIP-graph capture of the hot path (torch.cuda.graph): the forward (both kernels), the recomputing backward pair, the fused backward launch (sync-area memset +
persistent kernel) and forward + autograd through the public function are captured once and replayed on NEW contents of the same buffers; the replay must equal an
eager call bit for bit -- nothing on the path may synchronise, allocate outside the capture's pool or bake in a host value that changes between replays.
The cases run in a process of their own (tools/graph_probe.py says why)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["fwd_il", "fwd_w64", "bwd_pair_short", "bwd_pair_w64", "bwd_fused_default", "autograd"]


def test_hot_path_captures_and_replays():
    env = {k: v for k, v in os.environ.items() if not k.startswith("FA_") or k == "FA_GFX950_LIB"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "graph_probe.py")] + CASES, capture_output=True, text=True, timeout=600, env=env)
    lines = {m.group(1): m.group(0) for m in re.finditer(r"^(\w+): replay == eager: (True|False).*$", r.stdout, re.M)}
    assert r.returncode == 0 and set(lines) == set(CASES), (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert all("replay == eager: True" in l for l in lines.values()), lines
    assert "w64" in lines["fwd_w64"] and "bwd_spill 3" in lines["bwd_fused_default"] and "bwd_dkdv_nw 64" in lines["bwd_pair_w64"], lines


def test_fused_backward_next_to_other_work_on_the_gpu():
    """tools/fused_concurrency_probe.py: the fused launch (persistent workgroups, bounded spin-waits) on two streams at once while a third runs GEMMs -- every gradient
    equals the serial run's bit for bit and no hand-off times out (the launch's error flag is read after every call)."""
    env = {k: v for k, v in os.environ.items() if not k.startswith("FA_") or k == "FA_GFX950_LIB"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fused_concurrency_probe.py"), "6"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
