"""Batch slices more than 2^32 bytes apart (views into 6.5 GB buffers) and a packed batch whose last sequences start 4.9 GB into the tensors: forward and backward
(64-rows-per-wave kernels, the lock-step pair, the fused launch) equal the same call on small contiguous copies bit for bit -- the 64-bit half of every kernel's
addressing (a slice's base; inside a slice offsets are 32-bit / buffer descriptors, checked host-side by w64_span_ok).  tools/big_offset_probe.py, in a process of
its own (~50 GB of device memory for a few seconds)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_slices_beyond_4_gib():
    import torch
    if torch.cuda.get_device_properties(0).total_memory < 100 * 2 ** 30:
        pytest.skip("needs ~50 GB of device memory")
    env = {k: v for k, v in os.environ.items() if not k.startswith("FA_") or k == "FA_GFX950_LIB"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "big_offset_probe.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK") and "False" not in r.stdout, (r.returncode, r.stdout[-3000:], r.stderr[-2000:])


def test_slice_spanning_more_than_4_gib_falls_back():
    """tools/big_span_probe.py: rows 2 MB apart -- one (batch, head) slice spans 8.6 GB, past the 64-per-wave kernels' 32-bit offsets: the dispatch falls back
    (pipelined forward, lock-step backward: a tile's base in 64 bits) and the results equal the same kernels' on contiguous copies bit for bit (~60 GB for seconds)."""
    import torch
    if torch.cuda.get_device_properties(0).total_memory < 100 * 2 ** 30:
        pytest.skip("needs ~60 GB of device memory")
    env = {k: v for k, v in os.environ.items() if not k.startswith("FA_") or k == "FA_GFX950_LIB"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "big_span_probe.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK") and "False" not in r.stdout, (r.returncode, r.stdout[-3000:], r.stderr[-2000:])


def test_128k_and_256k_token_sequences():
    """tools/long_seq_probe.py: 2 048 / 4 096 key tiles per query block (causal, windowed, unmasked; GQA) -- the 64-per-wave kernels against the independent
    pipelined / lock-step kernel family on the same inputs: out <= 2e-2, LSE <= 1e-2, gradients <= 6e-2 apart (bf16, N(0,1) inputs), everything finite."""
    env = {k: v for k, v in os.environ.items() if not k.startswith("FA_") or k == "FA_GFX950_LIB"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "long_seq_probe.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK") and "BAD" not in r.stdout, (r.returncode, r.stdout[-3000:], r.stderr[-2000:])
