"""Builds experiments/libfa_gfx950_experiments.so: the product sources compiled with -DFA_EXPERIMENTS=1 plus the kernel that was measured and
did not win (see README.md).  Select it with FA_GFX950_LIB=<this file's directory>/libfa_gfx950_experiments.so; FA_BWD_MODE=2 picks
the experiment kernel (the 64-keys-per-wave dK/dV experiment of round 2 was superseded by csrc/fa_bwd_dkdv_w64.hip in round 5).  Tests: FA_GFX950_LIB=... python -m pytest experiments/test_bwd_schedules_gpu.py -m gpu"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "flash-attention_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
OUT = os.path.join(HERE, "build")


def main():
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "flash-attention_amd", "build.py"), "--no-torch-ext"])   # the unchanged objects are reused
    base = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DFA_EXPERIMENTS=1", "-I", CSRC, "-I", os.path.join(ROOT, "include")]
    units = [(os.path.join(CSRC, "fa_bwd.hip"), "x_bwd_dkdv.o", ["-DFA_BWD_PART=1"]),
             (os.path.join(CSRC, "fa_bwd_w64.hip"), "x_bwd_w64.o", ["-fno-slp-vectorize"]),
             (os.path.join(CSRC, "fa_api.cpp"), "x_api.o", ["-x", "hip"])]
    cmds = [base + extra + ["-c", src, "-o", os.path.join(OUT, obj)] for src, obj, extra in units]
    with ThreadPoolExecutor(max_workers=len(cmds)) as ex:
        list(ex.map(subprocess.check_call, cmds))
    reuse = [os.path.join(CSRC, o) for o in ("fa_fwd_bf16.o", "fa_fwd_f16.o", "fa_fwd_il.o", "fa_fwd_w64_bf16.o", "fa_fwd_w64_f16.o", "fa_bwd_dq.o", "fa_bwd_fused.o", "fa_bwd_dkdv_w64.o")]
    lib = os.path.join(HERE, "libfa_gfx950_experiments.so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + reuse + [os.path.join(OUT, o) for _, o, _ in units])
    print("built", lib)


if __name__ == "__main__":
    main()
