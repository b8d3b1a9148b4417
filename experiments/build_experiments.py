"""Builds experiments/libfa_gfx950_experiments.so: the product sources with experiments/ds_spill.patch applied (copies under experiments/build/, the tree
is not touched) plus the kernel that was measured and did not win (fa_bwd_dq_ds.inc.hip, see README.md).  Select it with
FA_GFX950_LIB=<this file's directory>/libfa_gfx950_experiments.so; FA_BWD_MODE=2 picks the experiment kernel.
Tests: FA_GFX950_LIB=... python -m pytest experiments/test_bwd_schedules_gpu.py -m gpu"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "flash-attention_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
OUT = os.path.join(HERE, "build")
PATCHED = ("fa_api.cpp", "fa_launch.h", "fa_bwd_w64.hip")   # the files ds_spill.patch touches


def patched_sources():
    """Copies of the files the patch touches, patched, next to each other under experiments/build/ (quote includes look there first; the rest comes from csrc/)."""
    os.makedirs(OUT, exist_ok=True)
    patch = open(os.path.join(HERE, "ds_spill.patch")).read()
    for f in PATCHED:
        shutil.copy(os.path.join(CSRC, f), os.path.join(OUT, f))
    subprocess.run(["patch", "-p3", "--no-backup-if-mismatch", "-d", OUT], input=patch.encode(), check=True, stdout=subprocess.DEVNULL)


def main():
    subprocess.check_call([sys.executable, os.path.join(ROOT, "flash-attention_amd", "build.py"), "--no-torch-ext"])   # the unchanged objects are reused
    patched_sources()
    base = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", OUT, "-I", CSRC, "-I", os.path.join(ROOT, "include")]
    units = [(os.path.join(OUT, "fa_bwd_w64.hip"), "x_bwd_w64.o", ["-fno-slp-vectorize"]),
             (os.path.join(OUT, "fa_api.cpp"), "x_api.o", ["-x", "hip"])]
    cmds = [base + extra + ["-c", src, "-o", os.path.join(OUT, obj)] for src, obj, extra in units]
    with ThreadPoolExecutor(max_workers=len(cmds)) as ex:
        list(ex.map(subprocess.check_call, cmds))
    reuse = [os.path.join(CSRC, o) for o in ("fa_fwd_bf16.o", "fa_fwd_f16.o", "fa_fwd_il.o", "fa_fwd_w64_bf16.o", "fa_fwd_w64_f16.o", "fa_bwd_dkdv.o", "fa_bwd_dq.o", "fa_bwd_fused.o",
                                             "fa_bwd_dkdv_w64.o")]
    lib = os.path.join(HERE, "libfa_gfx950_experiments.so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + reuse + [os.path.join(OUT, o) for _, o, _ in units])
    print("built", lib)


if __name__ == "__main__":
    main()
