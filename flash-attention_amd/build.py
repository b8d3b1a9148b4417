"""Build the gfx950 attention libraries in-tree (no JIT cache, no pip install).

  libfa_gfx950.so      C-ABI library (include/fa_gfx950.h): HIP kernels + host validation
  flash_attn_2_cuda*.so  torch extension exposing the reference backend-module API
                         (fwd / varlen_fwd / bwd / varlen_bwd / fwd_kvcache) on top of the C ABI
  probe_gfx950         hardware-semantics probe (lane layouts the kernels rely on)

Usage: python flash-attention_amd/build.py [--no-torch-ext] [--force]
The library holds only what the dispatch can pick; kernels that were measured and did not win live under experiments/ with a build script of
their own (experiments/build_experiments.py -> experiments/libfa_gfx950_experiments.so).
Outputs land next to this file so they travel to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
import sysconfig
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

LIB = os.path.join(HERE, "libfa_gfx950.so")
PROBE = os.path.join(HERE, "probe_gfx950")


def ext_path() -> str:
    suffix = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
    return os.path.join(HERE, "flash_attn_2_cuda" + suffix)


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    t0 = time.time()
    subprocess.check_call(cmd)
    print(f"  ({time.time() - t0:.1f}s)", flush=True)


def _sources(names):
    return [os.path.join(CSRC, n) for n in names]


def build_lib(force=False):
    hdrs = _sources(["fa_device.h", "fa_kernel_params.h", "fa_launch.h"]) + [os.path.join(ROOT, "include", "fa_gfx950.h")]
    # (source, object, extra flags).  fa_fwd.hip, fa_fwd_w64.hip and fa_bwd.hip are compiled twice each, side by side (bf16 / fp16 instantiations of the
    # forwards: FA_FWD_PART, FA_W64_PART; dK/dV part / dQ part / fused backward: FA_BWD_PART): they are the slowest units of the build.  The 64-per-wave units place their row-sum adds by hand (see the note on asm
    # helpers in fa_w64_asm.h), hence -fno-slp-vectorize there.
    units = [("fa_fwd.hip", "fa_fwd_bf16.o", ["-DFA_FWD_PART=1"]), ("fa_fwd.hip", "fa_fwd_f16.o", ["-DFA_FWD_PART=2"]), ("fa_fwd_il.hip", "fa_fwd_il.o", []), ("fa_fwd_w64.hip", "fa_fwd_w64_bf16.o", ["-fno-slp-vectorize", "-DFA_W64_PART=1"]), ("fa_fwd_w64.hip", "fa_fwd_w64_f16.o", ["-fno-slp-vectorize", "-DFA_W64_PART=2"]),
             ("fa_bwd.hip", "fa_bwd_dkdv.o", ["-DFA_BWD_PART=1"]), ("fa_bwd.hip", "fa_bwd_dq.o", ["-DFA_BWD_PART=2"]), ("fa_bwd.hip", "fa_bwd_fused.o", ["-DFA_BWD_PART=3"]),
             ("fa_bwd_w64.hip", "fa_bwd_w64.o", ["-fno-slp-vectorize"]), ("fa_bwd_dkdv_w64.hip", "fa_bwd_dkdv_w64.o", ["-fno-slp-vectorize", "-DFA_DKDV64_PART=1"]), ("fa_bwd_dkdv_w64.hip", "fa_bwd_c5.o", ["-fno-slp-vectorize", "-DFA_DKDV64_PART=2"]),
             ("fa_api.cpp", "fa_api.o", [])]
    hdrs += _sources(["fa_w64_asm.h", "fa_fwd_w64_regs.h"])
    objs, cmds = [], []
    for u, o, extra in units:
        src = os.path.join(CSRC, u)
        obj = os.path.join(CSRC, o)
        if force or not _newer(obj, [src] + hdrs):
            cmd = [HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC"] + os.environ.get("FA_EXTRA_HIPCC_FLAGS", "").split() + extra + ["-c", src, "-o", obj]
            if u.endswith(".cpp"):
                cmd.insert(1, "-x")
                cmd.insert(2, "hip")
            cmds.append(cmd)
        objs.append(obj)
    if cmds:  # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=len(cmds)) as ex:
            list(ex.map(_run, cmds))
    if force or not _newer(LIB, objs):
        _run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


def build_probe(force=False):
    src = os.path.join(CSRC, "probe_gfx950.hip")
    if force or not _newer(PROBE, [src]):
        _run([HIPCC, f"--offload-arch={ARCH}", "-O2", src, "-o", PROBE])
    return PROBE


def build_torch_ext(force=False):
    """Compile csrc/torch_binding.cpp against the installed torch with g++ (no hipify, no JIT)."""
    import torch
    from torch.utils import cpp_extension as ce

    src = os.path.join(CSRC, "torch_binding.cpp")
    out = ext_path()
    if not force and _newer(out, [src, LIB, os.path.join(ROOT, "include", "fa_gfx950.h")]):
        return out
    inc = []
    for p in ce.include_paths():
        inc += ["-isystem", p]
    inc += ["-isystem", sysconfig.get_paths()["include"], "-isystem", "/opt/rocm/include", "-I", os.path.join(ROOT, "include")]
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={abi}",
           "-DTORCH_EXTENSION_NAME=flash_attn_2_cuda", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-Wno-deprecated-declarations"] + inc + [src, "-o", out,
           f"-L{libdir}", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", "-ltorch_python",
           f"-L{HERE}", "-lfa_gfx950", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{libdir}"]
    _run(cmd)
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-torch-ext", action="store_true")
    ap.add_argument("--force", action="store_true")
    a = ap.parse_args(argv)
    build_lib(a.force)
    build_probe(a.force)
    if not a.no_torch_ext and os.path.exists(os.path.join(CSRC, "torch_binding.cpp")):
        build_torch_ext(a.force)
    print("build ok")


if __name__ == "__main__":
    main()
