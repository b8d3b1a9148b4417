"""Static resource check of the built library, no GPU: the code objects inside libfa_gfx950.so carry per-kernel metadata (registers, spills,
private segment = scratch bytes per lane).  A spill into scratch is not a correctness problem and so no parity test sees it, but a reload waits on
vmcnt(0) -- i.e. on the LDS-DMA prefetch in flight -- and round 3 found the ALiBi dK/dV variant 32 % slower for exactly that reason
(profiles/r03_bwd_alibi_spills.txt).  The rule pinned here: no scratch in any kernel of the plain / softcap / ALiBi variants at head dims <= 128,
none at all in the 64-rows-per-wave and pipelined kernels; the dropout combinations, the run-time-checked all-features variants and head dim 256
may keep the few spills listed in profiles/r03_resource_usage.txt (tools/resource_usage.py prints the same table from the sources)."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "flash-attention_amd", "libfa_gfx950.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FEAT_DROP = 4   # fa_device.h: FEAT bit of dropout; 7 = the run-time-checked "all" variant


def kernel_metadata():
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not os.path.exists(LIB) or not all(os.path.exists(t) for t in tools):
        pytest.skip("library not built or LLVM binutils not found")
    kernels = {}
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([tools[0], f"--dump-section=.hip_fatbin={fat}", LIB, os.path.join(tmp, "copy.so")])
        data = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]   # one bundle per translation unit
        for i, p in enumerate(starts):
            piece, obj = os.path.join(tmp, f"b{i}.bin"), os.path.join(tmp, f"co{i}.o")
            open(piece, "wb").write(data[p:starts[i + 1] if i + 1 < len(starts) else len(data)])
            subprocess.check_call([tools[1], "--unbundle", "--type=o", f"--input={piece}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={obj}"])
            notes = subprocess.run([tools[2], "--notes", obj], capture_output=True, text=True, check=True).stdout
            for block in notes.split("- .agpr_count:")[1:]:
                f = {k: v for k, v in re.findall(r"\.(name|private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|vgpr_count):\s+(\S+)", block)}
                kernels[f["name"]] = {k: int(v) for k, v in f.items() if k != "name"}
    return kernels


def template_ints(name):
    """Integer template arguments of an Itanium-mangled fa:: kernel name, in order."""
    return [int(x) for x in re.findall(r"Li(\d+)E", name)]


def test_no_scratch_in_the_hot_kernels():
    ks = kernel_metadata()
    assert len(ks) >= 250, len(ks)
    fam = lambda s: {n: v for n, v in ks.items() if s in n}
    # the hand-scheduled kernels: never any scratch or vector spill (asm-owned accumulators: a spill there is a miscompile waiting to happen)
    for s in ("fa_fwd_w64_kernel", "fa_bwd_dq_w64_kernel", "fa_bwd_dkdv_w64_kernel", "fa_fwd_il_kernel"):
        assert fam(s), s
        for n, v in fam(s).items():
            assert v["private_segment_fixed_size"] == 0 and v["vgpr_spill_count"] == 0, (n, v)
    assert sum("fa_fwd_w64_kernel" in n and template_ints(n)[-1] == 2 for n in ks) == 4   # causal-ALiBi variants: bf16 / fp16 x D = 64 / 128
    # the lock-step forward and the two backward kernels: <E, D, DV, [NW,] FEAT, ...>
    checked = 0
    for s, feat_at in (("fa_fwd_kernel", 3), ("fa_bwd_dkdv_kernel", 2), ("fa_bwd_dq_kernel", 3)):
        for n, v in fam(s).items():
            ints = template_ints(n)
            d, feat = ints[0], ints[feat_at]
            if d <= 128 and (feat < FEAT_DROP or feat == 8):   # none / softcap / ALiBi / softcap + ALiBi; 8 = FEAT_EXACT, the dK/dV kernel's default plain variant (round 4)
                assert v["private_segment_fixed_size"] == 0, (n, v)
                checked += 1
    assert checked >= 60, checked
    # and the total stays where round 3 left it
    with_scratch = sorted(n for n, v in ks.items() if v["private_segment_fixed_size"] > 0)
    assert len(with_scratch) <= 31, with_scratch   # (round 4: + the two D = 256 FEAT_EXACT dK/dV variants, 6 spills outside the tile loop like their FEAT_NONE twins)
