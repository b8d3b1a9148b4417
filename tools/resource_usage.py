"""hipcc -Rpass-analysis=kernel-resource-usage over every translation unit of the default library (same flags as flash-attention_amd/build.py):
one line per kernel instantiation -- registers, scalar / vector spills, scratch bytes per lane.  No GPU needed.
Usage: python tools/resource_usage.py [--scratch-only] > profiles/rNN_resource_usage.txt"""
import os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "flash-attention_amd", "csrc")
UNITS = [("fa_fwd.hip", ["-DFA_FWD_PART=1"]), ("fa_fwd.hip", ["-DFA_FWD_PART=2"]), ("fa_fwd_il.hip", []), ("fa_fwd_w64.hip", ["-fno-slp-vectorize", "-DFA_W64_PART=1"]),
         ("fa_fwd_w64.hip", ["-fno-slp-vectorize", "-DFA_W64_PART=2"]), ("fa_bwd.hip", ["-DFA_BWD_PART=1"]), ("fa_bwd.hip", ["-DFA_BWD_PART=2"]), ("fa_bwd.hip", ["-DFA_BWD_PART=3"]),
         ("fa_bwd_w64.hip", ["-fno-slp-vectorize"]), ("fa_bwd_dkdv_w64.hip", ["-fno-slp-vectorize", "-DFA_DKDV64_PART=1"]), ("fa_bwd_dkdv_w64.hip", ["-fno-slp-vectorize", "-DFA_DKDV64_PART=2"])]
def unit(u):
    src, extra = u
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--offload-device-only", "-Rpass-analysis=kernel-resource-usage"] + extra + \
          ["-c", os.path.join(CSRC, src), "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: +(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
        if not m: continue
        if m.group(1) == "Function Name":
            cur = {"name": m.group(2), "unit": src + " " + " ".join(extra)}; rows.append(cur)
        elif cur is not None: cur[m.group(1).split(" [")[0]] = m.group(2)
    return rows
def demangle(names):
    """fa::kernel<bf16|f16, ints / bools ...> from the Itanium names (the image's c++filt does not know DF16b)."""
    out = []
    for n in names:
        m = re.match(r"_ZN2fa\d+([A-Za-z_0-9]+?)I(.*)EEvNS_\d+\w+E$", n)
        if not m:
            out.append(subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n); continue
        args = re.findall(r"DF16b|DF16_|Li\d+E|Lb[01]E", m.group(2))
        dec = [{"DF16b": "bf16", "DF16_": "f16"}.get(a) or (a[2:-1] if a.startswith("Li") else ("true" if a == "Lb1E" else "false")) for a in args]
        out.append(f"fa::{m.group(1)}<{','.join(dec)}>")
    return out
if __name__ == "__main__":
    with ThreadPoolExecutor(max_workers=len(UNITS)) as ex: rows = [r for rs in ex.map(unit, UNITS) for r in rs]
    for r, n in zip(rows, demangle([r["name"] for r in rows])): r["name"] = n
    only = "--scratch-only" in sys.argv
    print(f"# {len(rows)} kernel instantiations; {sum(int(r['ScratchSize']) > 0 for r in rows)} with scratch")
    for r in rows:
        if only and int(r["ScratchSize"]) == 0: continue
        print(f"{r['name']:<72} vgpr {r['VGPRs']:>3} agpr {r['AGPRs']:>3} sgpr {r['TotalSGPRs']:>3} occ {r['Occupancy']} sspill {r['SGPRs Spill']:>3} vspill {r['VGPRs Spill']:>3} scratch {r['ScratchSize']:>4}")
