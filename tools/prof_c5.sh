#!/bin/bash
# per-launch durations of the 5-contraction backward (rocprofv3 kernel trace), config 3 shape; usage: tools/prof_c5.sh <tag> [cap_mb ...]
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/c5_$1; cd /tmp && export TMPDIR=/tmp
tag=$1; shift
for cap in "$@"; do
  for c in ${C5_CAUSAL:-1 0}; do
    FA_BWD_MODE=5 FA_BWD_C5_CAP_MB=$cap rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c5_$tag/cap${cap}_c$c -o p -- python $R/tools/run_kernels.py bwd $c 4096 4 > /dev/null 2>&1
    echo "cap=$cap causal=$c"
    python - <<PY
import sqlite3, glob, re
db = glob.glob("$R/gpurun_out/c5_$tag/cap${cap}_c$c/**/p_results.db", recursive=True)[0]
c = sqlite3.connect(db)
cols = [d[1] for d in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, start, end from kernels order by start").fetchall()
rows = [(re.sub(r"\(.*", "", n)[-40:], s, e) for n, s, e in rows if "fa" in n and "fwd" not in n]
n = len(rows) // 4
last = rows[-n:] if n else rows
t0 = last[0][1]
for nm, s, e in last: print(f"  {nm:40s} start {(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:8.1f} us")
print(f"  span {(last[-1][2] - t0) / 1e3:.1f} us")
PY
  done
done
