#!/bin/bash
# per-kernel backward times (rocprofv3 kernel trace) for config 3 shapes; usage: tools/prof_bwd.sh <tag>
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pb_$1; cd /tmp && export TMPDIR=/tmp
for c in 0 1; do rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pb_$1/c$c -o p -- python $R/tools/run_kernels.py bwd $c 4096 5 > /dev/null 2>&1; done
cd $R; for c in 0 1; do echo "causal=$c"; python tools/rocpd_summary.py gpurun_out/pb_$1/c$c/p_results.db | sed -n 2,4p | awk '{print substr($1,9,34), $2, $3}'; done
