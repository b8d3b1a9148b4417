#!/bin/bash
# per-launch durations of the default (recomputing) backward on the same box, config 3 shape; usage: tools/prof_pair.sh <tag>
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pair_$1; cd /tmp && export TMPDIR=/tmp
for c in ${C5_CAUSAL:-1 0}; do
  FA_BWD_MODE=-1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pair_$1/c$c -o p -- python $R/tools/run_kernels.py bwd $c 4096 4 > /dev/null 2>&1
  echo "pair causal=$c"; python $R/tools/rocpd_summary.py $(find $R/gpurun_out/pair_$1/c$c -name p_results.db) | sed -n 2,4p | awk '{print "  " substr($1,1,60), $2, $3}'
done
