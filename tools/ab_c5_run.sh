#!/bin/bash
# tools/ab_c5_run.sh cap name1 name2 ...: per-launch durations (rocprofv3) of each variant library built by tools/ab_c5_build.sh, config 3 shape, causal and not
R=$GRAFT_REPO_ROOT; cap=$1; shift
for name in "$@"; do
  echo "=== $name"
  FA_GFX950_LIB=$R/gpurun_abl/libfa_c5_$name.so $R/tools/prof_c5.sh ab_$name $cap 2>&1 | grep -v "^$"
done
