#!/bin/bash
# HBM traffic of the backward kernels at config 3 (causal): FETCH_SIZE and WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md:
# FETCH_SIZE is in KiB and under-reports by 2x on gfx950).  usage: tools/pmc_bwd_traffic.sh <tag> [lib.so]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcb_$1; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
[ -n "$2" ] && export FA_GFX950_LIB=$2
export BW_SHAPES="4,4096,32,1" FA_BWD_MODE=1
i=0
for P in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do i=$((i+1)); rocprofv3 --kernel-trace --pmc $P -d $O/p$i -o p -- python $R/tools/bw64_time.py > $O/log$i.txt 2>&1; done
cd $R; for i in 1 2 3; do python tools/rocpd_summary.py $O/p$i/p_results.db | grep -A20 "mean counter" | grep "bwd" | awk '{print $1, $2, $3}' | sed 's/^[^ ]*fa_bwd/fa_bwd/'; done
