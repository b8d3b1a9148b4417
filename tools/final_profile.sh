#!/bin/bash
# Kernel-time stats of the bench command and HBM traffic counters of the config-3 kernels -> gpurun_out/final_profile.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o p -- python $R/bench.py --no-cpu > $O/bench_line.txt 2>$O/bench_err.txt
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $C | tr ' ' '_'); rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$n -o p -- python $R/tools/run_kernels.py all 1 4096 3 > /dev/null 2>&1
done
cd $R
{
  echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu   (MI355X; profiled runs clock ~5% lower than unprofiled)"
  python tools/rocpd_summary.py $O/kt/p_results.db | head -8
  echo; echo "# bench.py line of the same run:"; tail -1 $O/bench_line.txt
  echo; echo "# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum (separate passes) -- python tools/run_kernels.py all 1 4096 3   (config 3)"
  for C in FETCH_SIZE WRITE_SIZE TCC_HIT_sum_TCC_MISS_sum; do python tools/rocpd_summary.py $O/pmc_$C/p_results.db | grep "_ZN2fa.*\(FETCH\|WRITE\|TCC\)"; done
} > gpurun_out/final_profile.txt
cat gpurun_out/final_profile.txt
