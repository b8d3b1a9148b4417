#!/bin/bash
# Round-3 final evidence pass: bench line (with PMC traffic and CPU baseline), kernel stats of the bench command, in-kernel clock stamps,
# SQ counters of the forward at config 3, the BASELINE configs, the full GPU test suite.  Output: gpurun_out/r03z/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03z; mkdir -p $O; cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/kt -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-traffic > $O/bench_line_profiled.json 2> $O/bench_prof_err.txt )
python tools/rocpd_summary.py $O/kt/p_results.db > $O/bench_kernel_stats.txt 2>&1
FA_GFX950_LIB=$R/gpurun_abl/libfa_s_new.so python tools/w64_stamps.py > $O/w64_stamps.txt 2>&1
REPS=1 MASKS="" VARIANTS="c_new:" bash tools/ablate_w64.sh run > $O/w64_clk.txt 2>&1
bash tools/pmc_fwd.sh r03z_c3 4 4096 32 128 1 > $O/fwd_w64_sq_counters_causal.txt 2>&1
bash tools/pmc_fwd.sh r03z_nc 4 4096 32 128 0 > $O/fwd_w64_sq_counters_noncausal.txt 2>&1
python tools/bench_configs.py > $O/baseline_configs.txt 2>&1
timeout 700 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu.txt 2>&1
tail -4 $O/pytest_gpu.txt; cut -c1-700 $O/bench_line.json; cat $O/baseline_configs.txt
rm -rf $O/kt $R/gpurun_out/pmc_r03z_c3/p? $R/gpurun_out/pmc_r03z_nc/p?
