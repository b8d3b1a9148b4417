#!/bin/bash
# Round-3 evidence pass on one GPU box: bench line, kernel stats of the bench command, in-kernel clock stamps and timing ablations of the
# 64-rows-per-wave forward, SQ counters of the forward at config 3, then the GPU test suite.  Everything lands under gpurun_out/r03a/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/kt -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-traffic > $O/bench_line_profiled.json 2> $O/bench_prof_err.txt )
python tools/rocpd_summary.py $O/kt/p_results.db > $O/bench_kernel_stats.txt 2>&1
FA_GFX950_LIB=$R/gpurun_abl/libfa_abl_2048.so python tools/w64_stamps.py > $O/w64_stamps.txt 2>&1
REPS=1 MASKS="256 257 258 260 264 384 288 768 1280 4352" VARIANTS="" bash tools/ablate_w64.sh run > $O/w64_ablations.txt 2>&1
bash tools/pmc_fwd.sh r03a_c3 4 4096 32 128 1 > $O/fwd_w64_sq_counters_causal.txt 2>&1
bash tools/pmc_fwd.sh r03a_nc 4 4096 32 128 0 > $O/fwd_w64_sq_counters_noncausal.txt 2>&1
python tools/bench_configs.py > $O/baseline_configs.txt 2>&1
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
cat $O/bench_line.json | cut -c1-1500
rm -rf $O/kt $R/gpurun_out/pmc_r03a_c3/p? $R/gpurun_out/pmc_r03a_nc/p?
