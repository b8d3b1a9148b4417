#!/bin/bash
# End-of-round evidence pass on one GPU box (usage: gpurun -- 'bash tools/evidence_pass.sh <tag>'): the GPU suite, the bench line (PMC traffic + CPU baseline), kernel stats of
# the same bench command under rocprofv3, in-kernel clock stamps of the forward (if gpurun_abl/libfa_abl_2048.so was built here first), SQ counters of the forward and the
# backward at config 3, and -- when the git-ignored scratch copy _ref_tmp/ travelled along (tools/ref_suite/make_scratch.sh) -- the reference's own suites.
# Output: gpurun_out/<tag>/ ; copy what is to be judged into profiles/.
# (the suite runs serially, as the driver runs it: 4.5 min; under `-n 6` the six processes share one GPU and eight host cores and take 11 min)
TAG=${1:-final}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
( time python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt ) 2> $O/bench_time.txt
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/kt -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-traffic --no-parity --no-sweep > $O/bench_line_profiled.json 2> $O/bench_prof_err.txt )
python tools/rocpd_summary.py $O/kt/p_results.db > $O/bench_kernel_stats.txt 2>&1
head -30 $O/bench_kernel_stats.txt
[ -f $R/gpurun_abl/libfa_abl_2048.so ] && FA_GFX950_LIB=$R/gpurun_abl/libfa_abl_2048.so python tools/w64_stamps.py > $O/w64_stamps.txt 2>&1
bash tools/pmc_fwd.sh ${TAG}_c3 4 4096 32 128 1 > $O/fwd_w64_sq_counters_causal.txt 2>&1
bash tools/pmc_bwd.sh $TAG > /dev/null 2>&1; cp gpurun_out/pmc_bwd_$TAG.txt $O/bwd_sq_counters.txt
[ -d $R/_ref_tmp ] && REF_SUITE_SHARD_TIMEOUT=${REF_SUITE_SHARD_TIMEOUT:-600} bash tools/ref_suite/run.sh ${REF_SUITE_PER_FN:-250} > $O/ref_suite_stdout.txt 2>&1 && cp gpurun_out/ref_suite/summary.txt $O/ref_suite_summary.txt
cut -c1-500 $O/bench_line.json; tail -3 $O/bench_time.txt; tail -12 $O/ref_suite_summary.txt 2>/dev/null | cut -c1-200
rm -rf $O/kt $R/gpurun_out/pmc_${TAG}_c3/p? $R/gpurun_out/pmcb_$TAG
