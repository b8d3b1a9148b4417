#!/bin/bash
# Round-4 second GPU pass: the suite with this round's regression tests, the reference's suites again (full sample), the bench line with its new keys,
# A/Bs: fast diagonal masking, dK/dV score scaling, dQ 64-rows-per-wave at D = 64, the 64-rows-per-wave forward under a window.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $O/pytest_gpu.txt 2>&1
tail -4 $O/pytest_gpu.txt
( time python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt ) 2> $O/bench_time.txt
tail -3 $O/bench_time.txt; cut -c1-600 $O/bench_line.json; tail -3 $O/bench_err.txt
REPS=2 MASKS="" VARIANTS="base:;fd0:" bash tools/ablate_w64.sh run > $O/w64_fastdiag_ab.txt 2>&1
cat $O/w64_fastdiag_ab.txt
for v in base fd0; do echo "== $v"; FA_GFX950_LIB=$R/gpurun_abl/libfa_$v.so python tools/knob_ab.py "default=" --shapes short,cfg3 --fwd; done > $O/w64_fastdiag_shapes.txt 2>&1
cat $O/w64_fastdiag_shapes.txt
FA_GFX950_LIB=$R/gpurun_abl/libfa_abl_2048.so python tools/w64_stamps.py > $O/w64_stamps_fastdiag.txt 2>&1
FA_GFX950_LIB=$R/gpurun_abl/libfa_fd0_2048.so python tools/w64_stamps.py > $O/w64_stamps_general.txt 2>&1
grep -A18 "m_block  n_it" $O/w64_stamps_fastdiag.txt | head -19; grep -A18 "m_block  n_it" $O/w64_stamps_general.txt | head -19
python tools/knob_ab.py "exact=;prescale=FA_DKDV_PRESCALE=1" --shapes cfg3,cfg2 --bwd > $O/bwd_dkdv_scaling_ab.txt 2>&1
cat $O/bwd_dkdv_scaling_ab.txt
python tools/knob_ab.py "dq4=FA_BWD_DQ_NW=4;dq8=FA_BWD_DQ_NW=8;dq64=FA_BWD_DQ_NW=64" --shapes cfg2,d64 --bwd > $O/bwd_dq_w64_d64_ab.txt 2>&1
cat $O/bwd_dq_w64_d64_ab.txt
python tools/knob_ab.py "default=;w64=FA_FWD_NW=64;il4=FA_FWD_NW=34;il8=FA_FWD_NW=38" --shapes cfg5 --fwd > $O/fwd_window_ab.txt 2>&1
cat $O/fwd_window_ab.txt
( cd /tmp && export TMPDIR=/tmp && timeout 120 rocprofv3 --att --kernel-trace -d $O/att -o a -- python $R/tools/run_fwd_only.py 4 4096 32 128 1 > $O/att_attempt.txt 2>&1; echo "rc=$?" >> $O/att_attempt.txt ); tail -5 $O/att_attempt.txt; rm -rf $O/att
REF_SUITE_SHARD_TIMEOUT=600 bash tools/ref_suite/run.sh 250 > $O/ref_suite_stdout.txt 2>&1
head -45 $O/ref_suite_stdout.txt
