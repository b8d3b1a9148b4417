#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03g; mkdir -p $O; cd $R
timeout 400 python -m pytest tests/test_bwd_gpu.py tests/test_bwd_schedules_gpu.py tests/test_baseline_configs_gpu.py tests/test_interface_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -5 > $O/pytest.txt
python tools/bench_configs.py > $O/baseline_configs.txt 2>&1
python tools/bw64_time.py > $O/bw64_time.txt 2>&1
cat $O/pytest.txt $O/baseline_configs.txt; tail -12 $O/bw64_time.txt
