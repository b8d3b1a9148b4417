#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03d; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_fwd_gpu.py tests/test_baseline_configs_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -8 > $O/pytest.txt
python tools/w64_time.py > $O/time_new.txt 2>&1
REPS=1 MASKS="" VARIANTS="c_new:" bash tools/ablate_w64.sh run > $O/clk_new.txt 2>&1
FA_GFX950_LIB=$R/gpurun_abl/libfa_s_new.so python tools/w64_stamps.py > $O/w64_stamps_new.txt 2>&1
python bench.py --no-cpu --no-traffic --no-sweep > $O/bench.json 2>&1
cat $O/pytest.txt; cut -c1-60 $O/time_new.txt; cat $O/clk_new.txt; grep -A18 "m_block  n_it" $O/w64_stamps_new.txt | head -20; cut -c1-400 $O/bench.json
