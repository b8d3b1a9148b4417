#!/bin/bash
# LDS-DMA issue cost: the microbenchmark and the wave-staggered slot variants of the w64 forward
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O; cd $R
timeout 120 tools/ubench/dma_issue > $O/dma_issue.txt 2>&1
REPS=2 MASKS="" VARIANTS="base:;stag1:;stag1g2:;stag2:;stag2g2:;c_base:;c_stag1:" bash tools/ablate_w64.sh run > $O/w64_stag.txt 2>&1
FA_GFX950_LIB=$R/gpurun_abl/libfa_stag1.so timeout 200 python -m pytest tests/test_fwd_gpu.py tests/test_baseline_configs_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -3 > $O/pytest_stag1.txt
cat $O/dma_issue.txt $O/w64_stag.txt $O/pytest_stag1.txt
