#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O; cd $R
REPS=1 MASKS="" VARIANTS="c_base:;c_1:;c_2:;c_8:;c_16:;c_32:;c_128:;c_160:;c_512:;c_ah1:;c_ah3:;c_ah4:;c_dg1:;c_base:" bash tools/ablate_w64.sh run > $O/w64_abl_clk.txt 2>&1
cat $O/w64_abl_clk.txt
