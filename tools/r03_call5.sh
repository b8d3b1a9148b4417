#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; mkdir -p $O; cd $R
python tools/ab_bench.py 64:8,34:8,38:8 --sweep-shapes > $O/ab_d128.txt 2>&1
python tools/ab_bench.py 64:8,34:8,38:8 --d64-shapes > $O/ab_d64.txt 2>&1
cat $O/ab_d128.txt $O/ab_d64.txt
