#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
python bench.py --no-cpu --no-traffic > $O/bench.json 2> $O/bench_err.txt
python tools/bench_configs.py > $O/baseline_configs.txt 2>&1
timeout 700 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu.txt 2>&1
tail -15 $O/pytest_gpu.txt; cat $O/baseline_configs.txt; python - <<'P'
import json
d=json.load(open("gpurun_out/r03f/bench.json"))
print(d["value"], d["roofline"]["frac"], d["fwd_bwd"])
for r in d["sweep"]["rows"]: print(r)
P
