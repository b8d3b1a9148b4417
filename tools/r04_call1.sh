#!/bin/bash
# Round-4 first GPU pass: the -m gpu suite on the tree as it stands, a sample of the reference's own acceptance suites on our module,
# board power / clocks under the forward with and without its K/V DMA, DMA cache-policy A/B, stamps with and without DMA.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
REF_SUITE_SHARD_TIMEOUT=420 bash tools/ref_suite/run.sh 100 > $O/ref_suite_stdout.txt 2>&1
tail -40 $O/ref_suite_stdout.txt
python tools/power_probe.py 3 > $O/power_default.txt 2>&1
FA_GFX950_LIB=$R/gpurun_abl/libfa_abl_32.so python tools/power_probe.py 3 > $O/power_no_kv_dma.txt 2>&1
cat $O/power_default.txt $O/power_no_kv_dma.txt
REPS=2 MASKS="" VARIANTS="base:;pol1:;pol2:;pol4:;pol6:;abl_32:" bash tools/ablate_w64.sh run > $O/w64_dma_policy.txt 2>&1
cat $O/w64_dma_policy.txt
FA_GFX950_LIB=$R/gpurun_abl/libfa_abl_2048.so python tools/w64_stamps.py > $O/w64_stamps.txt 2>&1
FA_GFX950_LIB=$R/gpurun_abl/libfa_abl_2080.so python tools/w64_stamps.py > $O/w64_stamps_no_kv_dma.txt 2>&1
grep -A20 "m_block  n_it" $O/w64_stamps.txt | head -45
grep -A20 "m_block  n_it" $O/w64_stamps_no_kv_dma.txt | head -45
grep "kernel\|GHz" $O/w64_stamps_no_kv_dma.txt | head -12
