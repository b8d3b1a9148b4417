#!/bin/bash
# Timing ablations of the 64-rows-per-wave dQ kernel: builds libfa_gfx950 variants with -DFA_BW64_ABL=<mask> into gpurun_abl/
# (run here), then `tools/ablate_bw64.sh run` on the GPU box reports the kernel's average duration under rocprofv3 for each
# (results of the ablated builds are wrong by construction).
set -e
cd "$(dirname "$0")/.."
. tools/ablate_common.sh
MASKS="${MASKS:-0 1 2 4 8 16 31}"
if [ "$1" != "run" ]; then
  SRC=$(abl_source fa_bwd_w64.hip)
  for m in $MASKS; do
    ( $HIPCC -fno-slp-vectorize -DFA_BW64_ABL=$m $EXTRA -c $SRC -o gpurun_abl/bw64_$m.o &&
      abl_link gpurun_abl/libfa_bwabl_$m.so fa_bwd_w64.o gpurun_abl/bw64_$m.o && rm gpurun_abl/bw64_$m.o ) &
  done
  wait
  ls -la gpurun_abl
else
  export TMPDIR=/tmp
  for m in $MASKS; do
    rm -rf /tmp/bwabl_$m
    FA_GFX950_LIB=$PWD/gpurun_abl/libfa_bwabl_$m.so FA_BWD_DQ_NW=64 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bwabl_$m -o r -- python tools/bw64_time.py > /dev/null 2>&1 || true
    echo "ABL=$m: $(python tools/kstats.py /tmp/bwabl_$m | tr '\n' ' ')"
  done
fi
