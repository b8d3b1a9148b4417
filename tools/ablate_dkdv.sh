#!/bin/bash
# Timing ablations of the dK/dV kernel: builds libfa_gfx950 variants with -DFA_DKDV_ABL=<mask> into gpurun_abl/ (run here),
# then `tools/ablate_dkdv.sh run` on the GPU box reports the kernel's duration under rocprofv3 for each
# (results of the ablated builds are wrong by construction).
set -e
cd "$(dirname "$0")/.."
. tools/ablate_common.sh
MASKS="${MASKS:-0 1 2 4 8 16 32 64}"
if [ "$1" != "run" ]; then
  SRC=$(abl_source fa_bwd.hip)
  for m in $MASKS; do
    ( $HIPCC -DFA_BWD_PART=1 -DFA_DKDV_ABL=$m $EXTRA -c $SRC -o gpurun_abl/bwd_$m.o &&
      abl_link gpurun_abl/libfa_dkdv_$m.so fa_bwd_dkdv.o gpurun_abl/bwd_$m.o && rm gpurun_abl/bwd_$m.o ) &
  done
  wait
  ls gpurun_abl
else
  export TMPDIR=/tmp
  for m in $MASKS; do
    rm -rf /tmp/dkdv_$m
    FA_GFX950_LIB=$PWD/gpurun_abl/libfa_dkdv_$m.so FA_BWD_DKDV=8 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dkdv_$m -o r -- python tools/bw64_time.py > /dev/null 2>&1 || true
    echo "ABL=$m: $(python tools/kstats.py /tmp/dkdv_$m dkdv | tr '\n' ' ')"
  done
fi
