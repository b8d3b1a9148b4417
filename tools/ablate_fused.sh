#!/bin/bash
# Timing ablations of the fused backward (FA_BWD_MODE=3): builds libfa_gfx950 variants with -DFA_FZ_ABL=<mask> into gpurun_abl/ (run here), then
# `tools/ablate_fused.sh run` on the GPU box times the backward with each (results of the ablated builds are wrong by construction).
set -e
cd "$(dirname "$0")/.."
. tools/ablate_common.sh
MASKS="${MASKS:-0 1 2 4 3}"
if [ "$1" != "run" ]; then
  SRC=$(abl_source fa_bwd.hip)
  for m in $MASKS; do
    ( $HIPCC -DFA_BWD_PART=3 -DFA_FZ_ABL=$m $EXTRA -c $SRC -o gpurun_abl/fz_$m.o &&
      abl_link gpurun_abl/libfa_fz_$m.so fa_bwd_fused.o gpurun_abl/fz_$m.o && rm gpurun_abl/fz_$m.o ) &
  done
  wait
  ls -la gpurun_abl
else
  for m in $MASKS; do
    echo "== FA_FZ_ABL=$m"
    FA_GFX950_LIB=$PWD/gpurun_abl/libfa_fz_$m.so python tools/bwd_fused_check.py ${FZ_ARGS:---time-only} 2>&1 | grep "^bwd\|^stats"
  done
fi
