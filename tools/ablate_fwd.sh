#!/bin/bash
# Build timing-ablation variants of the pipelined forward (FA_ABL=n removes one strand of the steady-state step;
# results are numerically wrong, timing only) and time each on the GPU box.  usage: tools/ablate_fwd.sh build|run
set -e
cd "$(dirname "$0")/.."
. tools/ablate_common.sh
if [ "$1" = build ]; then
  SRC=$(abl_source fa_fwd_il.hip)
  for n in 0 1 2 3 4 5 6 7 8 9; do
    ( $HIPCC -DFA_ABL=$n -c $SRC -o gpurun_abl/il_abl$n.o && abl_link gpurun_abl/libfa_abl$n.so fa_fwd_il.o gpurun_abl/il_abl$n.o && rm gpurun_abl/il_abl$n.o ) &
  done
  wait
else
  for n in 0 1 2 3 4 5 6 7 8 9; do
    echo "ABL=$n"; FA_GFX950_LIB=$PWD/gpurun_abl/libfa_abl$n.so python tools/ab_bench.py 38:8 2>&1 | grep fwd | sed -n '2p;4p'
  done
fi
