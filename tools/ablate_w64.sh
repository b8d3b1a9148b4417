#!/bin/bash
# Timing variants of the 64-rows-per-wave forward: builds libfa_gfx950 variants into gpurun_abl/ (run here), then `tools/ablate_w64.sh run`
# on the GPU box times each with tools/w64_time.py.
#   MASKS="0 1 2 ..."            -DFA_W64_ABL=<mask> ablations (results of ablated builds are wrong by construction); 256: clocks per MFMA
#                                 (tools/w64_time.py), 2048: per-iteration clock stamps (tools/w64_stamps.py)
#   VARIANTS="name:flags;..."    arbitrary -D variants, e.g. "k28:-DFA_W64_KDMA_G0=28 -DFA_W64_KDMA_GS=1"
set -e
cd "$(dirname "$0")/.."
. tools/ablate_common.sh
MASKS="${MASKS-0 1 2 4 8 16 32 64 3 15 31 63}"
LIST=""
for m in $MASKS; do LIST="$LIST;abl_$m:-DFA_W64_ABL=$m"; done
LIST="$LIST;$VARIANTS"

IFS=';'
if [ "$1" != "run" ]; then
  SRC=$(abl_source fa_fwd_w64.hip)
  for v in $LIST; do
    [ -z "$v" ] && continue
    name="${v%%:*}"; flags="${v#*:}"
    ( IFS=' '; $HIPCC -fno-slp-vectorize -DFA_W64_PART=1 $flags -c $SRC -o gpurun_abl/w64_$name.o &&
      abl_link gpurun_abl/libfa_$name.so fa_fwd_w64_bf16.o gpurun_abl/w64_$name.o && rm gpurun_abl/w64_$name.o ) &
  done
  wait
  ls -la gpurun_abl
else
  for rep in $(seq 1 ${REPS-2}); do
    for v in $LIST; do
      [ -z "$v" ] && continue
      name="${v%%:*}"
      echo "$name: $(FA_GFX950_LIB=$PWD/gpurun_abl/libfa_$name.so python tools/w64_time.py 2>/dev/null | cut -c1-110 | sed "s|^|    |")"
    done
  done
fi
