#!/bin/bash
# Variants of the 64-keys-per-wave dK/dV kernel (csrc/fa_bwd_dkdv_w64.hip): VARIANTS="name:flag,flag ..." are built into gpurun_abl/ here (hipcc cross-compiles),
# `tools/ablate_dkdv64.sh run` on the GPU box reports the kernel's duration under rocprofv3 for each (builds with -DFA_DKDV64_ABL compute wrong results).
set -e
cd "$(dirname "$0")/.."
. tools/ablate_common.sh
VARIANTS="${VARIANTS:-base:}"
if [ "$1" != "run" ]; then
  SRC=$(abl_source fa_bwd_dkdv_w64.hip)
  for v in $VARIANTS; do
    name=${v%%:*}; flags=$(echo "${v#*:}" | tr ',' ' ')
    ( $HIPCC -fno-slp-vectorize -DFA_DKDV64_PART=1 $flags -c $SRC -o gpurun_abl/dk64_$name.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Spill|ScratchSize" | sort | uniq -c | sed "s/^/$name: /" ;
      abl_link gpurun_abl/libfa_dk64_$name.so fa_bwd_dkdv_w64.o gpurun_abl/dk64_$name.o && rm gpurun_abl/dk64_$name.o ) &
  done
  wait
  ls gpurun_abl
else
  export TMPDIR=/tmp
  for v in $VARIANTS; do
    name=${v%%:*}
    rm -rf /tmp/dk64_$name
    FA_GFX950_LIB=$PWD/gpurun_abl/libfa_dk64_$name.so FA_BWD_DKDV=64 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dk64_$name -o r -- python tools/bw64_time.py > /dev/null 2>&1 || true
    echo "$name: $(python tools/kstats.py /tmp/dk64_$name dkdv | tr '\n' ' ')"
  done
fi
