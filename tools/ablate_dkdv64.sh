#!/bin/bash
# Variants of the 64-keys-per-wave dK/dV kernel (fa_bwd_dkdv64.hip): VARIANTS="name:flags ..." are built into gpurun_abl/ here,
# `tools/ablate_dkdv64.sh run` on the GPU box reports the kernel's duration under rocprofv3 for each (ablated builds compute wrong results).
set -e
cd "$(dirname "$0")/.."
PKG=flash-attention_amd
VARIANTS="${VARIANTS:-base:}"
if [ "$1" != "run" ]; then
  mkdir -p gpurun_abl
  for v in $VARIANTS; do
    name=${v%%:*}; flags=$(echo "${v#*:}" | tr ',' ' ')
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $flags -c $PKG/csrc/fa_bwd_dkdv64.hip -o gpurun_abl/dk64_$name.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Spill|Scratch" | sort | uniq -c | tr '\n' ' '; echo " <- $name";
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_abl/libfa_dk64_$name.so $PKG/csrc/fa_fwd_bf16.o $PKG/csrc/fa_fwd_f16.o $PKG/csrc/fa_fwd_il.o $PKG/csrc/fa_fwd_w64_bf16.o $PKG/csrc/fa_fwd_w64_f16.o $PKG/csrc/fa_bwd_dkdv.o $PKG/csrc/fa_bwd_dq.o $PKG/csrc/fa_bwd_w64.o gpurun_abl/dk64_$name.o $PKG/csrc/fa_api.o && rm gpurun_abl/dk64_$name.o ) &
  done
  wait
  ls gpurun_abl
else
  export TMPDIR=/tmp
  for v in $VARIANTS; do
    name=${v%%:*}
    rm -rf /tmp/dk64_$name
    FA_GFX950_LIB=$PWD/gpurun_abl/libfa_dk64_$name.so FA_BWD_DKDV=64 FA_BWD_MODE=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dk64_$name -o r -- python tools/bw64_time.py > /dev/null 2>&1 || true
    echo "$name: $(python tools/kstats.py /tmp/dk64_$name dkdv | tr '\n' ' ')"
  done
fi
