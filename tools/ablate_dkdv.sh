#!/bin/bash
# Timing ablations of the dK/dV kernel: builds libfa_gfx950 variants with -DFA_DKDV_ABL=<mask> into gpurun_abl/ (run here),
# then `tools/ablate_dkdv.sh run` on the GPU box reports the kernel's duration under rocprofv3 for each
# (results of the ablated builds are wrong by construction).
set -e
cd "$(dirname "$0")/.."
PKG=flash-attention_amd
MASKS="${MASKS:-0 1 2 4 8 16 32 64}"
if [ "$1" != "run" ]; then
  mkdir -p gpurun_abl
  for m in $MASKS; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFA_DKDV_ABL=$m $EXTRA -c $PKG/csrc/fa_bwd.hip -o gpurun_abl/bwd_$m.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_abl/libfa_dkdv_$m.so $PKG/csrc/fa_fwd_bf16.o $PKG/csrc/fa_fwd_f16.o $PKG/csrc/fa_fwd_il.o $PKG/csrc/fa_fwd_w64_bf16.o $PKG/csrc/fa_fwd_w64_f16.o gpurun_abl/bwd_$m.o $PKG/csrc/fa_bwd_w64.o $PKG/csrc/fa_bwd_dkdv64.o $PKG/csrc/fa_api.o && rm gpurun_abl/bwd_$m.o ) &
  done
  wait
  ls gpurun_abl
else
  export TMPDIR=/tmp
  for m in $MASKS; do
    rm -rf /tmp/dkdv_$m
    FA_GFX950_LIB=$PWD/gpurun_abl/libfa_dkdv_$m.so FA_BWD_MODE=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dkdv_$m -o r -- python tools/bw64_time.py > /dev/null 2>&1 || true
    echo "ABL=$m: $(python tools/kstats.py /tmp/dkdv_$m dkdv | tr '\n' ' ')"
  done
fi
