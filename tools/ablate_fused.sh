#!/bin/bash
# Timing ablations of the fused backward (FA_BWD_MODE=3): builds libfa_gfx950 variants with -DFA_FZ_ABL=<mask> into gpurun_abl/ (run here), then
# `tools/ablate_fused.sh run` on the GPU box times the backward with each (results of the ablated builds are wrong by construction).
set -e
cd "$(dirname "$0")/.."
PKG=flash-attention_amd
MASKS="${MASKS:-0 1 2 4 3}"
if [ "$1" != "run" ]; then
  mkdir -p gpurun_abl
  for m in $MASKS; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFA_BWD_PART=3 -DFA_FZ_ABL=$m $EXTRA -c $PKG/csrc/fa_bwd.hip -o gpurun_abl/fz_$m.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_abl/libfa_fz_$m.so $PKG/csrc/fa_fwd_bf16.o $PKG/csrc/fa_fwd_f16.o $PKG/csrc/fa_fwd_il.o $PKG/csrc/fa_fwd_w64_bf16.o $PKG/csrc/fa_fwd_w64_f16.o $PKG/csrc/fa_bwd_dkdv.o $PKG/csrc/fa_bwd_dq.o $PKG/csrc/fa_bwd_w64.o gpurun_abl/fz_$m.o $PKG/csrc/fa_api.o && rm gpurun_abl/fz_$m.o ) &
  done
  wait
  ls -la gpurun_abl
else
  for m in $MASKS; do
    echo "== FA_FZ_ABL=$m"
    FA_GFX950_LIB=$PWD/gpurun_abl/libfa_fz_$m.so python tools/bwd_fused_check.py ${FZ_ARGS:---time-only} 2>&1 | grep "^bwd\|^stats"
  done
fi
