#!/bin/bash
# Timing ablations of the 64-rows-per-wave forward: builds libfa_gfx950 variants with -DFA_W64_ABL=<mask> into gpurun_abl/
# (run here), then `tools/ablate_w64.sh run` on the GPU box times each (results of the ablated builds are wrong by construction).
set -e
cd "$(dirname "$0")/.."
PKG=flash-attention_amd
MASKS="${MASKS:-0 1 2 4 8 16 32 64 3 15 31 63}"   # 256: clocks per MFMA (tools/w64_time.py), 2048: per-iteration stamps (tools/w64_stamps.py)
if [ "$1" != "run" ]; then
  mkdir -p gpurun_abl
  for m in $MASKS; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DFA_W64_ABL=$m -c $PKG/csrc/fa_fwd_w64.hip -o gpurun_abl/w64_$m.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_abl/libfa_abl_$m.so $PKG/csrc/fa_fwd_bf16.o $PKG/csrc/fa_fwd_f16.o $PKG/csrc/fa_fwd_il.o gpurun_abl/w64_$m.o $PKG/csrc/fa_bwd_dkdv.o $PKG/csrc/fa_bwd_dq.o $PKG/csrc/fa_bwd_w64.o $PKG/csrc/fa_bwd_dkdv64.o $PKG/csrc/fa_api.o && rm gpurun_abl/w64_$m.o ) &
  done
  wait
  ls -la gpurun_abl
else
  for m in $MASKS; do
    echo "ABL=$m: $(FA_GFX950_LIB=$PWD/gpurun_abl/libfa_abl_$m.so python tools/w64_time.py 2>/dev/null | tr '\n' ' ')"
  done
fi
