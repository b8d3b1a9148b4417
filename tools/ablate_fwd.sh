#!/bin/bash
# Build timing-ablation variants of the pipelined forward (FA_ABL=n removes one strand of the steady-state step;
# results are numerically wrong, timing only) and time each on the GPU box.  usage: tools/ablate_fwd.sh build|run
set -e
cd "$(dirname "$0")/.."
C=flash-attention_amd/csrc
if [ "$1" = build ]; then
  for n in 0 1 2 3 4 5 6 7 8 9; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFA_ABL=$n -c $C/fa_fwd_il.hip -o /tmp/fa_fwd_il_abl$n.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_abl/libfa_abl$n.so $C/fa_fwd_bf16.o $C/fa_fwd_f16.o $C/fa_fwd_w64_bf16.o $C/fa_fwd_w64_f16.o /tmp/fa_fwd_il_abl$n.o $C/fa_bwd_dkdv.o $C/fa_bwd_dq.o $C/fa_bwd_w64.o $C/fa_bwd_dkdv64.o $C/fa_api.o
  done
else
  for n in 0 1 2 3 4 5 6 7 8 9; do
    echo "ABL=$n"; FA_GFX950_LIB=$PWD/gpurun_abl/libfa_abl$n.so python tools/ab_bench.py 38:8 2>&1 | grep fwd | sed -n '2p;4p'
  done
fi
