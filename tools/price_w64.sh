#!/bin/bash
# What softcap / dropout / exact fp32 score scaling would cost ON the 64-rows-per-wave forward: experiments/fa_fwd_w64_price.patch adds the
# extra per-score instructions of each feature to the hand-placed step with neutral constants (the results stay those of plain attention); this script builds the
# priced variants into gpurun_abl/ (here), `tools/price_w64.sh run` times them on the GPU box (tools/w64_time.py) -> profiles/r05_feat_pricing.txt.
#   -DFA_W64_PRICE: 1 = one fp32 multiply per score (exact scaling), 2 = softcap (2 transcendentals + 5 VALU), 4 = dropout (Philox2x32-7 per 4 scores + select)
set -e
cd "$(dirname "$0")/.."
PKG=flash-attention_amd
OBJS="fa_fwd_bf16.o fa_fwd_f16.o fa_fwd_il.o fa_bwd_dkdv.o fa_bwd_dq.o fa_bwd_fused.o fa_bwd_w64.o fa_bwd_dkdv_w64.o fa_api.o"
if [ "$1" != "run" ]; then
  mkdir -p gpurun_abl
  patch -s -o gpurun_abl/fa_fwd_w64_price.hip $PKG/csrc/fa_fwd_w64.hip < experiments/fa_fwd_w64_price.patch
  for m in 0 1 2 4; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DFA_W64_PRICE=$m -I $PKG/csrc -I include -c gpurun_abl/fa_fwd_w64_price.hip -o gpurun_abl/price_$m.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "VGPRs Spill|ScratchSize" | sort | uniq -c | tr '\n' ' '; echo " <- price $m";
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_abl/libfa_price_$m.so gpurun_abl/price_$m.o $(for o in $OBJS; do echo $PKG/csrc/$o; done) && rm gpurun_abl/price_$m.o ) &
  done
  wait
else
  for rep in 1 2; do
    for m in 0 1 2 4; do
      echo "price $m: $(FA_GFX950_LIB=$PWD/gpurun_abl/libfa_price_$m.so python tools/w64_time.py 2>/dev/null | cut -c1-40 | tr '\n' ' ')"
    done
  done
fi
