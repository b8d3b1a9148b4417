#!/bin/bash
# What softcap / dropout / exact fp32 score scaling would cost ON the 64-rows-per-wave forward (r05 item 6): experiments/fa_fwd_w64_price.patch adds the
# extra per-score instructions of each feature to the hand-placed step with neutral constants (the results stay those of plain attention); this script builds the
# priced variants into gpurun_abl/ (here), `tools/price_w64.sh run` times them on the GPU box (tools/w64_time.py) -> profiles/r05_feat_pricing.txt, r06_lse_exact_probe.txt.
#   -DFA_W64_PRICE: 1 = one fp32 multiply per score (exact scaling), 2 = softcap (2 transcendentals + 5 VALU), 4 = dropout (Philox2x32-7 per 4 scores + select),
#                   8 = exact scaling as one v_pk_mul_f32 per two scores (round 6)
set -e
cd "$(dirname "$0")/.."
. tools/ablate_common.sh
MODES=${MODES:-0 1 2 4 8}
if [ "$1" != "run" ]; then
  mkdir -p gpurun_abl/src
  patch -s -o gpurun_abl/src/fa_fwd_w64_price.hip $PKG/csrc/fa_fwd_w64.hip < experiments/fa_fwd_w64_price.patch
  for m in $MODES; do
    ( $HIPCC -fno-slp-vectorize -DFA_W64_PART=1 -DFA_W64_PRICE=$m -c gpurun_abl/src/fa_fwd_w64_price.hip -o gpurun_abl/price_$m.o &&
      abl_link gpurun_abl/libfa_price_$m.so fa_fwd_w64_bf16.o gpurun_abl/price_$m.o && rm gpurun_abl/price_$m.o ) &
  done
  wait
else
  for rep in 1 2; do
    for m in $MODES; do
      echo "price $m: $(FA_GFX950_LIB=$PWD/gpurun_abl/libfa_price_$m.so python tools/w64_time.py 2>/dev/null | cut -c1-40 | tr '\n' ' ')"
    done
  done
fi
