#!/bin/bash
# Variants of the 5-contraction backward's translation unit for A/B runs (the switches live in experiments/ablations/fa_bwd_dkdv_w64.patch: FA_C5_ABL_NOSTORE,
# FA_C5_ABL_VMCNT, FA_C5_ABL_SAMEADDR, FA_DS_ST_MOD, FA_DS_LD_MOD):  tools/ab_c5_build.sh name1 'flags1' name2 'flags2' ...  -> gpurun_abl/libfa_c5_<name>.so
set -e
cd "$(dirname "$0")/.."
. tools/ablate_common.sh
SRC=$(abl_source fa_bwd_dkdv_w64.hip)
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( eval $HIPCC -fno-slp-vectorize -DFA_DKDV64_PART=2 $flags -c $SRC -o gpurun_abl/c5_$name.o && \
    abl_link gpurun_abl/libfa_c5_$name.so fa_bwd_c5.o gpurun_abl/c5_$name.o && rm gpurun_abl/c5_$name.o && echo built $name ) &
done
wait
