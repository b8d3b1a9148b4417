#!/bin/bash
# Variants of the 5-contraction backward's translation unit for A/B runs: tools/ab_c5_build.sh name1 'flags1' name2 'flags2' ...  -> gpurun_abl/libfa_c5_<name>.so
. tools/ablate_common.sh
mkdir -p gpurun_abl
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( eval $HIPCC -fno-slp-vectorize -DFA_DKDV64_PART=2 $flags -c $PKG/csrc/fa_bwd_dkdv_w64.hip -o gpurun_abl/c5_$name.o && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_abl/libfa_c5_$name.so $(for o in $ALL_OBJS; do echo $PKG/csrc/$o; done) gpurun_abl/c5_$name.o && echo built $name ) &
done
wait
