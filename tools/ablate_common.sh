# Sourced by tools/ablate_*.sh.  The product sources carry no timing-ablation switches: each script first writes the kernel file WITH its switches (the product
# file + experiments/ablations/<file>.patch) to gpurun_abl/src/ and compiles variants of that copy with -D flags; the tree itself is not touched.
PKG=flash-attention_amd
HIPCC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $PKG/csrc -I include"
ALL_OBJS="fa_fwd_bf16.o fa_fwd_f16.o fa_fwd_il.o fa_fwd_w64_bf16.o fa_fwd_w64_f16.o fa_bwd_dkdv.o fa_bwd_dq.o fa_bwd_fused.o fa_bwd_w64.o fa_bwd_dkdv_w64.o fa_bwd_c5.o fa_api.o"

# abl_source fa_bwd_w64.hip -> gpurun_abl/src/fa_bwd_w64.hip (prints the path)
abl_source() {
  mkdir -p gpurun_abl/src
  patch -s -o gpurun_abl/src/$1 $PKG/csrc/$1 < experiments/ablations/${1%.hip}.patch
  echo gpurun_abl/src/$1
}

# abl_link out.so "replaced.o replaced2.o" new.o [new2.o ...]: the product objects minus the replaced ones, plus the new ones
abl_link() {
  local out=$1 skip=" $2 "; shift 2
  local objs=""
  for o in $ALL_OBJS; do case "$skip" in *" $o "*) ;; *) objs="$objs $PKG/csrc/$o";; esac; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs "$@"
}
