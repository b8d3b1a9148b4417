#!/bin/bash
# SQ counters of the backward kernels at config 3 (B=4 H=32 S=4096 D=128 bf16) causal and non-causal: three rocprofv3 --kernel-trace --pmc
# passes (8 counters each, kernel-trace only -- no other trace domain next to --pmc) over 3 backward calls, mean per dispatch and kernel.
# usage (GPU box): tools/pmc_bwd.sh <tag>   -> gpurun_out/pmc_bwd_<tag>.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcb_$1; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
P3="SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
for c in 1 0; do
  i=0
  for P in "$P1" "$P2" "$P3"; do i=$((i+1)); rocprofv3 --kernel-trace --pmc $P -d $O/c${c}p$i -o p -- python $R/tools/run_kernels.py bwd $c 4096 3 > $O/log_c${c}_$i.txt 2>&1; done
done
cd $R
{
  for c in 1 0; do
    echo "## causal=$c"
    for i in 1 2 3; do python tools/rocpd_summary.py $O/c${c}p$i/p_results.db | sed -n '/mean counter value/,$p' | grep "fa_bwd" | sed 's/^.*fa::\(fa_bwd[a-z_0-9]*\)[^ ]*/\1/' | awk '{printf "%-28s %-28s %16.0f\n", $1, $2, $3}'; done
  done
} > gpurun_out/pmc_bwd_$1.txt
cat gpurun_out/pmc_bwd_$1.txt | head -100
