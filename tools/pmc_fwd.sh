#!/bin/bash
# SQ counters of the forward kernel; usage: [REPS=n] tools/pmc_fwd.sh <tag> B S H D causal   (env FA_FWD_NW / FA_IL_SCHED select the schedule)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$1; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
P3="SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do i=$((i+1)); rocprofv3 --kernel-trace --pmc $P -d $O/p$i -o p -- python $R/tools/run_fwd_only.py $2 $3 $4 $5 $6 ${REPS-3} > $O/log$i.txt 2>&1; done
cd $R; for i in 1 2 3; do python tools/rocpd_summary.py $O/p$i/p_results.db | grep -v "^$" | awk '{print $1, $2, $3}' | sed 's/^[^ ]*fa_fwd/fa_fwd/' ; done
