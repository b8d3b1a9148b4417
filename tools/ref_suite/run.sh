#!/bin/bash
# GPU box: the reference's own acceptance suites on OUR flash_attn_2_cuda (sampled -- see conftest.py), four shards side by side.
# Usage: tools/ref_suite/run.sh [per_fn]      results: gpurun_out/ref_suite/{ck,cuda}_*.jsonl + logs; summary by tools/ref_suite/summarize.py
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
T=$ROOT/_ref_tmp
OUT=$ROOT/gpurun_out/ref_suite
mkdir -p $OUT; rm -f $OUT/*.jsonl $OUT/*.log
export PYTHONPATH=$ROOT/flash-attention_amd:$T:$ROOT
export REF_SUITE_PER_FN=${1:-250}
cd $T/tests
shard() {  # name file funcs [deselect-regex]
  REF_SUITE_OUT=$OUT/$1.jsonl REF_SUITE_FUNCS=$3 REF_SUITE_DESELECT=$4 timeout ${REF_SUITE_SHARD_TIMEOUT:-1500} python -m pytest $2 -q -p no:cacheprovider --timeout 300 > $OUT/$1.log 2>&1
  echo "shard $1 rc=$?" >> $OUT/rc.log
}
CK=test_flash_attn_ck.py; CU=test_flash_attn.py
# ids of test_flash_attn.py cases with dropout_p = 0.17: first parameter of the qkvpacked tests, second (after softcap) of the output tests
NA='\[([0-9.]+-)?0\.17-'
shard ck_a $CK test_flash_attn_qkvpacked,test_flash_attn_varlen_qkvpacked,test_flash_attn_causal,test_flash_attn_varlen_causal,test_flash_attn_bwd_overflow,test_flash_attn_bwd_transpose,test_flash_attn_bwd_varlen_overflow,test_flash_attn_bwd_varlen_seqq_zero &
shard ck_b $CK test_flash_attn_output,test_flash_attn_varlen_output,test_flash_attn_race_condition,test_flash_attn_deterministic,test_flash_attn_varlen_deterministic &
shard ck_c $CK test_flash_attn_kvcache &
# test_flash_attn.py: its dropout cases decode the CUDA kernels' S_dmask register layout (convert_flash_attn_S_to_softmax), which a ROCm backend does not
# return (csrc/flash_attn_ck/mha_fwd.cpp:275-279: uint8 random bytes) -- deselected by id; everything else runs
shard cuda_a $CU test_flash_attn_qkvpacked,test_flash_attn_varlen_qkvpacked,test_flash_attn_output,test_flash_attn_varlen_output,test_flash_attn_causal,test_flash_attn_varlen_causal,test_flash_attn_bwd_overflow,test_flash_attn_bwd_transpose,test_flash_attn_bwd_varlen_overflow,test_flash_attn_generator_arg_must_be_none "$NA" &
wait
shard cuda_b $CU test_flash_attn_splitkv,test_flash_attn_race_condition,test_flash_attn_deterministic,test_flash_attn_varlen_deterministic,test_flash_attn_kvcache_paged_block_table_bounds,test_flash_attn_varlen_paged_kv_num_splits "$NA" &
shard cuda_c $CU test_flash_attn_kvcache &
wait
cd $ROOT
# the five literal drop-in tests of this repository (the reference's package / modules on our module, against the fp64 oracle)
FLASH_ATTN_REF=$T timeout 600 python -m pytest tests/test_dropin_reference_gpu.py -v -p no:cacheprovider > $OUT/dropin_tests.log 2>&1
grep -E "PASSED|FAILED|SKIPPED|passed|failed" $OUT/dropin_tests.log | tail -8
python tools/ref_suite/summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
