#!/bin/bash
# Build-container only: a git-ignored scratch copy (_ref_tmp/) of the files of the reference that its own acceptance suites need, so that one
# gpurun call can run them on the GPU box (where /root/reference does not exist).  Nothing under _ref_tmp/ is committed; remove it after the run.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
R=${FLASH_ATTN_REF:-/root/reference}
T=$ROOT/_ref_tmp
rm -rf "$T"
mkdir -p $T/flash_attn/utils $T/flash_attn/layers $T/flash_attn/ops/triton $T/flash_attn/modules $T/tests
cp $R/flash_attn/__init__.py $R/flash_attn/flash_attn_interface.py $R/flash_attn/bert_padding.py $T/flash_attn/
cp $R/flash_attn/layers/__init__.py $R/flash_attn/layers/rotary.py $T/flash_attn/layers/
cp $R/flash_attn/ops/__init__.py $T/flash_attn/ops/
cp $R/flash_attn/ops/triton/__init__.py $R/flash_attn/ops/triton/rotary.py $T/flash_attn/ops/triton/
cp $R/flash_attn/utils/__init__.py $R/flash_attn/utils/distributed.py $T/flash_attn/utils/
cp $R/flash_attn/modules/__init__.py $R/flash_attn/modules/mha.py $T/flash_attn/modules/
cp $R/tests/test_util.py $R/tests/test_flash_attn_ck.py $R/tests/test_flash_attn.py $T/tests/
cp $ROOT/tools/ref_suite/conftest.py $T/tests/conftest.py
echo "scratch copy at $T ($(find $T -type f | wc -l) files)"
