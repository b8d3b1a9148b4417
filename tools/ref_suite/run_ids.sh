#!/bin/bash
# GPU box: run the reference-suite cases listed in a file (one pytest node id per line, relative to _ref_tmp/tests) on our module, with whatever FA_* knobs the
# environment carries.  usage: tools/ref_suite/run_ids.sh ids.txt
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
T=$ROOT/_ref_tmp
export PYTHONPATH=$ROOT/flash-attention_amd:$T:$ROOT
export REF_SUITE_PER_FN=100000000 REF_SUITE_OUT=/tmp/run_ids.jsonl
IDS=$(cd $ROOT && cat "$1" | tr '\n' ' ')
FUNCS=$(cat "$ROOT/$1" | sed 's/.*::\(test_[a-z_]*\).*/\1/' | sort -u | tr '\n' ',')
cd $T/tests
REF_SUITE_FUNCS=$FUNCS python -m pytest $(cat "$ROOT/$1" | while read l; do echo "$l"; done | tr '\n' ' ') -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|AssertionError: assert" | cut -c1-220
