#!/bin/bash
# Round 3, third session: the scalar-path mixes of tools/issue_bench (which VALU forms share the ~1 per 4 cycles per SIMD
# limit of the scalar unit), then the new plan / RCCL tests.  tools/r3_issue3.sh [tag]
set -u
TAG=${1:-c}
OUT=gpurun_out/r3i_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
tools/issue_bench 20000 3 > $OUT/issue_bench3.log 2>&1 || echo "issue_bench failed"
cat $OUT/issue_bench3.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_plan_run.py tests/test_sharding.py -m gpu -x -q > $OUT/pytest_new.log 2>&1
echo "pytest rc=$?"
tail -30 $OUT/pytest_new.log
