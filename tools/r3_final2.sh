#!/bin/bash
# round-3 session after the resident-entry kernel went in: whole GPU suite, fuzz, pack-kernel A/B (the final-set code moved
# into a shared function), bisect of the rule-set workload over the round's kernel commits
OUT=gpurun_out/r3_final2; mkdir -p $OUT
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 $OUT/pytest.log
[ $rc = 0 ] || exit 1
timeout -k 10 260 python3 tools/fuzz_gpu.py --seconds 200 --seed 515 > $OUT/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz.log
echo "== rule set, 65536 x 1 KB, round's kernel commits"; tools/r3_bisect.sh R --steps 6 > /dev/null; cat gpurun_out/r3_bisect_R.log
