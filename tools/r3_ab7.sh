#!/bin/bash
TAG=${1:-r}; OUT=gpurun_out/r3y_$TAG; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout -k 10 150 python3 tools/fuzz_gpu.py --seconds 100 --seed 61 > $OUT/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz.log
echo "T 65536"; tools/ab.sh 2>/dev/null | tee $OUT/ab_auto.log
echo "U 65536"; tools/ab.sh --workload U 2>/dev/null | tee $OUT/ab_U.log
echo "rule set 65536 x 1 KB"; tools/ab.sh --workload R --steps 8 2>/dev/null | tee $OUT/ab_R.log
tools/profile.sh r03_$TAG > $OUT/profile.log 2>&1; tail -2 $OUT/profile.log
