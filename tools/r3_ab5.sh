#!/bin/bash
# full GPU suite, then round-2 build vs current on one box over the workloads AUTO serves
TAG=${1:-o}; OUT=gpurun_out/r3w_$TAG; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout -k 10 150 python3 tools/fuzz_gpu.py --seconds 100 --seed 53 > $OUT/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz.log
echo "T 65536"; tools/ab.sh 2>/dev/null | tee $OUT/ab_auto.log
echo "rule set 65536 x 1 KB"; tools/ab.sh --workload R --steps 8 2>/dev/null | tee $OUT/ab_R.log
echo "l7 65536 x 1 KB"; tools/ab.sh --workload L --steps 8 2>/dev/null | tee $OUT/ab_L.log
echo "U 65536"; tools/ab.sh --workload U 2>/dev/null | tee $OUT/ab_U.log
echo "config 4"; tools/ab.sh --config 4 --steps 5 2>/dev/null | tee $OUT/ab_c4.log
