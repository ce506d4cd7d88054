#!/bin/bash
# parity subset + fuzz of the current build, then round-2 / round-3-v1 / current on one box (tools/ab.sh)
TAG=${1:-k}; OUT=gpurun_out/r3r_$TAG; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_plan_run.py -m gpu -x -q -k "kat_ab or synthetic_batches or handoff or random_automata or convention or larger_than_list or evictions" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout -k 10 200 python3 tools/fuzz_gpu.py --seconds 150 --seed 47 > $OUT/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $OUT/fuzz.log
echo "AUTO (S13)"; tools/ab.sh 2>/dev/null | tee $OUT/ab_auto.log
echo "S16"; tools/ab.sh --kernel sym_pack --group-lanes 16 2>/dev/null | tee $OUT/ab_s16.log
echo "131072 S13"; tools/ab.sh --kernel sym_pack --group-lanes 13 --streams-per-gpu 131072 2>/dev/null | tee $OUT/ab_131k.log
