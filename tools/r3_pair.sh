#!/bin/bash
# Round 3: interleaved double sweeps — parity subset + fuzz, then streams-per-wavefront sweeps (plain and FOLD builds).
TAG=${1:-j}; OUT=gpurun_out/r3q_$TAG; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_plan_run.py -m gpu -x -q -k "kat_ab or synthetic_batches or handoff or random_automata or convention or larger_than_list or evictions" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout -k 10 150 python3 tools/fuzz_gpu.py --seconds 90 --seed 47 > $OUT/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $OUT/fuzz.log
one() { python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-second-distribution "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['config']['streams_per_gpu'], d['config']['kernel_variant'], 'ms', d['roofline']['kernel_ms_avg'], 'Gbit/s', d['value'])"; }
for NS in 65536 131072 262144; do
  for S in 13 16 20 24 32; do one --kernel sym_pack --group-lanes $S --streams-per-gpu $NS; done
  for S in 24 32 48 64; do one --kernel sym_pack --group-lanes $S --flags 32 --streams-per-gpu $NS; done
done 2>&1 | tee $OUT/s_sweep.log
for w in 1 2 3; do echo -n "S32 w=$w "; one --kernel sym_pack --group-lanes 32 --streams-per-gpu $((32 * 1024 * w)); done 2>&1 | tee $OUT/w32.log
one --streams-per-gpu 65536 | tee $OUT/auto.log
