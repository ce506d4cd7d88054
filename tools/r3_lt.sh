#!/bin/bash
# Round 3: the LDS-table build of the pack kernel: parity (fast subset + fuzz), then timings against the plain build.
TAG=${1:-i}; OUT=gpurun_out/r3l_$TAG; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kat_ab or synthetic_batches or handoff or random_automata or convention or larger_than_list or shipped_traces" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout -k 10 200 python3 tools/fuzz_gpu.py --seconds 120 --seed 31 > $OUT/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $OUT/fuzz.log
one() { python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-second-distribution "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['config']['streams_per_gpu'], d['config']['kernel_variant'], 'ms', d['roofline']['kernel_ms_avg'], 'Gbit/s', d['value'])"; }
for NS in 65536 131072; do
  one --kernel sym_pack --group-lanes 13 --streams-per-gpu $NS
  for S in 8 13 16 24; do echo -n "LT "; one --kernel sym_pack --group-lanes $S --flags 516 --streams-per-gpu $NS 2>&1; done
done 2>&1 | tee $OUT/lt_sweep.log
for w in 1 2 3 4; do echo -n "LT16 w=$w "; one --kernel sym_pack --group-lanes 16 --flags 512 --streams-per-gpu $((16 * 1024 * w)); done 2>&1 | tee $OUT/lt_w.log
