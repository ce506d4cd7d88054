#!/bin/bash
# Round 3 perf session: round-2 build against the current one on ONE box, streams-per-wavefront sweep, waves-per-SIMD sweep.
# tools/r3_perf1.sh [tag]
TAG=${1:-f}; OUT=gpurun_out/r3p_$TAG; mkdir -p $OUT
tools/ab.sh > $OUT/ab.log 2>&1; cat $OUT/ab.log
one() { python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-second-distribution "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['config']['streams_per_gpu'], d['config']['kernel_variant'], 'ms', d['roofline']['kernel_ms_avg'], 'Gbit/s', d['value'])"; }
for NS in 65536 131072; do for S in 8 11 13 16 20 22 24 32; do one --kernel sym_pack --group-lanes $S --streams-per-gpu $NS; done; done 2>&1 | tee $OUT/s_sweep.log
for w in 1 2 3 4 5; do one --kernel sym_pack --group-lanes 13 --streams-per-gpu $((13 * 1024 * w)); done 2>&1 | tee $OUT/w_sweep.log
timeout -k 10 500 python3 bench.py > $OUT/bench.log 2> $OUT/bench.err || tail -5 $OUT/bench.err
tail -c 2500 $OUT/bench.log
