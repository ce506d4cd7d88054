#!/bin/bash
# Round 3, final measurement session: rocprofv3 summary of the default bench (kernel trace + PMC passes), the full bench line,
# the other BASELINE shapes.  tools/r3_final.sh [tag]
TAG=${1:-z}; OUT=gpurun_out/r3z_$TAG; mkdir -p $OUT
tools/profile.sh r03_$TAG > $OUT/profile.log 2>&1; tail -3 $OUT/profile.log
timeout -k 10 600 python3 bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err || tail -5 $OUT/bench_default.err
tail -c 600 $OUT/bench_default.log; echo
one() { python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-second-distribution "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['config']['baseline_config_index'], d['config']['streams_per_gpu'], 'x', d['config']['stream_len'], d['config']['kernel'], d['config']['kernel_variant'], 'ms', d['roofline']['kernel_ms_avg'], 'Gbit/s', d['value'])"; }
{ echo "config 3 T"; one --config 3; echo "config 3 U"; one --config 3 --workload U; echo "262144 T"; one --streams-per-gpu 262144; echo "262144 U"; one --streams-per-gpu 262144 --workload U;
  echo "config 4 T (4 KB windows)"; one --config 4; echo "config 4 R (rule set)"; one --config 4 --workload R; echo "65536 R 1KB"; one --workload R;
  echo "l7 65536"; one --workload L; echo "16384 T"; one --streams-per-gpu 16384; } 2>&1 | tee $OUT/shapes.log
