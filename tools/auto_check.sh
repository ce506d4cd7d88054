#!/bin/bash
# AUTO's choice against fixed streams-per-wavefront at several batch sizes: tools/auto_check.sh <workload> <len> <streams...>
W=${1:-T}; SL=${2:-1024}; shift 2 || true
one() { python3 bench.py --workload $W --streams-per-gpu $NS --stream-len $SL --steps 6 --warmup 2 --no-cpu-baseline --no-second-distribution "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$W ${NS}x$SL', '$*', '->', d['config']['kernel'], d['config']['kernel_variant'], 'ms', d['roofline']['kernel_ms_avg'], 'Gbit/s', d['value'])"; }
for NS in ${*:-16384 65536 131072 262144}; do
  one --kernel auto
  for S in 4 8 11 13 16 22 24 32; do one --kernel sym_pack --group-lanes $S --flags 16; done
  for S in 16 32 64; do one --kernel sym_pack --group-lanes $S --flags 32; done
done
