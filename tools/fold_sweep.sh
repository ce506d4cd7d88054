#!/bin/bash
# FOLD builds of the pack kernel over streams-per-wavefront: tools/fold_sweep.sh <workload> <streams> <len> [S...]
W=${1:-T}; NS=${2:-65536}; SL=${3:-1024}; shift 3 || true
one() { python3 bench.py --workload $W --streams-per-gpu $NS --stream-len $SL --steps 8 --warmup 2 --no-cpu-baseline --no-second-distribution "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$W ${NS}x$SL', d['config']['kernel'], d['config']['kernel_variant'], 'ms', d['roofline']['kernel_ms_avg'], 'Gbit/s', d['value'])"; }
for S in ${*:-8 13 16 24 32 48 64}; do one --kernel sym_pack --group-lanes $S --flags 32 || exit 1; done
one --kernel auto
one --kernel auto --flags 16
