#!/bin/bash
# PRUNE builds (look-ahead pruning incl. inline targets on narrow automata) vs plain, over streams per wavefront
W=${1:-T}; NS=${2:-65536}; SL=${3:-1024}
one() { python3 bench.py --workload $W --streams-per-gpu $NS --stream-len $SL --steps 8 --warmup 2 --no-cpu-baseline --no-second-distribution "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$W ${NS}x$SL', d['config']['kernel'], d['config']['kernel_variant'], 'ms', d['roofline']['kernel_ms_avg'], 'Gbit/s', d['value'])"; }
for S in 13 16 20 22 24; do one --kernel sym_pack --group-lanes $S --flags 2 || exit 1; done
one --kernel sym_pack --group-lanes 13 --flags 1
one --kernel auto
one --kernel auto --flags 4 2>&1 | tail -1
