#!/bin/bash
# many-streams register kernel over streams per wavefront: tools/rpack_sweep.sh <workload> <streams> <len>
W=${1:-T}; NS=${2:-65536}; SL=${3:-1024}
one() { python3 bench.py --workload $W --streams-per-gpu $NS --stream-len $SL --steps 8 --warmup 2 --no-cpu-baseline --no-second-distribution "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$W ${NS}x$SL', d['config']['kernel'], d['config']['kernel_variant'], 'ms', d['roofline']['kernel_ms_avg'], 'Gbit/s', d['value'])"; }
for S in 8 16 24 32; do one --kernel sym_rpack --group-lanes $S || exit 1; done
one --kernel auto
