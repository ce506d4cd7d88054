#!/bin/bash
# headline workload, pack kernel at several streams-per-wavefront: tools/s_sweep.sh [workload] [S...]
W=${1:-T}; shift || true
for gl in ${*:-8 12 16 20 24 32}; do
  python3 bench.py --workload $W --kernel sym_pack --group-lanes $gl --steps 10 --warmup 2 --no-cpu-baseline --no-second-distribution | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$W pack', $gl, d['ms_per_step'], d['value'])" || exit 1
done
