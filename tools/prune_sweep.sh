#!/bin/bash
# look-ahead pruning on/off over streams-per-wavefront: tools/prune_sweep.sh <workload> [S...]
W=${1:-R}; shift || true
for gl in ${*:-4 8 12 16}; do
  for np in 0 1; do
    python3 bench.py --workload $W --kernel sym_pack --group-lanes $gl --flags $np --steps 5 --warmup 1 --no-cpu-baseline --no-second-distribution | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$W S=$gl noprune=$np', d['ms_per_step'], d['value'])" || exit 1
  done
done
python3 bench.py --workload $W --kernel auto --steps 5 --warmup 1 --no-cpu-baseline --no-second-distribution | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$W auto', d['config']['kernel'], d['ms_per_step'], d['value'])"
