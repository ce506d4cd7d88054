#!/bin/bash
# l7-filter automaton on windows of its traces: AUTO vs explicit kernels
for k in "auto" "sym_wave" "sym_pack --group-lanes 4" "sym_pack --group-lanes 8" "sym_pack --group-lanes 16"; do
  python3 bench.py --workload L --kernel $k --steps 5 --warmup 1 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$k', d['config']['kernel'], d['ms_per_step'], d['value'])" || exit 1
done
