#!/bin/bash
# configs[4] stand-in (synthetic ruleset): pack kernel at several streams-per-wavefront vs the wave kernel
mkdir -p gpurun_out
for gl in ${*:-2 4 8}; do
  python3 bench.py --workload R --kernel sym_pack --group-lanes $gl --steps 5 --warmup 1 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('pack', $gl, d['ms_per_step'], d['value'], d.get('handoffs'))" || exit 1
done
python3 bench.py --workload R --kernel sym_wave --steps 5 --warmup 1 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('wave', d['ms_per_step'], d['value'])"
python3 bench.py --workload R --kernel auto --steps 5 --warmup 1 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('auto', d['config']['kernel'], d['ms_per_step'], d['value'])"
