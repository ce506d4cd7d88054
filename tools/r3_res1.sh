#!/bin/bash
# first session of the resident-entry kernel: parity subset, then timing per streams-per-wavefront
OUT=gpurun_out/r3_res1; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "kat or shipped or synthetic or ragged or idle or unaligned or chunked or larger_than or handoff or random_automata or reference_convention or compact or pipelined" > $OUT/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -15 $OUT/pytest.log
[ $rc = 0 ] || exit 1
for S in 8 16 24 32 48; do
  timeout -k 10 120 python3 bench.py --kernel sym_res --group-lanes $S --steps 10 --warmup 2 --no-cpu-baseline --no-second-distribution 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); c=d['config']
print('S=$S', c.get('kernel'), c.get('kernel_variant'), 'kernel_ms', d['roofline']['kernel_ms_avg'], 'Gbit/s', d['value'])" || exit 1
done
timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-second-distribution 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); c=d['config']
print('AUTO', c.get('kernel'), c.get('kernel_variant'), 'kernel_ms', d['roofline']['kernel_ms_avg'], 'Gbit/s', d['value'])"
