#!/bin/bash
# resident-entry kernel: streams handed off and time per streams-per-wavefront (verbose flag 4)
for S in 8 16 24 32 48; do
  timeout -k 10 120 python3 bench.py --kernel sym_res --group-lanes $S --steps 5 --warmup 1 --flags 4 --no-cpu-baseline --no-second-distribution 2> gpurun_out/r3_res2_$S.err | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); c=d['config']
print('S=$S', c.get('kernel'), c.get('kernel_variant'), 'kernel_ms', d['roofline']['kernel_ms_avg'], 'Gbit/s', d['value'])" || exit 1
  grep -m2 "handed\|launch geometry\|grid" gpurun_out/r3_res2_$S.err
done
