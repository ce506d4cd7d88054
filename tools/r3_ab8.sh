#!/bin/bash
# the per-stream hand-off code outside the loop of passes: suite + fuzz, then T / U / R / L against round 2, v6 (hand-off inside
# the pass) and the current build
for W in R T; do
  for r in 1 2; do
    for lib in regex-fpga_amd/librxmatch_base.so regex-fpga_amd/librxmatch_v6_6565ffd.so regex-fpga_amd/librxmatch_v9*.so regex-fpga_amd/librxmatch.so; do
      RX_LIBRARY_PATH=$PWD/$lib timeout -k 10 200 python3 bench.py --workload $W --steps 12 --warmup 3 --no-cpu-baseline --no-second-distribution 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); c=d['config']
print('$W $r $(basename $lib)', c.get('kernel_variant'), 'kernel_ms', d['roofline']['kernel_ms_avg'])" || exit 1
    done
  done
done
