#!/bin/bash
# same-box bisect of the round's kernel commits on one workload: tools/r3_bisect.sh <workload> [bench args]
# (libraries: regex-fpga_amd/librxmatch_base.so = round 2, librxmatch_v<n>_<commit>.so, librxmatch.so = HEAD)
W=${1:-R}; shift
OUT=gpurun_out/r3_bisect_$W.log; : > $OUT
ARGS="--workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-second-distribution $*"
for r in 1 2 3; do
  for lib in regex-fpga_amd/librxmatch_base.so regex-fpga_amd/librxmatch_v*.so regex-fpga_amd/librxmatch.so; do
    [ -f $lib ] || continue
    RX_LIBRARY_PATH=$PWD/$lib timeout -k 10 200 python3 bench.py $ARGS 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); c=d['config']
print('$r $(basename $lib)', c.get('kernel'), c.get('kernel_variant'), 'kernel_ms', d['roofline']['kernel_ms_avg'], 'step_ms', d['ms_per_step'])" | tee -a $OUT || exit 1
  done
done
