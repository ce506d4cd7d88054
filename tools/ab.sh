#!/bin/bash
# A/B two builds of librxmatch on ONE box: tools/ab.sh [bench args]   (base = regex-fpga_amd/librxmatch_base.so)
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-second-distribution $*"
mkdir -p gpurun_out
for r in 1 2; do
  RX_LIBRARY_PATH=$PWD/regex-fpga_amd/librxmatch_base.so python3 bench.py $ARGS | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('base', d['ms_per_step'], d['value'])" || exit 1
  python3 bench.py $ARGS | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('new ', d['ms_per_step'], d['value'])" || exit 1
done
