#!/bin/bash
# A/B builds of librxmatch on ONE box (box-to-box spread is ~3%): tools/ab.sh [bench args]
# variants: regex-fpga_amd/librxmatch_base.so, any regex-fpga_amd/librxmatch_v*.so, and the current build
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-second-distribution $*"
mkdir -p gpurun_out
for r in 1 2; do
  for lib in regex-fpga_amd/librxmatch_base.so regex-fpga_amd/librxmatch_v*.so regex-fpga_amd/librxmatch.so; do
    [ -f $lib ] || continue
    RX_LIBRARY_PATH=$PWD/$lib python3 bench.py $ARGS | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$(basename $lib)', d['ms_per_step'], d['value'])" || exit 1
  done
done
