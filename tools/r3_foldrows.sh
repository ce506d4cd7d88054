#!/bin/bash
# FOLD builds: final rows cleared behind the first window, one granule per entry-less stream at the end: suite, then U / T / L
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r3_fr_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r3_fr_pytest.log
[ $rc = 0 ] || exit 1
for W in U T; do
  for r in 1 2; do
    for lib in regex-fpga_amd/librxmatch_base.so regex-fpga_amd/librxmatch_v6_6565ffd.so regex-fpga_amd/librxmatch.so; do
      RX_LIBRARY_PATH=$PWD/$lib timeout -k 10 200 python3 bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-second-distribution 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); c=d['config']
print('$W $r $(basename $lib)', c.get('kernel_variant'), 'kernel_ms', d['roofline']['kernel_ms_avg'])" || exit 1
    done
  done
done
