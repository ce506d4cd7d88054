#!/bin/bash
# hand-off mix (bench.py handoff_mix_T): library builds side by side on one box
for lib in regex-fpga_amd/librxmatch_v6_6565ffd.so regex-fpga_amd/librxmatch.so; do
  RX_LIBRARY_PATH=$PWD/$lib timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); m=d.get('handoff_mix_T',{})
print('$(basename $lib)', 'T', d['roofline']['kernel_ms_avg'], 'handoff_mix', m.get('one_in_64_trapped',{}).get('kernel_ms_avg'), 'clean', m.get('clean',{}).get('kernel_ms_avg'), 'slowdown', m.get('slowdown'))"
done
