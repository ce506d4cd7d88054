#!/bin/bash
# last session of round 3: profile of the default bench line (traffic.json's source hash), the default line itself, hand-off mix
tools/profile.sh r03_final2 > gpurun_out/r3_f5_profile.log 2>&1; tail -2 gpurun_out/r3_f5_profile.log
timeout -k 10 400 python3 bench.py > gpurun_out/r3_f5_bench.log 2> gpurun_out/r3_f5_bench.err || { echo "bench failed"; tail -5 gpurun_out/r3_f5_bench.err; exit 3; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r3_f5_bench.log').read().strip().splitlines()[-1])
print('T', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], 'U', d['distribution_U']['gbit_s'], 'mix', d['handoff_mix_T']['one_in_64_trapped']['kernel_ms_avg'], d['handoff_mix_T']['slowdown'])
print('h2h', d['host_to_host_gbit_s'], d['host_to_host_compact_final_sets_gbit_s'], d['host_to_host_no_final_sets_gbit_s'], 'single', d['single_stream_config1']['kernel_ms'])"
for c in 3 4; do timeout -k 10 400 python3 bench.py --config $c --no-cpu-baseline --no-second-distribution --steps 6 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('config $c', d['config'].get('streams_per_gpu'), d['config'].get('stream_len'), d['value'], d['roofline']['kernel_ms_avg'], d['config'].get('kernel_variant'))"; done
for W in U R L; do timeout -k 10 400 python3 bench.py --workload $W --no-cpu-baseline --no-second-distribution --steps 10 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('$W', d['value'], d['roofline']['kernel_ms_avg'], d['config'].get('kernel_variant'))"; done
