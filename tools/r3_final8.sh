#!/bin/bash
# closing session after the staged downloads of rx_plan_run: suite, host-to-host figures, the default bench line
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r3_f8_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r3_f8_pytest.log
[ $rc = 0 ] || exit 1
timeout -k 10 200 python3 tools/h2h_compact.py 2>&1 | grep -v amdgpu | tail -3
timeout -k 10 400 python3 bench.py > gpurun_out/r3_f8_bench.log 2> gpurun_out/r3_f8_bench.err || { echo "bench failed"; tail -5 gpurun_out/r3_f8_bench.err; exit 3; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r3_f8_bench.log').read().strip().splitlines()[-1])
print('T', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], 'traffic', d['roofline']['traffic'])
print('h2h rows', d['host_to_host_gbit_s'], 'compact', d['host_to_host_compact_final_sets_gbit_s'], 'none', d['host_to_host_no_final_sets_gbit_s'], 'floor', d['host_to_host_link_floor_gbit_s'])"
