#!/bin/bash
# closing session: whole GPU suite, then profile of the default bench line (traffic.json's source hash) and the line itself
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r3_f7_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r3_f7_pytest.log
[ $rc = 0 ] || exit 1
tools/profile.sh r03_final4 > gpurun_out/r3_f7_profile.log 2>&1; tail -1 gpurun_out/r3_f7_profile.log
timeout -k 10 400 python3 bench.py > gpurun_out/r3_f7_bench.log 2> gpurun_out/r3_f7_bench.err || { echo "bench failed"; tail -5 gpurun_out/r3_f7_bench.err; exit 3; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r3_f7_bench.log').read().strip().splitlines()[-1])
print('T', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], 'U', d['distribution_U']['gbit_s'], 'mix', d['handoff_mix_T']['one_in_64_trapped']['kernel_ms_avg'], d['handoff_mix_T']['slowdown'])
print('traffic', d['roofline']['traffic'], d['roofline']['traffic_note'][:80])"
