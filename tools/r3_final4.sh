#!/bin/bash
# longer lists for the launch that finishes hand-offs: whole GPU suite, hand-off mix before / after, the wave kernels as
# whole-batch kernels, then the round's profile of the default bench line (refreshes profiles/traffic.json's source hash)
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r3_f4_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r3_f4_pytest.log
[ $rc = 0 ] || exit 1
tools/r3_handoff3.sh
timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-second-distribution --kernel sym_wave 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('sym_wave whole batch', d['roofline']['kernel_ms_avg'])"
timeout -k 10 200 python3 tools/fuzz_gpu.py --seconds 150 --seed 777 > gpurun_out/r3_f4_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/r3_f4_fuzz.log
tools/profile.sh r03_final > gpurun_out/r3_f4_profile.log 2>&1; tail -3 gpurun_out/r3_f4_profile.log
