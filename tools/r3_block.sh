#!/bin/bash
# workgroup-per-stream resume mode: hand-off tests, full suite, the mix, the wave kernel as a whole-batch kernel
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r3_blk_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r3_blk_pytest.log
[ $rc = 0 ] || exit 1
tools/r3_handoff3.sh
timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-second-distribution --kernel sym_wave 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('sym_wave whole batch', d['roofline']['kernel_ms_avg'])"
timeout -k 10 200 python3 tools/fuzz_gpu.py --seconds 150 --seed 999 > gpurun_out/r3_blk_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/r3_blk_fuzz.log
