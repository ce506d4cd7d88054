#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r3_mixprof -o mix -- python3 tools/r3_mix.py 4 > gpurun_out/r3_mixprof.log 2>&1
grep "mix:\|handed" gpurun_out/r3_mixprof.log | tail -3
f=$(find gpurun_out/r3_mixprof -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-200
