#!/bin/bash
# per-kernel times of the hand-off mix (tools/r3_mix.py) -> gpurun_out/r3_mixprof/ (summary kept in profiles/r03_handoff_mix/)
export TMPDIR=/tmp
rm -rf gpurun_out/r3_mixprof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_mixprof -- python3 tools/r3_mix.py 0 > gpurun_out/r3_mixprof.log 2>&1
grep "mix:" gpurun_out/r3_mixprof.log | tail -1
f=$(find gpurun_out/r3_mixprof -name "*kernel_stats.csv" | head -1); head -6 "$f" | cut -c1-220
