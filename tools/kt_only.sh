#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py for one workload: tools/kt_only.sh <tag> [bench args] -> gpurun_out/kt_<tag>/
TAG=${1:-run}; shift || true
OUT=gpurun_out/kt_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-second-distribution "$@" > $OUT/bench.log 2>&1 || echo "rocprofv3 failed"
tail -1 $OUT/bench.log | cut -c1-160
