#!/bin/bash
# One GPU-box session: the gpu tests, then (only if pytest itself ended normally) the default bench line.
# usage: tools/gpu_session.sh <tag> [pytest args...]
TAG=${1:-s}; shift || true
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q "$@" > gpurun_out/${TAG}_gputest.log 2>&1
rc=$?
echo "pytest exit=$rc" >> gpurun_out/${TAG}_gputest.log
tail -5 gpurun_out/${TAG}_gputest.log
if [ $rc -gt 1 ]; then echo "pytest was killed or interrupted: no further GPU step"; exit $rc; fi
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_bench.log 2> gpurun_out/${TAG}_bench.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}_bench.err; exit 3; }
tail -c 3000 gpurun_out/${TAG}_bench.log
