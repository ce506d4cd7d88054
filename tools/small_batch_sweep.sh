#!/bin/bash
# Small batches: the pack kernel (AUTO) against one wavefront per stream (register kernel), T and U, 1 KB streams.
# usage: tools/small_batch_sweep.sh <tag>
TAG=${1:-sb}; OUT=gpurun_out/${TAG}_small.log; : > $OUT
for W in T U; do
for N in 16 64 256 1024 2048 4096 8192 16384; do
  for K in auto sym_reg; do
    L=$(timeout -k 10 120 python bench.py --workload $W --streams-per-gpu $N --kernel $K --steps 10 --warmup 2 --no-cpu-baseline --no-second-distribution 2>/dev/null | tail -1)
    echo "$W $N $K $(echo "$L" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms", d["roofline"]["kernel_ms_avg"], "Gbit/s", round(d["value"],1), d["config"].get("kernel"), d["config"].get("kernel_variant"))' 2>/dev/null)" >> $OUT
  done
done
done
cat $OUT
