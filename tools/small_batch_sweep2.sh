#!/bin/bash
# As small_batch_sweep.sh, on the compiled rule set (R) and the l7 automaton (L).
TAG=${1:-sb}; OUT=gpurun_out/${TAG}_small2.log; : > $OUT
for W in R L; do
for N in 64 1024 4096; do
  for K in auto sym_reg; do
    L=$(timeout -k 10 120 python bench.py --workload $W --streams-per-gpu $N --kernel $K --steps 10 --warmup 2 --no-cpu-baseline --no-second-distribution 2>/dev/null | tail -1)
    echo "$W $N $K $(echo "$L" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms", d["roofline"]["kernel_ms_avg"], "Gbit/s", round(d["value"],1), d["config"].get("kernel"), d["config"].get("kernel_variant"))' 2>/dev/null)" >> $OUT
  done
done
done
cat $OUT
