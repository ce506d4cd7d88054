#!/bin/bash
# what AUTO decides (RX_OPT_VERBOSE) per library build on the rule-set and l7 workloads
OUT=gpurun_out/r3v; mkdir -p $OUT
for lib in base v1 v3nopair ""; do
  f=regex-fpga_amd/librxmatch${lib:+_$lib}.so
  for wl in R L; do
    echo "== $f $wl"
    RX_LIBRARY_PATH=$PWD/$f python3 bench.py --workload $wl --flags 4 --steps 3 --warmup 1 --no-cpu-baseline --no-second-distribution 2>&1 | grep -E "rxmatch\] (probe|AUTO)|kernel_variant" | sed -e 's/.*"kernel_variant": "\([^"]*\)".*"kernel_ms_avg": \([0-9.]*\).*/variant \1 ms \2/' | head -12
  done
done
