#!/bin/bash
# one box: round-2 build / v1 / current without pairs (v3nopair) / current, on the workloads with long lists too
TAG=${1:-n}; OUT=gpurun_out/r3u_$TAG; mkdir -p $OUT
echo "AUTO (S13) T 65536"; tools/ab.sh 2>/dev/null | tee $OUT/ab_auto.log
echo "config 4: T 131072 x 4 KB"; tools/ab.sh --config 4 --steps 5 2>/dev/null | tee $OUT/ab_c4.log
echo "rule set 65536 x 1 KB"; tools/ab.sh --workload R --steps 8 2>/dev/null | tee $OUT/ab_R.log
echo "l7 65536 x 1 KB"; tools/ab.sh --workload L --steps 8 2>/dev/null | tee $OUT/ab_L.log
echo "T 65536 S16"; tools/ab.sh --kernel sym_pack --group-lanes 16 2>/dev/null | tee $OUT/ab_s16.log
