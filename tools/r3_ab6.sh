#!/bin/bash
# one box: round-2 build / pinned row in LDS (v4pinrow) / masked gather (v5masked) / current
OUT=gpurun_out/r3x; mkdir -p $OUT
echo "T 65536"; tools/ab.sh 2>/dev/null | tee $OUT/ab_auto.log
echo "rule set 65536 x 1 KB"; tools/ab.sh --workload R --steps 8 2>/dev/null | tee $OUT/ab_R.log
echo "l7 65536 x 1 KB"; tools/ab.sh --workload L --steps 8 2>/dev/null | tee $OUT/ab_L.log
echo "T 131072"; tools/ab.sh --config 3 2>/dev/null | tee $OUT/ab_c3.log
echo "config 4"; tools/ab.sh --config 4 --steps 5 2>/dev/null | tee $OUT/ab_c4.log
