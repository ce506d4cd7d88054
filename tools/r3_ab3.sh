#!/bin/bash
# one box: round-2 build / v1 (trimmed sweep) / v2 (round-2 sweep + round-3 eviction and finals), then v2 with the
# de-synchronisation experiments (rx_opts.flags >> 16: stagger, static priorities)
TAG=${1:-l}; OUT=gpurun_out/r3t_$TAG; mkdir -p $OUT
rm -f regex-fpga_amd/librxmatch_v1.so.skip
echo "AUTO (S13)"; tools/ab.sh 2>/dev/null | tee $OUT/ab_auto.log
one() { python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-second-distribution "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['config']['streams_per_gpu'], d['config']['kernel_variant'], 'ms', d['roofline']['kernel_ms_avg'], 'Gbit/s', d['value'])"; }
for f in 0 $(( (1 + (2<<8)) << 16 )) $(( (1 + (4<<8)) << 16 )) $(( (1 + (8<<8)) << 16 )) $(( (1 + (16<<8)) << 16 )) $(( 2 << 16 )) $(( 4 << 16 )) $(( (3 + (4<<8)) << 16 )); do
  echo -n "flags=$f tweak=$((f >> 16)) : "; one --flags $f
done 2>&1 | tee $OUT/tweaks.log
