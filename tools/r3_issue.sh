#!/bin/bash
# Round 3, "which unit is behind the 0.25 instructions/cycle/SIMD of the pack kernel": tools/r3_issue.sh [tag]
# (1) attainable issue rates per mix (tools/issue_bench), (2) shader clock under load (stamped build), (3) time per pass
# against waves per SIMD, (4) three more --pmc groups on the shipped kernel (branches, scalar unit, instruction fetch, clock).
set -u
TAG=${1:-a}
OUT=gpurun_out/r3i_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
tools/issue_bench 20000 > $OUT/issue_bench.log 2>&1 || echo "issue_bench failed"
B="--steps 10 --warmup 2 --no-cpu-baseline --no-second-distribution"
python3 bench.py $B --kernel sym_pack --group-lanes 16 --flags 12 > $OUT/prof16.log 2>&1 || echo "prof failed"
for w in 1 2 3 4 5 6 8; do
  python3 bench.py $B --kernel sym_pack --group-lanes 13 --streams-per-gpu $((13 * 1024 * w)) > $OUT/w$w.log 2>&1 || echo "w$w failed"
done
P="--steps 3 --warmup 1 --no-cpu-baseline --no-second-distribution"
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/g1 -- python3 bench.py $P > $OUT/g1.log 2>&1 || echo "g1 failed"
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/g2 -- python3 bench.py $P > $OUT/g2.log 2>&1 || echo "g2 failed"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/g3 -- python3 bench.py $P > $OUT/g3.log 2>&1 || echo "g3 failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py $P > $OUT/kt.log 2>&1 || echo "kt failed"
python3 - <<PY
import csv, glob, collections
for d in ("g1", "g2", "g3"):
    for f in glob.glob("$OUT/" + d + "/*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "rx_sym_pack_kernel" in r["Kernel_Name"] and ", true," not in r["Kernel_Name"][:70]:
                agg[r["Kernel_Name"][28:72]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for kn, c in agg.items():
            for k, v in sorted(c.items()):
                print(d, kn, k, "%.5g" % (sum(v) / len(v)), "n=%d" % len(v))
PY
grep -h "MHz\|phase" $OUT/prof16.log | head -12
for w in 1 2 3 4 5 6 8; do python3 -c "
import json,sys
for l in open('$OUT/w$w.log'):
    if l.startswith('{'):
        d=json.loads(l); print('w=$w', d['config']['streams_per_gpu'], d['roofline']['kernel_ms_avg'], 'ms', d['value'], 'Gbit/s', d['config']['kernel_variant'])
"; done
tail -5 $OUT/issue_bench.log
