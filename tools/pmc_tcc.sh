#!/bin/bash
# L2 (TCC) request / hit / miss counts per kernel: how many of the slice gathers leave the CU's L1
TAG=${1:-q}; shift || true
OUT=gpurun_out/tcc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-second-distribution $*"
timeout -k 10 150 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum --output-format csv -d $OUT/a -- python3 bench.py $ARGS > $OUT/a.log 2>&1 || { echo "rocprofv3 failed"; tail -5 $OUT/a.log; exit 1; }
python3 - <<PY
import csv,glob,collections
for f in glob.glob('$OUT/a/*/*_counter_collection.csv'):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'false' in r['Kernel_Name'] and 'pack' in r['Kernel_Name']:
            agg[r['Kernel_Name'][28:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    for kn,c in agg.items():
        for k,v in sorted(c.items()): print(kn,k,'%.4g'%(sum(v)/len(v)),'n=%d'%len(v))
PY
