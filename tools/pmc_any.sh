#!/bin/bash
# SQ instruction mix of every match kernel in a bench run: tools/pmc_any.sh <tag> [bench args]
TAG=${1:-q}; shift || true
OUT=gpurun_out/pa_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-second-distribution $*"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -- python3 bench.py $ARGS > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -- python3 bench.py $ARGS > $OUT/b.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ['a','b']:
    for f in glob.glob('$OUT/'+d+'/*/*_counter_collection.csv'):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if 'false>' in r['Kernel_Name']:
                agg[r['Kernel_Name'][28:60]][r['Counter_Name']].append(float(r['Counter_Value']))
        for kn,c in agg.items():
            for k,v in sorted(c.items()): print(kn,k,'%.4g'%(sum(v)/len(v)),'n=%d'%len(v))
PY
