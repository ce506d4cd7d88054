#!/bin/bash
# quick SQ instruction-mix profile: tools/pmc_quick.sh <tag> [bench args]
TAG=${1:-q}; shift || true
OUT=gpurun_out/pq_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-second-distribution $*"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -- python3 bench.py $ARGS > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -- python3 bench.py $ARGS > $OUT/b.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ['a','b']:
    for f in glob.glob('$OUT/'+d+'/*/*_counter_collection.csv'):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'false>' in r['Kernel_Name'] and 'group' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in sorted(agg.items()): print(k,'%.4g'%(sum(v)/len(v)))
PY
