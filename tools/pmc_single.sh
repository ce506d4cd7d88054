#!/bin/bash
# Counters of the register kernel on one shipped trace as one stream: tools/pmc_single.sh <tag> hi|lo
TAG=${1:-s}; W=${2:-hi}
OUT=gpurun_out/ps_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -- python3 tools/single_one.py $W 3 > $OUT/a.log 2>&1 || echo "a failed"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH --output-format csv -d $OUT/b -- python3 tools/single_one.py $W 3 > $OUT/b.log 2>&1 || echo "b failed"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/c -- python3 tools/single_one.py $W 3 > $OUT/c.log 2>&1 || echo "c failed"
python3 - <<PY
import csv,glob,collections
for d in ['a','b','c']:
    for f in glob.glob('$OUT/'+d+'/*/*_counter_collection.csv'):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'rx_sym_reg' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in sorted(agg.items()): print('$W',k,'%.5g'%(sum(v)/len(v)),'per pass %.2f'%(sum(v)/len(v)/199999),'n=%d'%len(v))
PY
