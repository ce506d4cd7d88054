#!/bin/bash
# average VMEM / LDS instruction latency of the match kernel: tools/pmc_lat.sh <tag> [bench args]
TAG=${1:-q}; shift || true
OUT=gpurun_out/pl_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-second-distribution $*"
rocprofv3 -L > $OUT/counters.txt 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/a -- python3 bench.py $ARGS > $OUT/a.log 2>&1
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $OUT/b -- python3 bench.py $ARGS > $OUT/b.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/c -- python3 bench.py $ARGS > $OUT/c.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ['a','b','c']:
    for f in glob.glob('$OUT/'+d+'/*/*_counter_collection.csv'):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'false>' in r['Kernel_Name'] and ('pack' in r['Kernel_Name'] or 'group' in r['Kernel_Name']):
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in sorted(agg.items()): print(k,'%.5g'%(sum(v)/len(v)))
PY
tail -3 $OUT/b.log | cut -c1-300; tail -3 $OUT/c.log | cut -c1-300
