#!/bin/bash
# instruction mix of the rule-set stand-in's kernel (pack kernel S=8, look-ahead pruning): round-2 build against the current one
export TMPDIR=/tmp
ARGS="--workload R --steps 3 --warmup 1 --no-cpu-baseline --no-second-distribution"
for lib in librxmatch_base.so librxmatch.so; do
  OUT=gpurun_out/pmcR_${lib%.so}; mkdir -p $OUT
  export RX_LIBRARY_PATH=$PWD/regex-fpga_amd/$lib
  timeout -k 5 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -- python3 bench.py $ARGS > $OUT/a.log 2>&1
  timeout -k 5 240 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -- python3 bench.py $ARGS > $OUT/b.log 2>&1
  timeout -k 5 240 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/c -- python3 bench.py $ARGS > $OUT/c.log 2>&1
  echo "== $lib"
  python3 - <<PY
import csv,glob,collections
for d in ['a','b','c']:
    for f in glob.glob('$OUT/'+d+'/*/*_counter_collection.csv'):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'rx_sym_pack_kernel<8, false, false, true' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in sorted(agg.items()): print(k,'%.5g'%(sum(v)/len(v)), len(v))
PY
done
