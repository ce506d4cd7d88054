#!/bin/bash
# LDS A/B of the pack kernel (shipped build vs RX_AB_PREDICATE_IDLE build): kernel time and LDS counters of both, on the
# default bench command.  -> gpurun_out/lds_ab/{base,ab}_{kt,pmc}   usage: tools/pmc_ab_lds.sh
OUT=gpurun_out/lds_ab; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-second-distribution"
for v in base ab; do
  if [ $v = ab ]; then export RX_LIBRARY_PATH=$PWD/regex-fpga_amd/librxmatch_ab.so; else unset RX_LIBRARY_PATH; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${v}_kt -- python3 bench.py $ARGS > $OUT/${v}_kt.log 2>&1 || echo "kt failed"
  rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/${v}_pmc -- python3 bench.py $ARGS > $OUT/${v}_pmc.log 2>&1 || echo "pmc failed"
  # un-profiled wall time, three runs
  for i in 1 2 3; do python3 bench.py $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', d['roofline']['kernel_ms_avg'], d['value'])"; done
done
unset RX_LIBRARY_PATH
python3 - <<PY
import csv,glob,collections
for v in ('base','ab'):
    for f in glob.glob('$OUT/'+v+'_pmc/*/*_counter_collection.csv'):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'rx_sym_pack_kernel' in r['Kernel_Name'] and 'false, false, false, false' in r['Kernel_Name'].replace('true','x') :
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        print(v, {k: round(sum(x)/len(x)) for k,x in sorted(agg.items())})
    for f in glob.glob('$OUT/'+v+'_kt/*/*_kernel_stats.csv'):
        for r in csv.DictReader(open(f)):
            if 'rx_sym_pack' in r['Name']: print(v, r['Name'][:70], r['Calls'], r['AverageNs'])
PY
