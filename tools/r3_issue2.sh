#!/bin/bash
# Round 3, second session on the issue ceiling: more microbenchmark mixes (VALU->SGPR hops, VOP3, waits, LDS atomics,
# scattered gathers) and the vector-memory path's counters (TA / TCP / TD) of the shipped pack kernel.  tools/r3_issue2.sh [tag]
set -u
TAG=${1:-b}
OUT=gpurun_out/r3i_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
tools/issue_bench 20000 quick > $OUT/issue_bench2.log 2>&1 || echo "issue_bench failed"
P="--steps 3 --warmup 1 --no-cpu-baseline --no-second-distribution"
i=0
for grp in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_BUSY_avr" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_LFIFO_STALL_CYCLES_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_SPI_STALL_sum" \
           "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_LDS_ATOMIC"; do
  i=$((i + 1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/t$i -- python3 bench.py $P > $OUT/t$i.log 2>&1 || echo "t$i failed: $grp"
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/t*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "rx_sym_pack_kernel<13" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][28:72]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, c in agg.items():
        for k, v in sorted(c.items()):
            print(kn, k, "%.5g" % (sum(v) / len(v)), "n=%d" % len(v))
PY
cat $OUT/issue_bench2.log
