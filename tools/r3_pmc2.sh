#!/bin/bash
# Round 3: vector-memory path / LDS occupancy counters of the shipped pack kernel, one guarded rocprofv3 pass per group
# (a TA counter group once aborted the profiled process and hung the profiler).  tools/r3_pmc2.sh [tag]
TAG=${1:-g}; OUT=gpurun_out/r3m_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
P="--steps 3 --warmup 1 --no-cpu-baseline --no-second-distribution"
i=0
for grp in "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i + 1))
  echo "pass $i: $grp"
  timeout -k 5 240 rocprofv3 --pmc $grp --output-format csv -d $OUT/t$i -- python3 bench.py $P > $OUT/t$i.log 2>&1 || echo "t$i failed/killed: $grp"
  python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/t$i/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "rx_sym_pack_kernel<13" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print("  ", k, "%.6g" % (sum(v) / len(v)), "n=%d" % len(v))
PY
done
