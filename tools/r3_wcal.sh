#!/bin/bash
OUT=gpurun_out/r3wc; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 5 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- tools/write_calib > $OUT/w.log 2>&1 || echo "failed"
timeout -k 5 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- tools/write_calib > $OUT/f.log 2>&1 || echo "failed"
cat $OUT/w.log | tail -2
python3 - <<PY
import csv, glob, collections
for d, name in (("w", "WRITE_SIZE"), ("f", "FETCH_SIZE")):
  for f in glob.glob("$OUT/" + d + "/*/*_counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print(k, "launches", len(v), name, "KB total", round(sum(v), 1), "per launch", round(sum(v)/len(v), 1))
PY
