#!/bin/bash
for c in 3 4; do timeout -k 10 400 python3 bench.py --config $c --no-cpu-baseline --no-second-distribution --steps 6 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('config $c', d['config'].get('streams_per_gpu'), d['config'].get('stream_len'), d['value'], d['roofline']['kernel_ms_avg'], d['config'].get('kernel_variant'))"; done
for W in U R L; do timeout -k 10 400 python3 bench.py --workload $W --no-cpu-baseline --no-second-distribution --steps 10 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('$W', d['value'], d['roofline']['kernel_ms_avg'], d['config'].get('kernel_variant'))"; done
timeout -k 10 300 python3 bench.py --config 3 --workload U --no-cpu-baseline --no-second-distribution --steps 6 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('config 3 U', d['value'], d['roofline']['kernel_ms_avg'], d['config'].get('kernel_variant'))"
