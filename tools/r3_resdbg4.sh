#!/bin/bash
# the fuzz case on the debug build (kernel printf of every eviction) next to the CPU model's
RX_LIBRARY_PATH=$PWD/regex-fpga_amd/librxmatch_dbg.so timeout -k 10 200 python3 - <<'PY' > gpurun_out/r3_resdbg4.log 2>&1
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools"); sys.path.insert(0, "tests")
import fuzz_gpu as f
rx = f.rx
seed, trial = 515, 5195
rng = np.random.default_rng([seed, trial])
kind = int(rng.integers(6))
while True:
    try:
        nfa = rx.Nfa.compile(f.rand_regexes(rng, int(rng.integers(1, 12))), icase=bool(rng.integers(2)))
        break
    except rx.RxError:
        continue
ns = int(rng.integers(1, 130)); sl = int(rng.choice([0, 1, 3, 15, 16, 17, 31, 33, 64, 100, 255, 256, 257, 400, 1000]))
rows = rng.choice(np.frombuffer(b"abcx0123 \n", np.uint8), size=(ns, sl)); mode = int(rng.integers(2))
got = rx.match(nfa, rows, mode=mode, kernel=rx.KERNEL_SYM_RES, group_lanes=16, events_cap=1 << 22)
print("events", got["n_events"], flush=True)
PY
grep "slot 10\|of slot 10\|leaves (" gpurun_out/r3_resdbg4.log | grep -v "^\[res\] wave [0-35-9]" | head -80
