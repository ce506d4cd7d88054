#!/usr/bin/env python3
"""One shipped snort_16 trace as ONE stream on the register kernel (for rocprofv3 passes).
usage: python3 tools/single_one.py hi|lo [launches] [flags]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
rx = importlib.import_module("regex-fpga_amd")
wl = rx.workloads
which = sys.argv[1] if len(sys.argv) > 1 else "hi"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
nfa = rx.Nfa.load_coe(wl.SNORT_COE)
trace = rx.load_mem(wl.TRACES[("snort_16", which)])[:200000]
p = rx.Plan(nfa, 1, 200000, mode=rx.MODE_TB_COMPAT, device=0, events_cap=1 << 20, kernel=rx.KERNEL_SYM_REG, flags=flags)
p.upload(trace[None, :])
for _ in range(reps):
    p.launch()
p.sync()
n, s, mn, mx = p.kernel_times()
print(f"{which} reg {mn:.3f} ms {mn * 2.4e6 / 199999:.0f} cyc/pass", flush=True)
p.close()
