#!/usr/bin/env python3
"""rx_plan_run with the final sets as bitmask rows, as compact lists, and without them (config 2 shape)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
rx = importlib.import_module("regex-fpga_amd")
wl = rx.workloads
nfa = rx.Nfa.load_coe(wl.SNORT_COE)
hi, lo = rx.load_mem(wl.TRACES[("snort_16", "hi")]), rx.load_mem(wl.TRACES[("snort_16", "lo")])
ns, sl = 65536, 1024
rows = wl.trace_windows(lo, hi, ns, sl)
p = rx.Plan(nfa, ns, sl, device=0, events_cap=1 << 22, want_final=True, flags=rx.host.OPT_VERBOSE if len(sys.argv) > 1 else 0)
for label, kw in (("rows", {}), ("compact", dict(compact_final=1 << 22)), ("compact small cap", dict(compact_final=1 << 19))):
    p.run(rows, **kw); p.run(rows, **kw)
    best = 1e9
    for rep in range(4):
        t = time.perf_counter(); r = p.run(rows, **kw); best = min(best, time.perf_counter() - t)
    print(f"{label}: {best*1e3:.2f} ms = {8*ns*sl/best/1e9:.1f} Gbit/s, kernel_ms {r['stats']['kernel_ms']:.3f}", flush=True)
p.close()
