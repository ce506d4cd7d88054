#!/usr/bin/env python3
"""Host-to-host timing of rx_plan_run vs upload+launch+download (config 2 shape)."""
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
for want_final in (True, False):
    p = rx.Plan(nfa, ns, sl, device=0, events_cap=1 << 22, want_final=want_final, flags=rx.host.OPT_VERBOSE)
    p.run(rows)
    for rep in range(2):
        t = time.perf_counter(); r = p.run(rows); dt = time.perf_counter() - t
        print(f"run want_final={want_final}: {dt*1e3:.2f} ms = {8*ns*sl/dt/1e9:.1f} Gbit/s, events {r['stats']['n_events']}", flush=True)
    p.upload(rows); p.launch(); p.download()
    t = time.perf_counter(); p.upload(rows); t1 = time.perf_counter(); p.launch(); p.sync(); t2 = time.perf_counter(); r = p.download(); t3 = time.perf_counter()
    print(f"serial want_final={want_final}: upload {1e3*(t1-t):.2f} kernel {1e3*(t2-t1):.2f} download {1e3*(t3-t2):.2f} ms", flush=True)
    p.close()
