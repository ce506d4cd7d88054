#!/usr/bin/env python3
"""BASELINE configs[1] timing: the shipped snort_16 hi (and lo) trace as ONE stream, tb-compat, per kernel variant.
usage: python tools/single.py [n_streams]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
rx = importlib.import_module("regex-fpga_amd")
wl = rx.workloads
nfa = rx.Nfa.load_coe(wl.SNORT_COE)
hi, lo = rx.load_mem(wl.TRACES[("snort_16", "hi")]), rx.load_mem(wl.TRACES[("snort_16", "lo")])
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H = rx.host
for name, trace in (("hi", hi), ("lo", lo)):
    rows = np.stack([trace[:200000]] * ns)
    for label, kw in (("auto", dict(kernel=rx.KERNEL_AUTO)), ("reg+fold", dict(kernel=rx.KERNEL_SYM_REG, flags=H.OPT_VERBOSE)),
                      ("reg", dict(kernel=rx.KERNEL_SYM_REG, flags=H.OPT_NO_FOLD | H.OPT_VERBOSE)),
                      ("pack16", dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=16)),
                      ("pack8+fold", dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=8, flags=H.OPT_FORCE_FOLD)),
                      ("wave", dict(kernel=rx.KERNEL_SYM_WAVE))):
        p = rx.Plan(nfa, ns, rows.shape[1], mode=rx.MODE_TB_COMPAT, device=0, events_cap=1 << 20, **kw)
        p.upload(rows)
        p.launch()
        p.sync()
        p.kernel_times()
        for _ in range(3):
            p.launch()
        n, s, mn, mx = p.kernel_times()
        r = p.download()
        print(f"{name} x{ns} {label:12s} {mn:8.3f} ms  {mn * 1e6 / 199999:7.1f} ns/pass  {mn * 2.4e6 / 199999:7.0f} cyc@2.4GHz/pass  events {r['stats']['n_events']} "
              f"kernel {H.KERNEL_NAMES[r['stats']['kernel_used']]} {r['stats']['variant']}", flush=True)
        p.close()
