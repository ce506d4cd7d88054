#!/usr/bin/env python3
"""What would activity-sorted batching buy on workload T?  Times the quiet half (lo-trace windows) and the busy half
(hi-trace windows) of configs[2] separately, each with its best kernel variant, against the mixed batch."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
rx = importlib.import_module("regex-fpga_amd")
wl = rx.workloads
H = rx.host
nfa = rx.Nfa.load_coe(wl.SNORT_COE)
hi, lo = rx.load_mem(wl.TRACES[("snort_16", "hi")]), rx.load_mem(wl.TRACES[("snort_16", "lo")])
ns, sl = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 1024
rows = wl.trace_windows(lo, hi, ns, sl)
def t(data, **kw):
    p = rx.Plan(nfa, data.shape[0], sl, device=0, events_cap=1 << 22, **kw)
    p.upload(np.ascontiguousarray(data)); p.launch(); p.sync(); p.kernel_times()
    for _ in range(6): p.launch()
    n, s, mn, mx = p.kernel_times()
    r = p.download(); p.close()
    return s / n, r["stats"]["variant"]
print("mixed AUTO", t(rows, kernel=rx.KERNEL_AUTO))
for name, part in (("quiet (lo windows)", rows[0::2]), ("busy (hi windows)", rows[1::2])):
    for label, kw in (("auto", dict(kernel=rx.KERNEL_AUTO)),
                      ("S8", dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=8)), ("S11", dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=11)),
                      ("S13", dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=13)), ("S16", dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=16)),
                      ("S24", dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=24)), ("S32", dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=32)),
                      ("S16f", dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=16, flags=H.OPT_FORCE_FOLD)),
                      ("S32f", dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=32, flags=H.OPT_FORCE_FOLD)),
                      ("S64f", dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=64, flags=H.OPT_FORCE_FOLD))):
        ms, var = t(part, **kw)
        print(f"{name:20s} {label:5s} {var:10s} {ms:.4f} ms", flush=True)
