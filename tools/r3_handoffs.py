#!/usr/bin/env python3
"""How many streams the pack kernel hands to the wave kernel on the rule-set stand-in (verbose output of ONE launch per choice)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
rx = importlib.import_module("regex-fpga_amd")
wl = rx.workloads
pats = wl.synthetic_ruleset()
nfa = rx.Nfa.compile(pats)
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rows = wl.ruleset_traffic(pats, ns, 1024, workers=8)
H = rx.host
for S, fl in ((8, H.OPT_FORCE_PRUNE), (4, H.OPT_FORCE_PRUNE), (13, H.OPT_FORCE_PRUNE), (8, 0)):
    p = rx.Plan(nfa, ns, 1024, device=0, events_cap=1 << 22, kernel=rx.KERNEL_SYM_PACK, group_lanes=S, flags=fl | H.OPT_VERBOSE)
    p.upload(rows)
    for _ in range(3):
        p.launch()
    n, s, mn, mx = p.kernel_times()
    print(f"S={S} flags={fl}: kernel {s / n:.3f} ms", flush=True)
    sys.stderr.flush()
    r = p.download()
    print("   events", r["n_events"], flush=True)
