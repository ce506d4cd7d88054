#!/usr/bin/env python3
"""Debug aid: the resident-entry kernel against the oracle on the T batch of the parity test; which final sets differ."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orx  # noqa: E402
rx = importlib.import_module("regex-fpga_amd")
W = orx.load_coe(os.path.join(ROOT, "data", "CSR_BlockMem_snort_16.coe")); size = orx.infer_size(W)
lo = orx.load_mem(os.path.join(ROOT, "data", "input_trace_lo_snort_16.mem")); hi = orx.load_mem(os.path.join(ROOT, "data", "input_trace_hi_snort_16.mem"))
rows = rx.workloads.trace_windows(lo, hi, 1536, 1024, first=1000)
ref = orx.match_batch(W, size, rows)
nfa = rx.Nfa.from_words(W, size)
def bits(row):
    out = []
    for wi, w in enumerate(row):
        w = int(w)
        while w:
            b = (w & -w).bit_length() - 1; out.append(wi * 64 + b); w &= w - 1
    return out
for S in [int(x) for x in sys.argv[1:]] or [32, 48]:
    got = rx.match(nfa, rows, kernel=rx.KERNEL_SYM_RES, group_lanes=S)
    print("S", S, "kernel", got["stats"]["kernel_used"], "events", got["n_events"], ref["n_events"], "spilled", got["stats"].get("spilled_streams"), flush=True)
    bad = [s for s in range(rows.shape[0]) if not np.array_equal(got["final_active"][s], ref["final_active"][s])]
    print(" streams with a wrong final set:", len(bad), bad[:20])
    for s in bad[:6]:
        g, r = set(bits(got["final_active"][s])), set(bits(ref["final_active"][s]))
        print("  stream", s, "slot", s % S, "wave", s // S, "missing", sorted(r - g), "extra", sorted(g - r), "size", len(r))
