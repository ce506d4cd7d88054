#!/usr/bin/env python3
"""What the final sets cost the pack kernel on T (or, with argument R, on the rule-set stand-in): kernel time with bitmask rows / compact lists / none, any-match on / off."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oracle import orx  # noqa: E402  (only the .coe / .mem readers)
rx = importlib.import_module("regex-fpga_amd")
wl = rx.workloads
W = orx.load_coe(wl.SNORT_COE); size = orx.infer_size(W)
lo = orx.load_mem(wl.TRACES[("snort_16", "lo")]); hi = orx.load_mem(wl.TRACES[("snort_16", "hi")])
ns, sl = 65536, 1024
if len(sys.argv) > 1 and sys.argv[1] == "R":  # the rule-set stand-in instead of snort_16 / T
    pats = wl.synthetic_ruleset()
    nfa = rx.Nfa.compile(pats)
    rows = wl.ruleset_traffic(pats, ns, sl, workers=8)
else:
    nfa = rx.Nfa.from_words(W, size)
    rows = wl.trace_windows(lo, hi, ns, sl)
d = torch.from_numpy(rows).to("cuda:0")
for label, kw in (("rows + any-match", dict(want_final=True, want_anymatch=True)), ("rows only", dict(want_final=True, want_anymatch=False)),
                  ("any-match only", dict(want_final=False, want_anymatch=True)), ("events only", dict(want_final=False, want_anymatch=False))):
    for rep in range(2):
        p = rx.Plan(nfa, ns, sl, mode=rx.MODE_FULL, device=0, events_cap=1 << 22, want_match_count=False, **kw)
        p.set_device_input(d.data_ptr(), ns, sl, sl, keepalive=d)
        for _ in range(3):
            p.launch()
        p.kernel_times()
        for _ in range(20):
            p.launch()
        n, s, mn, mx = p.kernel_times()
        print(f"{label:18s} rep {rep}: kernel {s / n:.4f} ms (min {mn:.4f})", flush=True)
        p.close()
