#!/usr/bin/env python3
"""The hand-off mix of bench.py (one stream in 64 trapped) on its own, for rocprofv3 --kernel-trace --stats."""
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
tw, tsize = wl.table_with_trap(W, size)
nfa = rx.Nfa.from_words(tw, tsize)
rows = wl.handoff_mix(lo, hi, ns, sl)
d = torch.from_numpy(rows).to("cuda:0")
p = rx.Plan(nfa, ns, sl, mode=rx.MODE_FULL, device=0, events_cap=1 << 22, flags=int(sys.argv[1]) if len(sys.argv) > 1 else 0)
p.set_device_input(d.data_ptr(), ns, sl, sl, keepalive=d)
for _ in range(6):
    p.launch()
n, s, mn, mx = p.kernel_times()
print("mix: kernel bracket", round(s / n, 4), "ms", flush=True)
