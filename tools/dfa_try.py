#!/usr/bin/env python3
"""Quick check of the lazy-DFA kernel: parity vs oracle on small cases, then cold/warm timing on config 3."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import orx
from nfa_util import kat_ab, late_blowup_nfa, blowup_nfa
rx = importlib.import_module("regex-fpga_amd")
wl = rx.workloads

def check(nfa, W, size, rows, mode, tag):
    ref = orx.match_batch(W, size, rows, mode=mode, want_match_count=True, events_cap=1 << 22)
    got = rx.match(nfa, rows, mode=mode, kernel=rx.KERNEL_DFA, want_match_count=True, collect_stats=True, events_cap=1 << 22)
    ok = (got["n_events"] == ref["n_events"] and np.array_equal(got["events"], ref["events"].astype(got["events"].dtype))
          and np.array_equal(got["final_active"], ref["final_active"]) and np.array_equal(got["match_count"], ref["match_count"])
          and np.array_equal(got["anymatch"][:, :ref["anymatch"].shape[1]], ref["anymatch"])
          and all(got["stats"][k] == ref["stats"][k] for k in ("sum_active", "sum_edges", "alg_bytes")))
    print(tag, "OK" if ok else "MISMATCH", got["n_events"], ref["n_events"], nfa.dfa_info(0), flush=True)
    return ok

W, size = kat_ab(); nfa = rx.Nfa.from_words(W)
assert check(nfa, W, size, np.frombuffer(b"xabab", np.uint8)[None, :], 0, "kat full")
assert check(nfa, W, size, np.frombuffer(b"xabab", np.uint8)[None, :], 1, "kat tb")
W, size = late_blowup_nfa(120); nfa = rx.Nfa.from_words(W)
rows = np.frombuffer((b"xabxab..abZYYBabYab" * 6)[:96], np.uint8)[None, :].repeat(70, 0).copy()
rows[::3, 10:16] = np.frombuffer(b"abxabx", np.uint8)
assert check(nfa, W, size, rows, 0, "late blowup (EXIT path)")
W = orx.load_coe(wl.SNORT_COE); size = orx.infer_size(W)
snort = rx.Nfa.load_coe(wl.SNORT_COE)
lo, hi = rx.load_mem(wl.TRACES[("snort_16", "lo")]), rx.load_mem(wl.TRACES[("snort_16", "hi")])
rows = wl.trace_windows(lo, hi, 700, 1024, first=300)
assert check(snort, W, size, rows, 0, "snort T 700x1024 full")
assert check(snort, W, size, rows[:, :333], 1, "snort T 700x333 tb")
assert check(snort, W, size, np.stack([lo[:20000], hi[:20000]]), 1, "snort shipped prefix pair")
# timing on config 3
import torch
rows = wl.trace_windows(lo, hi, 65536, 1024)
d = torch.from_numpy(rows).cuda()
snort.dfa_reset(0)
plan = rx.Plan(snort, 65536, 1024, kernel=rx.KERNEL_DFA, device=0, events_cap=1 << 22)
plan.set_device_input(d.data_ptr(), 65536, 1024, 1024, keepalive=d)
for i in range(6):
    plan.launch(); ms = plan.sync()
    print(f"launch {i}: {ms:.3f} ms  -> {8*65536*1024/ms/1e6:.1f} Gbit/s  cache {snort.dfa_info(0)}", flush=True)
res = plan.download()
ref = orx.match_batch(W, size, rows[:4096], events_cap=1 << 22)
ev = res["events"]; ev = ev[ev["stream"] < 4096]
print("config3 sample parity:", np.array_equal(ev, ref["events"].astype(ev.dtype)), np.array_equal(res["final_active"][:4096], ref["final_active"]), res["n_events"])
# fresh streams with the warm cache
rows2 = wl.trace_windows(lo, hi, 65536, 1024, first=65536)
d2 = torch.from_numpy(rows2).cuda()
plan.set_device_input(d2.data_ptr(), 65536, 1024, 1024, keepalive=d2)
plan.launch(); ms = plan.sync(); print(f"fresh streams, warm cache: {ms:.3f} ms {8*65536*1024/ms/1e6:.1f} Gbit/s cache {snort.dfa_info(0)}")
plan.launch(); ms = plan.sync(); print(f"again: {ms:.3f} ms")
