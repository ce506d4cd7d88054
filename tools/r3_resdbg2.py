#!/usr/bin/env python3
"""Debug aid: replays tests/test_gpu_parity.py::test_automata_in_the_reference_convention for the resident-entry kernel and
prints the first differing events of every failing (trial, S)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from nfa_util import convention_nfa  # noqa: E402
from oracle import orx  # noqa: E402
rx = importlib.import_module("regex-fpga_amd")
rng = np.random.default_rng(1711)
for trial in range(30):
    alpha = int(rng.integers(2, 10))
    Wc, sz = convention_nfa(rng, int(rng.integers(4, 200)), alphabet=alpha, n_first=int(rng.integers(1, 5)))
    nfa = rx.Nfa.from_words(Wc, sz)
    ns, sl = int(rng.integers(1, 70)), int(rng.integers(0, 260))
    rows = rng.integers(0, alpha, size=(ns, sl), dtype=np.uint8)
    mode = int(trial & 1)
    ref = orx.match_batch(Wc, sz, rows, mode=mode)
    for S in (8, 16, 24, 32, 48):
        got = rx.match(nfa, rows, mode=mode, kernel=rx.KERNEL_SYM_RES, group_lanes=S)
        ge = set(map(tuple, np.asarray(got["events"]).tolist())) if got["n_events"] else set()
        re_ = set(map(tuple, np.asarray(ref["events"]).tolist())) if ref["n_events"] else set()
        fin_bad = [s for s in range(ns) if not np.array_equal(got["final_active"][s], ref["final_active"][s])]
        if got["n_events"] != ref["n_events"] or ge != re_ or fin_bad:
            print("trial", trial, "size", sz, "alpha", alpha, "streams", ns, "len", sl, "mode", mode, "S", S, "kernel", got["stats"]["kernel_used"],
                  "events", got["n_events"], ref["n_events"], "launches", got["stats"]["n_launches"], flush=True)
            ev = np.asarray(got["events"])
            from collections import Counter
            cnt = Counter(map(tuple, ev.tolist()))
            dups = [(e, c) for e, c in cnt.items() if c > 1]
            print("  duplicated events:", sorted(dups)[:8])
            print("  extra:", sorted(ge - re_)[:8], "missing:", sorted(re_ - ge)[:8], "final sets wrong:", fin_bad[:8])
print("done", flush=True)
