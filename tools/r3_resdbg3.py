#!/usr/bin/env python3
"""Debug aid: one case of tools/fuzz_gpu.py (seed, trial; compiled-regex kinds) on the resident-entry kernel, event diff."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_gpu as f  # noqa: E402
from oracle import orx  # noqa: E402
rx = f.rx
seed, trial = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng([seed, trial])
kind = int(rng.integers(6))
assert kind >= 3
while True:
    try:
        nfa = rx.Nfa.compile(f.rand_regexes(rng, int(rng.integers(1, 12))), icase=bool(rng.integers(2)))
        break
    except rx.RxError:
        continue
W, size = nfa.words, nfa.size
ns = int(rng.integers(1, 130))
sl = int(rng.choice([0, 1, 3, 15, 16, 17, 31, 33, 64, 100, 255, 256, 257, 400, 1000]))
rows = rng.choice(np.frombuffer(b"abcx0123 \n", np.uint8), size=(ns, sl))
mode = int(rng.integers(2))
ref = orx.match_batch(W, size, rows, mode=mode, want_match_count=True, events_cap=1 << 22)
re_ = set(map(tuple, np.asarray(ref["events"]).tolist()))
for S in (8, 16, 24, 32, 48):
    for rep in range(3):
        got = rx.match(nfa, rows, mode=mode, kernel=rx.KERNEL_SYM_RES, group_lanes=S, events_cap=1 << 22, flags=4 if rep == 0 else 0)
        ge = set(map(tuple, np.asarray(got["events"]).tolist()))
        print("S", S, "rep", rep, "events", got["n_events"], ref["n_events"], "extra", sorted(ge - re_)[:4], "missing", sorted(re_ - ge)[:4], flush=True)
