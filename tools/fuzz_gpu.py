#!/usr/bin/env python3
"""Randomized differential test: every kernel variant vs the CPU oracle on seeded random automata and inputs.
    python tools/fuzz_gpu.py --seconds 300 --seed 1
Writes progress lines (so a long run is not taken for hung) and stops at the first mismatch with a repro seed."""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from nfa_util import blowup_nfa, late_blowup_nfa, random_nfa  # noqa: E402
from oracle import orx  # noqa: E402

rx = importlib.import_module("regex-fpga_amd")

KERNELS = [dict(kernel=rx.KERNEL_CSR_WAVE), dict(kernel=rx.KERNEL_SYM_WAVE)] + \
          [dict(kernel=rx.KERNEL_SYM_GROUP, group_lanes=g) for g in (1, 2, 4, 8, 16)] + \
          [dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=s) for s in (2, 4, 8, 11, 12, 13, 16, 20, 22, 24, 32, 48, 64)] + \
          [dict(kernel=rx.KERNEL_SYM_REG), dict(kernel=rx.KERNEL_SYM_REG), dict(kernel=rx.KERNEL_DFA), dict(kernel=rx.KERNEL_AUTO)]


def rand_regexes(rng, n):
    atoms = [b"a", b"b", b"c", b"[ab]", b"[^a]", b".", b"\\d", b"(ab|c)", b"x"]
    out = []
    for _ in range(n):
        parts = []
        for _ in range(int(rng.integers(1, 6))):
            a = atoms[int(rng.integers(len(atoms)))]
            q = [b"", b"", b"", b"*", b"+", b"?", b"{1,3}", b"{2}"][int(rng.integers(8))]
            parts.append(a + q)
        p = b"".join(parts)
        out.append(p)
    return out


def one_case(rng, trial):
    kind = int(rng.integers(6))
    if kind == 0:
        size = int(rng.integers(2, 600))
        alpha = int(rng.integers(2, 20))
        W, size = random_nfa(rng, size, max_deg=int(rng.integers(1, 30)), alphabet=alpha, dense_rows=int(rng.integers(0, 4)))
        nfa = rx.Nfa.from_words(W, size)  # (a last edge word of 0 — symbol 0 to state 0 — makes the size ambiguous without it)
    elif kind == 1:
        W, size = blowup_nfa(int(rng.integers(20, 400)))
        nfa = rx.Nfa.from_words(W, size)
        alpha = None
    elif kind == 2:
        W, size = late_blowup_nfa(int(rng.integers(20, 300)))
        nfa = rx.Nfa.from_words(W, size)
        alpha = None
    else:
        while True:
            try:
                nfa = rx.Nfa.compile(rand_regexes(rng, int(rng.integers(1, 12))), icase=bool(rng.integers(2)))
                break
            except rx.RxError:
                continue
        W, size = nfa.words, nfa.size
        alpha = None
    ns = int(rng.integers(1, 130))
    sl = int(rng.choice([0, 1, 3, 15, 16, 17, 31, 33, 64, 100, 255, 256, 257, 400, 1000]))
    if kind == 0:
        rows = rng.integers(0, alpha, size=(ns, sl), dtype=np.uint8)
    elif kind in (1, 2):
        rows = rng.choice(np.frombuffer(b"ABCDXYZab.", np.uint8), size=(ns, sl))
    else:
        rows = rng.choice(np.frombuffer(b"abcx0123 \n", np.uint8), size=(ns, sl))
    mode = int(rng.integers(2))
    CAP = 1 << 22
    ref = orx.match_batch(W, size, rows, mode=mode, want_match_count=True, events_cap=CAP)
    overflow = ref["n_events"] > CAP  # then only the counters are comparable (device order is arrival order)
    ks = [KERNELS[i] for i in rng.choice(len(KERNELS), size=5, replace=False)] + [KERNELS[-1]]
    for kern in ks:
        stats = bool(rng.integers(2))  # the statistics build and the plain build are different kernels (pruning, marks)
        # batches here are too small for the probe that normally decides: force the pruned / folded builds half the time
        flags = (rx.host.OPT_FORCE_PRUNE if rng.integers(2) else 0) | (rx.host.OPT_FORCE_FOLD if rng.integers(2) else 0) | \
            (rx.host.OPT_REG_NO_SKIP if rng.integers(2) else 0)
        got = rx.match(nfa, rows, mode=mode, want_match_count=True, collect_stats=stats, events_cap=CAP, flags=flags, **kern)
        ok = (got["n_events"] == ref["n_events"] and (overflow or np.array_equal(got["events"], ref["events"].astype(got["events"].dtype)))
              and np.array_equal(got["match_count"], ref["match_count"]) and np.array_equal(got["final_active"], ref["final_active"])
              and np.array_equal(got["anymatch"][:, :ref["anymatch"].shape[1]], ref["anymatch"])
              and (not stats or all(got["stats"][k] == ref["stats"][k] for k in ("sum_active", "sum_edges", "alg_bytes"))))
        if not ok:
            which = [k for k in ("events", "match_count", "final_active") if not np.array_equal(got[k], ref[k].astype(got[k].dtype))]
            which += [k for k in ("sum_active", "sum_edges", "alg_bytes", "n_events") if stats and got["stats"][k] != ref["stats"][k]]
            which.append(f"stats={stats}")
            if not np.array_equal(got["anymatch"][:, :ref["anymatch"].shape[1]], ref["anymatch"]):
                which.append("anymatch")
            bad = np.nonzero((got["final_active"] != ref["final_active"]).any(axis=1))[0][:8].tolist()
            return (f"MISMATCH trial {trial} kind {kind} size {size} ns {ns} sl {sl} mode {mode} kernel {kern} fields {which} "
                    f"bad final rows {bad} max_active {ref['stats']['max_active']} n_events {got['n_events']}/{ref['n_events']}")
    if rng.integers(4) == 0:  # every fourth case also with the final sets as lists (one-shot form of rx_plan_run)
        cap = int(max(64, np.unpackbits(ref["final_active"].view(np.uint8)).sum() + 1))
        ckw = [dict(), dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=int(rng.choice([4, 8, 13, 16, 32]))),
               dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=16, flags=rx.host.OPT_FORCE_FOLD)][int(rng.integers(3))]
        got = rx.match(nfa, rows, mode=mode, events_cap=CAP, compact_final=cap, **ckw)
        if got["final_states_overflow"] or not np.array_equal(rx.host.expand_final(got, nfa.nw64), ref["final_active"]) or \
                got["n_events"] != ref["n_events"]:
            return f"MISMATCH trial {trial} kind {kind} size {size} ns {ns} sl {sl} mode {mode}: final sets as lists (cap {cap})"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--first-trial", type=int, default=0)
    ap.add_argument("--n-trials", type=int, default=0)
    a = ap.parse_args()
    t0 = time.time()
    trial = a.first_trial
    last = t0
    while time.time() - t0 < a.seconds and (a.n_trials == 0 or trial < a.first_trial + a.n_trials):
        rng = np.random.default_rng([a.seed, trial])  # every case reproducible on its own
        err = one_case(rng, trial)
        if err:
            print(err, "seed", a.seed, flush=True)
            sys.exit(1)
        trial += 1
        if time.time() - last > 20:
            print(f"[fuzz] {trial} cases ok, {time.time() - t0:.0f} s", flush=True)
            last = time.time()
    print(f"[fuzz] done: {trial} cases, all kernels == oracle (seed {a.seed})", flush=True)


if __name__ == "__main__":
    main()
