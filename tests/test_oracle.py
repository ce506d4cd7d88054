"""The CPU oracle itself: known answers, golden digests, and agreement of its two independent
restatements of the reference (functional rx_oracle.c vs clock-accurate rx_cycle.c).  CPU-only."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from nfa_util import blowup_nfa, build_words, kat_ab, random_nfa

G = json.load(open(os.path.join(GOLDEN, "golden.json")))
S = json.load(open(os.path.join(GOLDEN, "survey_digests.json")))
N = 200000


def test_kat_ab(orx):
    """"xabab" -> S = {0},{1},{1,2},{1,3},{1,2},{1,3}; M_3 = M_5 = {3}."""
    W, size = kat_ab()
    assert W.size == 520 and orx.infer_size(W) == 4
    data = np.frombuffer(b"xabab", np.uint8)
    full = orx.match_batch(W, size, data, mode=orx.MODE_FULL, want_match_count=True)
    assert [(int(e["k"]), int(e["state"])) for e in full["events"]] == [(3, 3), (5, 3)]
    assert full["match_count"][0].tolist() == [0, 0, 0, 2]
    assert orx.bits_to_states(full["final_active"][0]) == [1, 3]
    assert full["anymatch"][0, 0] == (1 << 3) | (1 << 5)
    tb = orx.match_batch(W, size, data, mode=orx.MODE_TB_COMPAT, want_match_count=True)
    assert tb["match_count"][0].tolist() == [0, 0, 0, 1]  # the testbench never sees M_5
    assert tb["stats"]["n_passes"] == 4
    # per-pass accounting: S_0..S_4 consume bytes in full mode: rows 257,257,257+1,257+0,257+1
    assert full["stats"]["sum_active"] == 1 + 1 + 2 + 2 + 2
    assert full["stats"]["sum_edges"] == 257 * 5 + 1 + 0 + 1


@pytest.mark.parametrize("key", sorted(S["tb_compat"]))
def test_shipped_runs_match_survey_digests(orx, automata, traces, key):
    """The 4 shipped automaton x trace runs, tb-compat: digests transcribed from SURVEY App. D."""
    name, lh = key.split(":")
    W, size = automata[name]
    r = orx.match_batch(W, size, traces[(name, lh)][:N], mode=orx.MODE_TB_COMPAT, nthreads=1, want_match_count=True)
    s = S["tb_compat"][key]
    assert r["n_events"] == s["events"]
    assert r["stats"]["sum_active"] == s["sum_active"] and r["stats"]["sum_edges"] == s["sum_edges"]
    assert 8 * r["stats"]["sum_active"] + 4 * r["stats"]["sum_edges"] == s["csr_bytes"]
    assert orx.h_match_count(r["match_count"][0]) == s["H_mc"]
    assert orx.h_events(r["events"]) == s["H_ev"]
    fe = [[int(e["k"]), int(e["state"])] for e in r["events"][:len(s["first_events"])]]
    assert fe == s["first_events"]
    if "final_active" in s:
        assert orx.bits_to_states(r["final_active"][0]) == s["final_active"]
    # and the committed golden file (regression pin for the full-mode outputs too)
    g = G["tb_compat"][key]
    assert (r["n_events"], orx.h_events(r["events"]), r["stats"]["alg_bytes"]) == (g["n_events"], g["H_ev"], g["alg_bytes"])


@pytest.mark.parametrize("key", sorted(G["full"]))
def test_full_mode_goldens(orx, automata, traces, key):
    name, lh = key.split(":")
    W, size = automata[name]
    r = orx.match_batch(W, size, traces[(name, lh)][:N], mode=orx.MODE_FULL, nthreads=1, want_match_count=True)
    g = G["full"][key]
    assert (r["n_events"], orx.h_events(r["events"]), orx.h_match_count(r["match_count"][0])) == (
        g["n_events"], g["H_ev"], g["H_mc"])
    assert orx.bits_to_states(r["final_active"][0]) == g["final_active"]
    assert r["n_events"] >= G["tb_compat"][key]["n_events"]


@pytest.mark.parametrize("name", ["l7", "snort_16"])
def test_clock_model_reproduces_testbench_totals(orx, automata, traces, name):
    """Blk_Mem_tb on the shipped lo+hi pair: "Total no. cycles" and both match_count arrays from the
    clock-accurate restatement equal the survey's predictions and the functional model."""
    W, size = automata[name]
    lo, hi = traces[(name, "lo")][:N + 1], traces[(name, "hi")][:N + 1]
    c = orx.tb_cycle(W, size, lo, hi, N, skip_idle=True)
    assert not c["hung"] and c["passes"] == N - 1
    assert c["total_cycles"] == S["total_cycles"][name] == G["cycles"][name]["total_cycles"]
    assert orx.h_match_count(c["match_count"]) == S["tb_compat"][f"{name}:lo"]["H_mc"]
    assert orx.h_match_count(c["match_count_2"]) == S["tb_compat"][f"{name}:hi"]["H_mc"]
    assert orx.predict_cycles(W, size, lo, hi, N - 1) == c["total_cycles"]
    # events of stream 1 / 2 in pulse order == functional events of lo / hi
    for sid, lh in ((0, "lo"), (1, "hi")):
        ev = c["events"][c["events"]["stream"] == sid]
        f = orx.match_batch(W, size, traces[(name, lh)][:N], mode=orx.MODE_TB_COMPAT, nthreads=1)
        assert np.array_equal(ev["k"], f["events"]["k"]) and np.array_equal(ev["state"], f["events"]["state"])


@pytest.mark.parametrize("name", ["l7", "snort_16"])
def test_idle_fast_forward_is_exact(orx, automata, traces, name):
    W, size = automata[name]
    lo, hi = traces[(name, "lo")], traces[(name, "hi")]
    a = orx.tb_cycle(W, size, lo, hi, 600, skip_idle=False)
    b = orx.tb_cycle(W, size, lo, hi, 600, skip_idle=True)
    assert a["total_cycles"] == b["total_cycles"] and a["passes"] == b["passes"] == 599
    assert np.array_equal(a["events"], b["events"]) and np.array_equal(a["event_cycles"], b["event_cycles"])
    assert a["total_cycles"] == orx.predict_cycles(W, size, lo, hi, 599)


@pytest.mark.parametrize("name", ["l7", "snort_16"])
def test_rtl_pipeline_compares_exactly_the_csr_row(orx, automata, name):
    """For EVERY state: with only that state active, the restated 3-deep line pipeline of FPGA.v
    compares exactly the words size+1+row_ptr[i] .. +deg-1, each once (latency-1 ROM)."""
    W, size = automata[name]
    rp = W[:size + 1].astype(np.int64)
    for i in range(size):
        p = orx.probe_row(W, size, i, 0x61)
        assert p["rc"] == 0
        want = np.arange(size + 1 + rp[i], size + 1 + rp[i + 1])
        assert np.array_equal(np.sort(p["addrs"]), want), i
        assert p["accepted"] == (rp[i] == rp[i + 1])
        deg = int(rp[i + 1] - rp[i])
        a = size + 1 + int(rp[i])
        nl = ((a + deg - 1) >> 2) - (a >> 2) + 1
        cost = 3 + (1 if (i & 3) == 3 else 0) + (1 if deg == 0 else nl + 2) + 1
        assert p["clocks"] == size + cost - 1, i  # SURVEY §3.2 cost formula


def test_rom_latency_must_be_one(orx, automata, traces):
    """With a 2-clock ROM the design reads stale lines: it hangs on l7 and mis-matches on snort_16."""
    W, size = automata["l7"]
    r = orx.tb_cycle(W, size, traces[("l7", "lo")], traces[("l7", "hi")], 50, bram_latency=2, skip_idle=False,
                     max_cycles=3_000_000)
    assert r["hung"]
    W, size = automata["snort_16"]
    lo, hi = traces[("snort_16", "lo")], traces[("snort_16", "hi")]
    good = orx.tb_cycle(W, size, lo, hi, 400, skip_idle=False)
    bad = orx.tb_cycle(W, size, lo, hi, 400, bram_latency=2, skip_idle=False, max_cycles=good["total_cycles"] * 4)
    assert bad["hung"] or bad["total_cycles"] != good["total_cycles"] or not np.array_equal(bad["events"], good["events"])


def test_two_restatements_agree_on_random_automata(orx):
    """Functional vs clock-accurate model on seeded random NFAs and byte streams (both streams)."""
    rng = np.random.default_rng(20261004)
    for trial in range(60):
        size = int(rng.integers(2, 90))
        W, size = random_nfa(rng, size, max_deg=int(rng.integers(1, 12)), alphabet=int(rng.integers(2, 9)))
        n = int(rng.integers(2, 120))
        lo = rng.integers(0, 8, size=n, dtype=np.uint8)
        hi = rng.integers(0, 8, size=n, dtype=np.uint8)
        c = orx.tb_cycle(W, size, lo, hi, n, skip_idle=bool(trial & 1))
        assert not c["hung"]
        for sid, b in ((0, lo), (1, hi)):
            f = orx.match_batch(W, size, b, mode=orx.MODE_TB_COMPAT, nthreads=1, want_match_count=True)
            ev = c["events"][c["events"]["stream"] == sid]
            assert np.array_equal(ev["k"], f["events"]["k"]) and np.array_equal(ev["state"], f["events"]["state"]), trial
            mc = c["match_count"] if sid == 0 else c["match_count_2"]
            assert np.array_equal(mc, f["match_count"][0])
        assert c["total_cycles"] == orx.predict_cycles(W, size, lo, hi, n - 1)


def test_batch_threading_and_edge_cases(orx, automata, traces):
    W, size = automata["snort_16"]
    hi = traces[("snort_16", "hi")]
    rows = np.stack([hi[i * 300:i * 300 + 700] for i in range(37)])
    a = orx.match_batch(W, size, rows, nthreads=1, want_match_count=True)
    b = orx.match_batch(W, size, rows, nthreads=5, want_match_count=True)
    for k in ("events", "match_count", "match_count_total", "anymatch", "final_active"):
        assert np.array_equal(a[k], b[k]), k
    assert a["stats"] == b["stats"] and a["n_events"] > 0
    assert np.array_equal(a["match_count"].sum(0), a["match_count_total"])
    # empty stream: full mode = 1 pass (S_0 only), tb-compat = 0 passes
    e = orx.match_batch(W, size, np.zeros((3, 0), np.uint8))
    assert e["stats"]["n_passes"] == 1 and e["n_events"] == 0
    assert [orx.bits_to_states(r) for r in e["final_active"]] == [[0]] * 3
    assert orx.n_passes(0, orx.MODE_TB_COMPAT) == 0 and orx.n_passes(1, orx.MODE_TB_COMPAT) == 0
    # events beyond the cap are counted but not written
    c = orx.match_batch(W, size, rows, events_cap=5)
    assert c["n_events"] == a["n_events"] and len(c["events"]) == 5
    assert np.array_equal(c["events"], a["events"][:5])


def test_chunk_chaining(orx, automata, traces):
    """final_active of one chunk as init_active of the next == one uninterrupted run."""
    W, size = automata["snort_16"]
    hi = traces[("snort_16", "hi")][:6000]
    whole = orx.match_batch(W, size, hi, nthreads=1)
    cut = 2500
    a = orx.match_batch(W, size, hi[:cut], nthreads=1)
    b = orx.match_batch(W, size, hi[cut:], nthreads=1, init_active=a["final_active"])
    ev_a = a["events"][a["events"]["k"] < cut]            # pass `cut` of chunk 1 is pass 0 of chunk 2
    ev_b = b["events"].copy()
    ev_b["k"] += cut
    assert np.array_equal(np.concatenate([ev_a, ev_b]), whole["events"])
    assert np.array_equal(b["final_active"], whole["final_active"])


def test_blowup_automaton(orx):
    W, size = blowup_nfa(300)
    data = np.frombuffer(b"\x43\x41\x41\x43\x42\x43", np.uint8)
    r = orx.match_batch(W, size, data, want_match_count=True)
    assert r["stats"]["sum_active"] == 1 + 300 * 4 + 1  # S_5 = {accept} only
    assert r["n_events"] == 1 and int(r["events"][0]["k"]) == 5
