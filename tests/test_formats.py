"""File formats of the reference: .coe tables and .mem traces.  CPU-only."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import DATA, GOLDEN

G = json.load(open(os.path.join(GOLDEN, "golden.json")))
S = json.load(open(os.path.join(GOLDEN, "survey_digests.json")))


def test_reference_files_are_verbatim():
    # sha256 of the reference's own files
    for f, h in G["files"].items():
        assert hashlib.sha256(open(os.path.join(DATA, f), "rb").read()).hexdigest() == h, f
    assert G["files"]["CSR_BlockMem_snort_16.coe"].startswith("5ee475f4")
    assert G["files"]["input_trace_hi_snort_16.mem"].startswith("412ff794")


@pytest.mark.parametrize("name", ["l7", "snort_16"])
def test_oracle_coe_decode_matches_survey(automata, name):
    W, size = automata[name]
    t = S["tables"][name]
    assert (W.size, size, int(W[size])) == (t["n_words"], t["size"], t["nnz"])
    assert hashlib.sha256(W.astype("<u4").tobytes()).hexdigest() == S["words_sha256"][name]
    rp = W[:size + 1].astype(np.int64)
    deg = np.diff(rp)
    assert (deg >= 0).all() and int((deg == 0).sum()) == t["n_accept"] and int(deg.max()) == t["max_degree"]
    col = W[size + 1:size + 1 + t["nnz"]]
    assert int((col & 0xFFFFFF).max()) < size
    assert not W[size + 1 + t["nnz"]:].any()  # 0-3 zero pad words


@pytest.mark.parametrize("name,coe", [("l7", "CSR_BlockMem.coe"), ("snort_16", "CSR_BlockMem_snort_16.coe")])
def test_product_loader_keeps_table_unchanged(rx, automata, name, coe):
    """librxmatch's own parser (csrc/rx_host.cpp) yields the same words the oracle's parser does,
    infers the same size, and reports the same shape facts."""
    W, size = automata[name]
    nfa = rx.Nfa.load_coe(os.path.join(DATA, coe))
    assert np.array_equal(nfa.words, W)
    t = S["tables"][name]
    assert (nfa.size, nfa.nnz, nfa.n_accept, nfa.max_degree, nfa.n_words) == (
        t["size"], t["nnz"], t["n_accept"], t["max_degree"], t["n_words"])
    nfa2 = rx.Nfa.from_words(W, size)  # explicit size, as the testbench passes size_range
    assert nfa2.size == size
    with pytest.raises(rx.RxError) as e:
        rx.Nfa.from_words(W, size + 1)
    assert e.value.code == -4


def test_mem_traces(rx, orx, traces):
    for (name, lh), b in traces.items():
        tag = {"l7": "l-7_filter", "snort_16": "snort_16"}[name]
        p = os.path.join(DATA, f"input_trace_{lh}_{tag}.mem")
        assert len(b) == (262144 if name == "l7" else 200000)
        assert np.array_equal(rx.load_mem(p), b)
    # first lines of input_trace_lo_snort_16.mem are c6 c6 7f 53 50
    assert traces[("snort_16", "lo")][:5].tolist() == [0xC6, 0xC6, 0x7F, 0x53, 0x50]


def test_malformed_inputs(rx, tmp_path):
    bad = tmp_path / "bad.coe"
    bad.write_text("memory_initialization_radix=16;\nmemory_initialization_vector=0000 1111;\n")
    with pytest.raises(rx.RxError) as e:
        rx.Nfa.load_coe(str(bad))
    assert e.value.code == -3
    bad.write_text("memory_initialization_radix=10;\nmemory_initialization_vector=" + "0" * 32 + ";\n")
    with pytest.raises(rx.RxError):
        rx.Nfa.load_coe(str(bad))
    with pytest.raises(rx.RxError) as e:
        rx.Nfa.load_coe(str(tmp_path / "missing.coe"))
    assert e.value.code == -2
    m = tmp_path / "t.mem"
    m.write_text("1\nff\n0\n7f\n")
    assert rx.load_mem(str(m)).tolist() == [1, 255, 0, 127]
    m.write_text("1\n100\n")
    with pytest.raises(rx.RxError):
        rx.load_mem(str(m))
    # comma-separated COE with a terminating ';' is standard and accepted
    W = [0, 1, 1, 0x61000001]  # size 2: state 0 --'a'--> state 1; state 1 accepts
    tok = "".join(f"{w:08x}" for w in W)
    ok = tmp_path / "ok.coe"
    ok.write_text(f"memory_initialization_radix=16;\nmemory_initialization_vector=\n{tok},\n" + "0" * 32 + ";\n")
    with pytest.raises(rx.RxError):  # a whole line of padding exceeds the 0-3 pad words rule
        rx.Nfa.load_coe(str(ok))
    ok.write_text(f"memory_initialization_radix=16;\nmemory_initialization_vector=\n{tok};\n")
    n = rx.Nfa.load_coe(str(ok))
    assert (n.size, n.nnz, n.n_accept) == (2, 1, 1)


def test_ambiguous_size_is_refused_not_guessed(rx, orx):
    """The .coe carries no state count (the testbench passes it as a parameter, testbench_BLK_Mem.sv:20).  When the last edge
    is (symbol 0 -> state 0) its word is 0 and reads like padding, and a second size fits the words: size inference says so
    (found by the GPU fuzzer, seed 777 case 9162) instead of picking one; with the size given the automaton loads."""
    # 4 states: 0 --a--> 1, 1 --NUL--> 0, states 2 and 3 without edges: row_ptr 0 1 2 2 2, edges (0x61, 1) (0x00, 0).
    # Read with size 3 the words are row_ptr 0 1 2 2, edges (0x00, 2) (0x61, 1) and one word of padding: valid as well.
    W = np.array([0, 1, 2, 2, 2, 0x61000001, 0x00000000], np.uint32)
    with pytest.raises(rx.RxError) as e:
        rx.Nfa.from_words(W)
    assert "size" in str(e.value)
    n = rx.Nfa.from_words(W, 4)
    assert (n.size, n.nnz) == (4, 2)
    assert rx.Nfa.from_words(W, 3).nnz == 2          # (the other reading, when the caller says so)
    ref = orx.match_batch(W, 4, np.frombuffer(b"a\x00a", np.uint8))
    assert ref["stats"]["n_passes"] == 4


def test_size_inference_rejects_garbage(rx):
    with pytest.raises(rx.RxError):
        rx.Nfa.from_words(np.array([5, 1, 2, 3], np.uint32))
    with pytest.raises(rx.RxError):  # target out of range
        rx.Nfa.from_words(np.array([0, 1, 1, 0x61000009], np.uint32), 2)
