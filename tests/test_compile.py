"""Regex list -> CSR automaton compiler (csrc/rx_compile.cpp).  The reference has no compiler, so this
step's parity is UNPINNED; the independent check is Python's `re` on the same patterns and inputs:
for every pattern and every byte position j, "some match of the pattern ends at j" must agree."""
import os
import re

import numpy as np
import pytest

from nfa_util import build_words


def end_positions(pat, text, icase=False, dotall=False):
    """Set of j such that a match of `pat` (bytes regex, maybe with leading ^) ends at text[j]."""
    flags = (re.I if icase else 0) | (re.S if dotall else 0)
    anchored = pat.startswith(b"^")
    body = pat[1:] if anchored else pat
    rx_ = re.compile(b"(?:" + body + b")\\Z", flags)
    out = set()
    for j in range(len(text)):
        m = rx_.match(text, 0, j + 1) if anchored else rx_.search(text, 0, j + 1)
        if m and m.end() == j + 1 and m.end() > m.start():
            out.add(j)
    return out


def nfa_end_positions(rx, orx, nfa, text):
    W = nfa.words
    r = orx.match_batch(W, nfa.size, np.frombuffer(text, np.uint8))
    per = {}
    for e in r["events"]:
        per.setdefault(nfa.accept_pattern(int(e["state"])), set()).add(int(e["k"]) - 1)
    return per


CASES = [
    (b"ab", {}), (b"a|bc|def", {}), (b"he(l+)o", dict(icase=True)), (b"a[0-9]{2,3}z", {}), (b"^GET +/", {}),
    (b"x.*y", {}), (b"x.*y", dict(dotall=True)), (b"[^a-c]b{2}", {}), (b"(ab|cd)*ef", {}), (b"\\d+\\.\\d+", {}),
    (b"\\x41\\x00?B", {}), (b"(?:foo|bar){1,2}baz", {}), (b"a?b?c", {}), (b"[\\w-]+@\\w+", {}), (b"q{3,}", {}),
    (b"(a|b)(c|d)?(e|f)+", {}), (b"\\s[A-Z][a-z]*\\s", {}), (b"a.{0,3}b", {}),
]


@pytest.mark.parametrize("pat,kw", CASES)
def test_single_patterns_vs_python_re(rx, orx, pat, kw):
    rng = np.random.default_rng(abs(hash(pat)) % (2**32))
    nfa = rx.Nfa.compile([pat], **kw)
    alphabet = np.frombuffer(b"abcdefxyzqGET /.0123456789ABhelo \n@-_B\x00\x41", np.uint8)
    for trial in range(6):
        text = rng.choice(alphabet, size=int(rng.integers(1, 90))).tobytes()
        want = end_positions(pat, text, **kw)
        got = nfa_end_positions(rx, orx, nfa, text).get(0, set())
        assert got == want, (pat, text)


def test_many_patterns_in_one_automaton(rx, orx):
    pats = [c[0] for c in CASES if not c[1]]
    nfa = rx.Nfa.compile(pats)
    assert nfa.n_accept >= len(pats)
    rng = np.random.default_rng(11)
    alphabet = np.frombuffer(b"abcdefxyzqGET /.0123456789ABhelo \n@-_", np.uint8)
    for trial in range(8):
        text = rng.choice(alphabet, size=200).tobytes()
        got = nfa_end_positions(rx, orx, nfa, text)
        for i, p in enumerate(pats):
            assert got.get(i, set()) == end_positions(p, text), (p, text)


def test_table_conventions_and_coe_round_trip(rx, orx, tmp_path):
    nfa = rx.Nfa.compile([b"/snort/i", b"^HTTP/1\\.[01]", b"a+b"])
    W = nfa.words
    size = nfa.size
    rp = W[:size + 1].astype(np.int64)
    col = W[size + 1:size + 1 + rp[size]]
    # state 0 and state 1 as in the shipped tables: 0 -> 1 on all bytes, 1 loops on all bytes
    row0 = col[rp[0]:rp[1]]
    assert {int(w >> 24) for w in row0 if (w & 0xFFFFFF) == 1} == set(range(256))
    row1 = col[rp[1]:rp[2]]
    assert {int(w >> 24) for w in row1 if (w & 0xFFFFFF) == 1} == set(range(256))
    assert not (col & 0xFFFFFF == 0).any()                       # state 0 is never re-entered
    for s in range(size):                                        # accept <=> empty row
        assert (rp[s] == rp[s + 1]) == (nfa.accept_pattern(s) >= 0)
    assert len(np.unique(col.astype(np.uint64) + (np.repeat(np.arange(size), np.diff(rp)).astype(np.uint64) << 32))) == len(col)
    p = str(tmp_path / "compiled.coe")
    nfa.save_coe(p)
    assert open(p).readline() == "memory_initialization_radix=16;\n"
    W2 = orx.load_coe(p)                                         # the oracle's parser reads it back
    assert np.array_equal(W2, W) and orx.infer_size(W2) == size
    again = rx.Nfa.load_coe(p)                                   # and so does the product loader (size inferred)
    assert again.size == size and np.array_equal(again.words, W)


def test_errors(rx):
    for bad in (b"a**b(", b"(ab", b"a{3,2}", b"[z-a]", b"a*", b"", b"x$", b"\\k", b"(?=a)b"):
        with pytest.raises(rx.RxError) as e:
            rx.Nfa.compile([bad])
        assert e.value.code in (-3, -1), bad
    with pytest.raises(rx.RxError):
        rx.Nfa.compile([b"a{1000}{1000}"])                        # expansion budget


def test_hostile_patterns_return_errors_instead_of_crashing(rx, orx):
    """Nothing throws or overflows the stack across the C boundary: deeply nested groups and quantifier towers are
    refused with RX_EFORMAT, and very long literals / alternations — sequences are built as balanced trees — compile."""
    for bad in (b"(" * 20000 + b"a" + b")" * 20000, b"(?:" * 5000 + b"a" + b")" * 5000,
                b"(?:(?:a{0,1000}b){0,1000}c){0,1000}", b"a" + b"?" * 3 + b"{1000}" * 3):
        with pytest.raises(rx.RxError) as e:
            rx.Nfa.compile([bad])
        assert e.value.code == -3, bad[:20]
    lit = bytes(np.random.default_rng(1).integers(97, 123, size=80000, dtype=np.uint8))   # one 80 000-byte literal
    nfa = rx.Nfa.compile([lit])
    assert nfa.size == 80000 + 2 and nfa.n_accept == 1
    alts = b"|".join(b"w%05d" % i for i in range(20000))                                 # 20 000 alternatives
    nfa = rx.Nfa.compile([alts])
    assert nfa.n_accept == 20000
    # the balanced tree is the same language as the left-deep one: spot-check against the oracle + Python
    pat = b"ab(c|d|e|f|g)h{2,4}i"
    W = rx.Nfa.compile([pat]).words
    size = orx.infer_size(W)
    data = np.frombuffer(b"xxabchhixabghhhhhi.abdhi", np.uint8)
    ends = sorted({m.end() for i in range(len(data)) for m in [re.compile(pat).match(data.tobytes(), i)] if m})
    got = orx.match_batch(W, size, data)
    assert sorted({int(e["k"]) for e in got["events"]}) == ends


def make_ruleset(n, seed=20261004):
    """Synthetic stand-in for a Snort-like ruleset (BASELINE configs[4]): the reference ships no rules and
    none can be fetched, so patterns are seeded random content strings with classes, gaps and case folding."""
    rng = np.random.default_rng(seed)
    words = [b"admin", b"passwd", b"select", b"union", b"cmd.exe", b"/etc/", b"script", b"GET ", b"POST ", b"User-Agent",
             b"Content-Length", b"%00", b"../", b"eval(", b"base64", b"login", b"root", b"shell", b"wget ", b"chmod "]
    pats = []
    for i in range(n):
        a = words[int(rng.integers(len(words)))]
        tail = bytes(rng.choice(np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789", np.uint8), size=int(rng.integers(3, 9))))
        kind = int(rng.integers(5))
        esc = re.escape(a)
        if kind == 0:
            p = esc + tail
        elif kind == 1:
            p = esc + b"[^\\n]{0,8}" + tail
        elif kind == 2:
            p = b"/" + esc + b"\\s*=\\s*" + tail + b"/i"
        elif kind == 3:
            p = esc + b"(" + tail[:3] + b"|" + tail[3:] + b"x)+\\d"
        else:
            p = tail + b"[0-9a-f]{4}" + esc
        pats.append(p)
    return pats


def test_ruleset_scale(rx, orx):
    pats = make_ruleset(700)
    nfa = rx.Nfa.compile(pats)
    assert 8000 < nfa.size < 20000          # "~10k states" class, like snort_16's 9 514
    text = b"GET /admin" + b"abc123 " + b"wget  =  " + b"zz" * 40
    r = orx.match_batch(nfa.words, nfa.size, np.frombuffer(text, np.uint8))
    assert r["stats"]["sum_active"] >= len(text)


@pytest.mark.gpu
def test_compiled_ruleset_on_gpu(rx, orx):
    """GPU kernels on a compiler-made ~10k-state automaton (configs[4] stand-in) == oracle."""
    pats = make_ruleset(700)
    nfa = rx.Nfa.compile(pats)
    rng = np.random.default_rng(5)
    frag = [b"GET /admin", b"passwd=", b"cmd.exe", b"../..", b"User-Agent: ", b"select x union", b"wget  = "] + [p[:12] for p in pats[:40]]
    rows = np.zeros((96, 512), np.uint8)
    for s in range(96):
        buf = bytearray()
        while len(buf) < 512:
            buf += frag[int(rng.integers(len(frag)))] if rng.random() < 0.5 else bytes(rng.integers(32, 127, size=int(rng.integers(1, 12)), dtype=np.uint8))
        rows[s] = np.frombuffer(bytes(buf[:512]), np.uint8)
    ref = orx.match_batch(nfa.words, nfa.size, rows, want_match_count=True)
    assert ref["n_events"] > 0
    for kern in (dict(kernel=rx.KERNEL_CSR_WAVE), dict(kernel=rx.KERNEL_SYM_WAVE), dict(kernel=rx.KERNEL_SYM_GROUP, group_lanes=4),
                 dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=16), dict(kernel=rx.KERNEL_AUTO)):
        got = rx.match(nfa, rows, want_match_count=True, collect_stats=True, **kern)
        assert got["n_events"] == ref["n_events"], kern
        assert np.array_equal(got["events"], ref["events"].astype(got["events"].dtype)), kern
        assert np.array_equal(got["final_active"], ref["final_active"]) and np.array_equal(got["match_count"], ref["match_count"])
        assert got["stats"]["alg_bytes"] == ref["stats"]["alg_bytes"]


@pytest.mark.gpu
def test_auto_probe_picks_the_kernel_by_active_set_size(rx, orx):
    """RX_KERNEL_AUTO probes the batch: the ruleset stand-in (about 14 active states per stream, bursts of 70, many targets
    per pass) goes to the pack kernel; an automaton that keeps 300 states active goes to the wavefront-per-stream slice
    kernel; snort_16 trace windows (2-3 active states) go to the pack kernel when the batch gives every SIMD more than four
    wavefronts, and to one wavefront per stream on the register kernel when it is smaller — with identical results."""
    wl = rx.workloads
    from nfa_util import blowup_nfa
    W, size = blowup_nfa(300)
    wide = rx.Nfa.from_words(W, size)
    rng = np.random.default_rng(7)
    wrows = rng.choice(np.array([0x41, 0x43, 0x43, 0x43], dtype=np.uint8), size=(512, 640))
    wref = orx.match_batch(W, size, wrows)
    wgot = rx.match(wide, wrows, collect_stats=True)
    assert rx.host.KERNEL_NAMES[wgot["stats"]["kernel_used"]] == "sym_wave"
    assert np.array_equal(wgot["final_active"], wref["final_active"]) and wgot["stats"]["alg_bytes"] == wref["stats"]["alg_bytes"]
    pats = wl.synthetic_ruleset()
    nfa = rx.Nfa.compile(pats)
    rows = wl.ruleset_traffic(pats, 640, 1024)
    ref = orx.match_batch(nfa.words, nfa.size, rows)
    got = rx.match(nfa, rows, collect_stats=True)
    assert rx.host.KERNEL_NAMES[got["stats"]["kernel_used"]] == "sym_pack"
    assert got["n_events"] == ref["n_events"] and np.array_equal(got["events"], ref["events"].astype(got["events"].dtype))
    assert np.array_equal(got["final_active"], ref["final_active"]) and got["stats"]["alg_bytes"] == ref["stats"]["alg_bytes"]
    for kern in (dict(kernel=rx.KERNEL_SYM_GROUP, group_lanes=16), dict(kernel=rx.KERNEL_SYM_GROUP, group_lanes=8),
                 dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=8), dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=2),
                 dict(kernel=rx.KERNEL_SYM_WAVE)):
        g2 = rx.match(nfa, rows, collect_stats=True, **kern)
        assert np.array_equal(g2["events"], got["events"]) and np.array_equal(g2["final_active"], got["final_active"]), kern
        assert g2["stats"]["alg_bytes"] == ref["stats"]["alg_bytes"], kern
    snort = rx.Nfa.load_coe(wl.SNORT_COE)
    lo, hi = rx.load_mem(wl.TRACES[("snort_16", "lo")]), rx.load_mem(wl.TRACES[("snort_16", "hi")])
    rows = wl.trace_windows(lo, hi, 640, 1024)
    got = rx.match(snort, rows)
    assert rx.host.KERNEL_NAMES[got["stats"]["kernel_used"]] == "sym_reg"      # small batch: latency per pass is what counts
    ref = orx.match_batch(snort.words, snort.size, rows)
    assert got["n_events"] == ref["n_events"] and np.array_equal(got["final_active"], ref["final_active"])
    big = rx.match(snort, wl.trace_windows(lo, hi, 8192, 1024))
    assert rx.host.KERNEL_NAMES[big["stats"]["kernel_used"]] == "sym_pack"
    assert np.array_equal(big["final_active"][:640], got["final_active"])
