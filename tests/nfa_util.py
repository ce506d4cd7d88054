"""Helpers to build CSR tables in the reference's word layout for tests."""
import numpy as np


def build_words(size, edges):
    """edges: iterable of (src, symbol, dst).  Row order = order given per src (rows need not be sorted)."""
    rows = [[] for _ in range(size)]
    for s, c, t in edges:
        rows[s].append((c << 24) | t)
    rp = [0]
    for r in rows:
        rp.append(rp[-1] + len(r))
    w = rp + [e for r in rows for e in r]
    while len(w) % 4:
        w.append(0)  # 0-3 zero pad words fill the last 128-bit line
    return np.array(w, dtype=np.uint32)


def kat_ab():
    """unanchored "ab" in the reference's encoding.  size 4."""
    e = [(0, c, 1) for c in range(256)] + [(0, ord("a"), 2)]
    e += [(1, c, 1) for c in range(256)] + [(1, ord("a"), 2)]
    e += [(2, ord("b"), 3)]
    return build_words(4, e), 4


def random_nfa(rng, size, max_deg=6, n_sinks=None, alphabet=8, dense_rows=1):
    """Random automaton in the reference's conventions: state 0 fans out, accepts are empty rows, rows
    unsorted, several targets per symbol allowed (true NFA), no duplicate edges."""
    n_sinks = max(1, size // 5) if n_sinks is None else n_sinks
    sinks = set(rng.choice(np.arange(1, size), size=min(n_sinks, size - 1), replace=False).tolist()) if size > 1 else set()
    edges = set()
    for s in range(size):
        if s in sinks:
            continue
        deg = int(rng.integers(1, max_deg + 1))
        if s < dense_rows:
            deg = min(size * alphabet, 40 + int(rng.integers(0, 30)))
        for _ in range(deg):
            edges.add((s, int(rng.integers(0, alphabet)), int(rng.integers(0, size))))
    edges = list(edges)
    rng.shuffle(edges)
    return build_words(size, edges), size


def blowup_nfa(width):
    """State 0 --any byte--> states 1..width, each of which loops to all of 1..width on byte 0x41 and to
    an accept state on 0x42: drives |S_k| past any fixed list capacity (dense-form path)."""
    size = width + 2
    acc = width + 1
    e = []
    for c in (0x41, 0x42, 0x43):
        e += [(0, c, t) for t in range(1, width + 1)]
    for s in range(1, width + 1):
        e += [(s, 0x41, ((s + j) % width) + 1) for j in range(3)]
        e.append((s, 0x42, acc))
        e.append((s, 0x43, s))
    return build_words(size, e), size


def late_blowup_nfa(width):
    """Unanchored "ab" (accept state 3) next to a trap: byte 'Z' seen by the `.*` state 1 activates `width`
    states that keep each other alive on 'Y' and all fall into accept state 3 on 'B'.  A stream can therefore
    produce accept pulses, THEN outgrow any fixed list (hand-off), THEN produce more pulses."""
    size = 4 + width
    e = [(0, c, 1) for c in range(256)] + [(1, c, 1) for c in range(256)]
    e += [(0, ord("a"), 2), (1, ord("a"), 2), (2, ord("b"), 3)]
    wide = list(range(4, 4 + width))
    e += [(0, ord("Z"), t) for t in wide] + [(1, ord("Z"), t) for t in wide]
    for i, s in enumerate(wide):
        e += [(s, ord("Y"), s), (s, ord("Y"), wide[(i + 1) % width]), (s, ord("B"), 3)]
    return build_words(size, e), size


def convention_nfa(rng, size, alphabet=8, n_first=3, max_deg=4):
    """Random automaton in the shipped tables' convention: state 0 -> `.*` state 1 on all 256
    bytes, state 1 loops on all 256 bytes, both start patterns on `n_first` byte values (sometimes several targets
    on one byte), nothing leads back to 0 or 1, accepts are empty rows.  Qualifies for always-on-state folding."""
    assert size >= 4
    e = [(0, c, 1) for c in range(256)] + [(1, c, 1) for c in range(256)]
    sinks = set(rng.choice(np.arange(2, size), size=max(1, (size - 2) // 4), replace=False).tolist())
    firsts = set()
    for _ in range(n_first):
        c = int(rng.integers(0, alphabet))
        for _ in range(int(rng.integers(1, 4))):
            firsts.add((c, int(rng.integers(2, size))))
    for c, t in firsts:
        e += [(0, c, t), (1, c, t)]
    inner = set()
    for s in range(2, size):
        if s in sinks:
            continue
        for _ in range(int(rng.integers(1, max_deg + 1))):
            inner.add((s, int(rng.integers(0, alphabet)), int(rng.integers(2, size))))
    inner = list(inner)
    rng.shuffle(inner)
    return build_words(size, e + inner), size
