"""Synthetic stream batches of BASELINE.json configs 3-5, bit-reproducible.

T ("trace windows", headline): stream s = `stream_len` consecutive bytes of the reference's
    input_trace_hi_snort_16.mem if s is odd else input_trace_lo_snort_16.mem, starting at offset
    ((s >> 1) * 977) mod (200000 - stream_len).
U ("uniform"): little-endian bytes of successive splitmix64 outputs, state0 = 20261004 + (s << 32).
Every stream starts from reset (S_0 = {0}).  `first` lets a rank generate only its shard.
"""
import os

import numpy as np

DATA_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")
SNORT_COE = os.path.join(DATA_DIR, "CSR_BlockMem_snort_16.coe")
L7_COE = os.path.join(DATA_DIR, "CSR_BlockMem.coe")
TRACES = {
    ("snort_16", "lo"): os.path.join(DATA_DIR, "input_trace_lo_snort_16.mem"),
    ("snort_16", "hi"): os.path.join(DATA_DIR, "input_trace_hi_snort_16.mem"),
    ("l7", "lo"): os.path.join(DATA_DIR, "input_trace_lo_l-7_filter.mem"),
    ("l7", "hi"): os.path.join(DATA_DIR, "input_trace_hi_l-7_filter.mem"),
}
TRACE_LEN = 200000
U_SEED = 20261004


def trace_windows(lo, hi, n_streams, stream_len, first=0):
    """Workload T.  lo/hi: the two snort_16 traces as uint8 arrays (>= 200000 bytes)."""
    lo = np.asarray(lo, np.uint8)[:TRACE_LEN]
    hi = np.asarray(hi, np.uint8)[:TRACE_LEN]
    if stream_len >= TRACE_LEN:
        raise ValueError("stream_len must be < 200000")
    s = np.arange(first, first + n_streams, dtype=np.int64)
    off = ((s >> 1) * 977) % (TRACE_LEN - stream_len)
    idx = off[:, None] + np.arange(stream_len, dtype=np.int64)[None, :]
    out = np.empty((n_streams, stream_len), np.uint8)
    odd = (s & 1) == 1
    out[odd] = hi[idx[odd]]
    out[~odd] = lo[idx[~odd]]
    return out


def _splitmix64_next(state):
    state += np.uint64(0x9E3779B97F4A7C15)
    z = state.copy()
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return state, z ^ (z >> np.uint64(31))


def uniform(n_streams, stream_len, first=0):
    """Workload U."""
    with np.errstate(over="ignore"):
        s = np.arange(first, first + n_streams, dtype=np.uint64)
        state = np.uint64(U_SEED) + (s << np.uint64(32))
        nq = (stream_len + 7) // 8
        out = np.empty((n_streams, nq), dtype="<u8")
        for q in range(nq):
            state, z = _splitmix64_next(state)
            out[:, q] = z
    return out.view(np.uint8).reshape(n_streams, nq * 8)[:, :stream_len].copy()


# ---- stand-in for BASELINE configs[4] ("full Snort community ruleset compiled to one CSR NFA") ----------
# No rules and no network exist offline, so the ruleset is synthetic: seeded content-style patterns compiled by
# the product's own regex compiler (csrc/rx_compile.cpp) into one ~10k-state automaton.  Reported as a
# stand-in, never as the real ruleset.
_RULE_WORDS = [b"admin", b"passwd", b"select", b"union", b"cmd.exe", b"/etc/", b"script", b"GET ", b"POST ",
               b"User-Agent", b"Content-Length", b"%00", b"../", b"eval(", b"base64", b"login", b"root", b"shell",
               b"wget ", b"chmod "]


def synthetic_ruleset(n=700, seed=20261004):
    import re
    rng = np.random.default_rng(seed)
    alnum = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789", np.uint8)
    pats = []
    for _ in range(n):
        a = _RULE_WORDS[int(rng.integers(len(_RULE_WORDS)))]
        tail = bytes(rng.choice(alnum, size=int(rng.integers(3, 9))))
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


def _ruleset_traffic_block(args):
    pats, n_streams, stream_len, first, seed = args
    return ruleset_traffic(pats, n_streams, stream_len, first=first, seed=seed)


def ruleset_traffic(pats, n_streams, stream_len, first=0, seed=7, workers=1):
    """Printable pseudo-traffic with rule fragments mixed in (about one fragment per 40 bytes).  Every stream has its
    own seeded generator, so blocks can be produced by `workers` processes and any shard equals the same rows of the
    whole batch."""
    if workers > 1 and n_streams >= 4 * workers:
        from concurrent.futures import ProcessPoolExecutor
        per = (n_streams + workers - 1) // workers
        jobs = [(pats, min(per, n_streams - b), stream_len, first + b, seed) for b in range(0, n_streams, per)]
        with ProcessPoolExecutor(workers) as ex:
            return np.concatenate(list(ex.map(_ruleset_traffic_block, jobs)))
    out = np.empty((n_streams, stream_len), np.uint8)
    frags = [p[:10] for p in pats[:64]] + _RULE_WORDS
    for i in range(n_streams):
        rng = np.random.default_rng(seed + 1000003 * (first + i))
        row = rng.integers(32, 127, size=stream_len, dtype=np.uint8)
        for _ in range(max(stream_len // 40, 1)):
            f = np.frombuffer(frags[int(rng.integers(len(frags)))], np.uint8)
            at = int(rng.integers(0, max(stream_len - len(f), 1)))
            row[at:at + len(f)] = f[:stream_len - at]
        out[i] = row
    return out


# ---- hand-off mix (bench.py `handoff_mix_T`): what one stream costs whose active set outgrows the pack kernel's list ----
# No input makes ONE snort_16 stream hold more than a few dozen states (the shipped hi trace peaks at 37), so the
# mechanism is measured on the shipped table with a trap grafted on: the `.*` state 1 enters a gate state on byte 0x00, the
# gate enters `width` states on a second 0x00, and those loop on 0x00 and fall into a new accept state on 0x01.  The byte
# pair (0x00, 0x00) occurs nowhere in the shipped traces, so trace windows never enter the trap.
def table_with_trap(words, size, width=220):
    """-> (words', size') in the .coe layout: the automaton `words` plus gate, `width` trap states and one accept state."""
    W = np.asarray(words, np.uint32)
    rp = W[:size + 1].astype(np.int64)
    col = W[size + 1:size + 1 + rp[size]]
    src = np.repeat(np.arange(size, dtype=np.int64), np.diff(rp))
    gate, first, acc = size, size + 1, size + 1 + width
    new_size = size + width + 2
    wide = np.arange(first, first + width, dtype=np.int64)
    e_src = [src, np.array([1], np.int64), np.full(width, gate, np.int64), wide, wide]
    e_col = [col.astype(np.int64), np.array([gate], np.int64), wide, wide, (1 << 24) | np.full(width, acc, np.int64)]
    s_all, c_all = np.concatenate(e_src), np.concatenate(e_col)
    order = np.argsort(s_all, kind="stable")
    s_all, c_all = s_all[order], c_all[order]
    nrp = np.zeros(new_size + 1, np.int64)
    np.add.at(nrp, s_all + 1, 1)
    nrp = np.cumsum(nrp)
    out = np.concatenate([nrp, c_all]).astype(np.uint32)
    pad = (-len(out)) % 4
    return np.concatenate([out, np.zeros(pad, np.uint32)]), new_size


def handoff_mix(lo, hi, n_streams, stream_len, every=64, first=0):
    """Workload T with every `every`-th stream replaced by one that enters the trap of table_with_trap at once and never
    leaves it (all bytes 0x00)."""
    rows = trace_windows(lo, hi, n_streams, stream_len, first=first)
    s = np.arange(first, first + n_streams)
    rows[(s % every) == every - 1] = 0
    return rows
