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
