"""ctypes binding of the CPU oracle (oracle/liborx.so).  TEST INFRASTRUCTURE — see rx_oracle.h.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parity is unpinned by the reference (it ships no expected outputs); see DESIGN.md "Oracle".
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MODE_FULL, MODE_TB_COMPAT = 0, 1


class Event(C.Structure):
    _fields_ = [("stream", C.c_uint32), ("k", C.c_uint32), ("state", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("n_passes", C.c_uint64), ("n_events", C.c_uint64), ("sum_active", C.c_uint64),
                ("sum_edges", C.c_uint64), ("alg_bytes", C.c_uint64), ("max_active", C.c_uint64)]


class TbResult(C.Structure):
    _fields_ = [("total_cycles", C.c_uint64), ("passes", C.c_uint64), ("n_events", C.c_uint64 * 2),
                ("bram_reads", C.c_uint64), ("hung", C.c_uint64)]


def build(force=False):
    """Compile liborx.so with gcc (the oracle's own recipe, oracle/Makefile)."""
    so = os.path.join(_HERE, "liborx.so")
    srcs = [os.path.join(_HERE, f) for f in ("rx_oracle.c", "rx_cycle.c", "rx_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liborx.so"], stdout=subprocess.DEVNULL)
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        L = _lib
        u32p, u64p, u8p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)
        L.orx_load_coe.argtypes = [C.c_char_p, C.POINTER(u32p), C.POINTER(C.c_size_t)]
        L.orx_infer_size.argtypes = [u32p, C.c_size_t, u32p]
        L.orx_load_mem.argtypes = [C.c_char_p, C.POINTER(u8p), C.POINTER(C.c_size_t)]
        L.orx_free.argtypes = [C.c_void_p]
        L.orx_free.restype = None
        L.orx_match_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                      C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, u64p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                      C.POINTER(Stats), C.POINTER(C.c_int)]
        L.orx_tb_cycle.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t,
                                   C.c_uint64, C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(TbResult)]
        L.orx_cycle_probe_row.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_int, C.c_uint8,
                                          C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), u64p, C.c_void_p,
                                          C.POINTER(C.c_int)]
        L.orx_predict_cycles.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, u64p]
        L.orx_passes.argtypes = [C.c_size_t, C.c_int]
        L.orx_passes.restype = C.c_uint64
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def load_coe(path):
    """-> np.uint32 word array exactly as it will sit in HBM."""
    p, n = C.POINTER(C.c_uint32)(), C.c_size_t()
    rc = lib().orx_load_coe(path.encode(), C.byref(p), C.byref(n))
    if rc:
        raise ValueError(f"orx_load_coe({path}) -> {rc}")
    out = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
    lib().orx_free(p)
    return out


def infer_size(words):
    s = C.c_uint32()
    rc = lib().orx_infer_size(words.ctypes.data_as(C.POINTER(C.c_uint32)), words.size, C.byref(s))
    if rc:
        raise ValueError(f"orx_infer_size -> {rc}")
    return s.value


def load_mem(path):
    p, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    rc = lib().orx_load_mem(path.encode(), C.byref(p), C.byref(n))
    if rc:
        raise ValueError(f"orx_load_mem({path}) -> {rc}")
    out = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
    lib().orx_free(p)
    return out


def n_passes(n, mode):
    return int(lib().orx_passes(n, mode))


EVENT_DT = np.dtype([("stream", "<u4"), ("k", "<u4"), ("state", "<u4")])


def match_batch(words, size, data, mode=MODE_FULL, nthreads=0, init_active=None, events_cap=1 << 20,
                want_match_count=False, want_total=True, want_anymatch=True, want_final=True):
    """data: uint8 [n_streams, stream_len] (C-contiguous rows; stride = data.strides[0]).
    Returns dict(events, n_events, match_count, match_count_total, anymatch, final_active, stats, threads)."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    if data.ndim == 1:
        data = data[None, :]
    ns, sl = data.shape
    npass = n_passes(sl, mode)
    nw = (size + 63) // 64
    ev = np.zeros(events_cap, dtype=EVENT_DT)
    nev = C.c_uint64(0)
    mc = np.zeros((ns, size), np.uint32) if want_match_count else None
    tot = np.zeros(size, np.uint64) if want_total else None
    am_stride = (npass + 31) // 32
    am = np.zeros((ns, am_stride), np.uint32) if want_anymatch else None
    fin = np.zeros((ns, nw), np.uint64) if want_final else None
    st, thr = Stats(), C.c_int(0)
    if init_active is not None:
        init_active = np.ascontiguousarray(init_active, dtype=np.uint64)
    rc = lib().orx_match_batch(_ptr(words), size, _ptr(data), ns, sl, sl, mode, nthreads,
                               _ptr(init_active), _ptr(ev), events_cap, C.byref(nev), _ptr(mc), _ptr(tot),
                               _ptr(am), am_stride, _ptr(fin), C.byref(st), C.byref(thr))
    if rc:
        raise RuntimeError(f"orx_match_batch -> {rc}")
    n = min(nev.value, events_cap)
    return dict(events=ev[:n], n_events=nev.value, match_count=mc, match_count_total=tot, anymatch=am,
                final_active=fin, threads=thr.value,
                stats=dict(n_passes=st.n_passes, n_events=st.n_events, sum_active=st.sum_active,
                           sum_edges=st.sum_edges, alg_bytes=st.alg_bytes, max_active=st.max_active))


def tb_cycle(words, size, lo, hi, m_stop, bram_latency=1, skip_idle=True, max_cycles=0, events_cap=1 << 16):
    """Clock-accurate Blk_Mem_tb run.  Returns dict(total_cycles, passes, match_count, match_count_2, events)."""
    lo = np.ascontiguousarray(lo, np.uint8)
    hi = np.ascontiguousarray(hi, np.uint8)
    mc1, mc2 = np.zeros(size, np.uint32), np.zeros(size, np.uint32)
    ev = np.zeros(events_cap, dtype=EVENT_DT)
    cyc = np.zeros(events_cap, np.uint64)
    res = TbResult()
    rc = lib().orx_tb_cycle(_ptr(words), words.size, size, _ptr(lo), _ptr(hi), min(lo.size, hi.size), m_stop,
                            bram_latency, int(skip_idle), max_cycles, _ptr(mc1), _ptr(mc2), _ptr(ev), events_cap,
                            _ptr(cyc), C.byref(res))
    if rc:
        raise RuntimeError(f"orx_tb_cycle -> {rc}")
    n = min(res.n_events[0] + res.n_events[1], events_cap)
    return dict(total_cycles=res.total_cycles, passes=res.passes, hung=bool(res.hung), match_count=mc1,
                match_count_2=mc2, events=ev[:n], event_cycles=cyc[:n], n_events=(res.n_events[0], res.n_events[1]))


def probe_row(words, size, state, c, bram_latency=1):
    cap = 4096
    addrs = np.zeros(cap, np.uint32)
    n, clk, acc = C.c_size_t(), C.c_uint64(), C.c_int()
    nxt = np.zeros((size + 63) // 64, np.uint64)
    rc = lib().orx_cycle_probe_row(_ptr(words), words.size, size, state, bram_latency, c, _ptr(addrs), cap,
                                   C.byref(n), C.byref(clk), _ptr(nxt), C.byref(acc))
    return dict(rc=rc, addrs=addrs[:min(n.value, cap)].copy(), clocks=clk.value, next=nxt, accepted=bool(acc.value))


def predict_cycles(words, size, lo, hi, passes):
    out = C.c_uint64()
    rc = lib().orx_predict_cycles(_ptr(words), size, _ptr(np.ascontiguousarray(lo, np.uint8)),
                                  _ptr(np.ascontiguousarray(hi, np.uint8)), passes, C.byref(out))
    if rc:
        raise RuntimeError(f"orx_predict_cycles -> {rc}")
    return out.value


# ---- digests (SURVEY App. B.3 / D convention: sha256 over little-endian u32) ----
def h_match_count(mc):
    return hashlib.sha256(np.asarray(mc, dtype="<u4").tobytes()).hexdigest()


def h_events(events):
    """sha256 of concatenated (k, state) u32-LE pairs in (k, state) order — one stream."""
    a = np.empty((len(events), 2), dtype="<u4")
    a[:, 0] = events["k"]
    a[:, 1] = events["state"]
    return hashlib.sha256(a.tobytes()).hexdigest()


def bits_to_states(row):
    """u64 bitmask row -> sorted list of state ids."""
    out = []
    for wi, w in enumerate(np.asarray(row, dtype=np.uint64).tolist()):
        while w:
            b = (w & -w).bit_length() - 1
            out.append(wi * 64 + b)
            w &= w - 1
    return out
