"""ctypes binding of librxmatch.so — one Python call per C-ABI entry point (include/rxmatch.h).

Mirrors the reference's only interface, the ports + protocol of `CSR_traversal` as driven by
`Blk_Mem_tb` (Design/FPGA.v:23-43, Simulation/testbench_BLK_Mem.sv:49-87): load a .coe, load
.mem traces, feed bytes, get accept pulses back.  No computation happens here; if the HIP library is
missing this module raises instead of falling back to anything.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

MODE_FULL, MODE_TB_COMPAT = 0, 1
KERNEL_AUTO, KERNEL_CSR_WAVE, KERNEL_SYM_WAVE, KERNEL_SYM_GROUP, KERNEL_SYM_PACK, KERNEL_DFA, KERNEL_SYM_REG = 0, 1, 2, 3, 4, 5, 6
KERNEL_NAMES = {0: "auto", 1: "csr_wave", 2: "sym_wave", 3: "sym_group", 4: "sym_pack", 5: "dfa", 6: "sym_reg"}

# rx_opts.flags (A/B and diagnostic switches; read at plan creation, never from the environment)
OPT_NO_PRUNE, OPT_FORCE_PRUNE, OPT_VERBOSE, OPT_PROFILE_PACK, OPT_NO_FOLD, OPT_FORCE_FOLD, OPT_REG_NO_SKIP = 1, 2, 4, 8, 16, 32, 64
OPT_INJECT_RUN_FAULT, OPT_NO_PROBE = 128, 256

EVENT_DT = np.dtype([("stream", "<u4"), ("k", "<u4"), ("state", "<u4")])


class RxError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        L = lib()
        msg = L.rx_strerror(code).decode()
        hip = L.rx_last_hip_error().decode()
        super().__init__(f"{what}: {msg} [{code}]" + (f" — {hip}" if hip and code in (-6, -7, -5) else ""))


class _Opts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("mode", C.c_uint32), ("kernel", C.c_uint32),
                ("stream", C.c_void_p), ("k_base", C.c_uint64), ("collect_stats", C.c_uint32),
                ("group_lanes", C.c_uint32), ("flags", C.c_uint32)]


class _Stats(C.Structure):
    _fields_ = [("n_passes", C.c_uint64), ("n_events", C.c_uint64), ("sum_active", C.c_uint64),
                ("sum_edges", C.c_uint64), ("alg_bytes", C.c_uint64), ("kernel_ms", C.c_double),
                ("h2d_ms", C.c_double), ("d2h_ms", C.c_double), ("kernel_used", C.c_uint32),
                ("n_launches", C.c_uint32), ("tb_cycles", C.c_uint64), ("lanes_used", C.c_uint32),
                ("variant", C.c_uint32)]


class _Result(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("events_overflow", C.c_uint32), ("events", C.c_void_p),
                ("events_cap", C.c_size_t), ("n_events", C.c_size_t), ("match_count", C.c_void_p),
                ("match_count_total", C.c_void_p), ("anymatch", C.c_void_p), ("anymatch_stride", C.c_size_t),
                ("final_active", C.c_void_p), ("stats", _Stats),
                ("final_states", C.c_void_p), ("final_off", C.c_void_p), ("final_cnt", C.c_void_p),
                ("final_states_cap", C.c_size_t), ("n_final_states", C.c_size_t), ("final_states_overflow", C.c_uint32),
                ("reserved0", C.c_uint32)]


class _Info(C.Structure):
    _fields_ = [("size", C.c_uint32), ("nnz", C.c_uint32), ("n_accept", C.c_uint32), ("n_words", C.c_uint32),
                ("max_degree", C.c_uint32), ("n_bitmask_words64", C.c_uint32)]


# every symbol include/rxmatch.h declares
RE_ICASE, RE_DOTALL = 1, 2

ABI_SYMBOLS = ["rx_nfa_dfa_info", "rx_nfa_dfa_reset", "rx_compile_patterns", "rx_nfa_accept_pattern", "rx_nfa_save_coe", "rx_strerror", "rx_last_hip_error", "rx_abi_version", "rx_nfa_load_coe", "rx_nfa_from_words",
               "rx_nfa_get_info", "rx_nfa_words", "rx_nfa_free", "rx_trace_load_mem", "rx_free", "rx_match",
               "rx_match_sharded", "rx_plan_create", "rx_plan_upload", "rx_plan_set_device_input",
               "rx_plan_set_init_active", "rx_plan_launch", "rx_plan_sync", "rx_plan_kernel_times", "rx_plan_download", "rx_plan_free",
               "rx_device_count", "rx_device_name", "rx_plan_run", "rx_host_register", "rx_host_unregister", "rx_plan_tune",
               "rx_plan_busy"]

_lib = None


def lib_path():
    """In-tree library; RX_LIBRARY_PATH overrides it (A/B runs of two builds in one session)."""
    return os.environ.get("RX_LIBRARY_PATH") or os.path.join(_HERE, "librxmatch.so")


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm bundles its own libamdhip64.so (same soname as /opt/rocm's).  Two HIP runtimes in one
    process cannot share a GPU, so if torch is installed, map ITS runtime first: librxmatch.so and a later
    (or earlier) `import torch` then resolve to the same library whatever the import order.  torch itself is
    not imported here."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(p):
        try:
            C.CDLL(p, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load librxmatch.so (built in-tree by __graft_entry__.build() / csrc/Makefile).  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise ImportError(f"{p} is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    _share_hip_runtime_with_torch()
    L = C.CDLL(p)
    vp, sz, u32, i32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int
    L.rx_strerror.restype = C.c_char_p
    L.rx_strerror.argtypes = [i32]
    L.rx_last_hip_error.restype = C.c_char_p
    L.rx_abi_version.restype = i32
    L.rx_nfa_load_coe.argtypes = [C.c_char_p, u32, C.POINTER(vp)]
    L.rx_nfa_from_words.argtypes = [vp, sz, u32, C.POINTER(vp)]
    L.rx_compile_patterns.argtypes = [C.POINTER(C.c_char_p), sz, u32, C.POINTER(vp), C.c_char_p, sz]
    L.rx_nfa_accept_pattern.argtypes = [vp, u32, C.POINTER(C.c_int32)]
    L.rx_nfa_save_coe.argtypes = [vp, C.c_char_p]
    L.rx_nfa_dfa_info.argtypes = [vp, i32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.rx_nfa_dfa_reset.argtypes = [vp, i32]
    L.rx_nfa_get_info.argtypes = [vp, C.POINTER(_Info)]
    L.rx_nfa_words.restype = C.POINTER(C.c_uint32)
    L.rx_nfa_words.argtypes = [vp, C.POINTER(sz)]
    L.rx_nfa_free.argtypes = [vp]
    L.rx_nfa_free.restype = None
    L.rx_trace_load_mem.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(sz)]
    L.rx_free.argtypes = [vp]
    L.rx_free.restype = None
    L.rx_match.argtypes = [vp, vp, sz, sz, sz, vp, C.POINTER(_Opts), C.POINTER(_Result)]
    L.rx_match_sharded.argtypes = [vp, vp, sz, sz, sz, C.POINTER(C.c_int), i32, C.POINTER(_Opts), C.POINTER(_Result)]
    L.rx_plan_create.argtypes = [vp, C.POINTER(_Opts), sz, sz, sz, u32, u32, u32, C.POINTER(vp)]
    L.rx_plan_upload.argtypes = [vp, vp, sz, sz, sz]
    L.rx_plan_set_device_input.argtypes = [vp, vp, sz, sz, sz]
    L.rx_plan_set_init_active.argtypes = [vp, vp]
    L.rx_plan_launch.argtypes = [vp]
    if hasattr(L, "rx_plan_tune"):  # (an older build loaded through RX_LIBRARY_PATH for an A/B run has neither)
        L.rx_plan_tune.argtypes = [vp]
        L.rx_plan_busy.argtypes = [vp, C.POINTER(u32)]
    L.rx_plan_sync.argtypes = [vp, C.POINTER(C.c_double)]
    L.rx_plan_kernel_times.argtypes = [vp, C.POINTER(u32), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double)]
    L.rx_plan_download.argtypes = [vp, C.POINTER(_Result)]
    L.rx_plan_run.argtypes = [vp, vp, sz, sz, sz, C.POINTER(_Result)]
    L.rx_host_register.argtypes = [vp, sz]
    L.rx_host_unregister.argtypes = [vp]
    L.rx_plan_free.argtypes = [vp]
    L.rx_plan_free.restype = None
    L.rx_device_count.argtypes = [C.POINTER(i32)]
    L.rx_device_name.argtypes = [i32, C.c_char_p, sz]
    _lib = L
    return L


def _chk(rc, what):
    if rc != 0:
        raise RxError(rc, what)


def device_count():
    n = C.c_int(0)
    _chk(lib().rx_device_count(C.byref(n)), "rx_device_count")
    return n.value


def device_name(dev=0):
    buf = C.create_string_buffer(256)
    _chk(lib().rx_device_name(dev, buf, 256), "rx_device_name")
    return buf.value.decode()


def load_mem(path):
    """$readmemh trace -> np.uint8 array (testbench_BLK_Mem.sv:34-35)."""
    p, n = C.POINTER(C.c_uint8)(), C.c_size_t()
    _chk(lib().rx_trace_load_mem(os.fsencode(path), C.byref(p), C.byref(n)), f"rx_trace_load_mem({path})")
    out = np.ctypeslib.as_array(p, shape=(max(n.value, 1),))[:n.value].copy()
    lib().rx_free(p)
    return out


class Nfa:
    """Immutable CSR automaton handle (rx_nfa).  `words` is the .coe content unchanged."""

    def __init__(self, handle):
        self._h = handle
        info = _Info()
        _chk(lib().rx_nfa_get_info(self._h, C.byref(info)), "rx_nfa_get_info")
        self.size, self.nnz, self.n_accept = info.size, info.nnz, info.n_accept
        self.n_words, self.max_degree, self.nw64 = info.n_words, info.max_degree, info.n_bitmask_words64

    @classmethod
    def load_coe(cls, path, size=0):
        h = C.c_void_p()
        _chk(lib().rx_nfa_load_coe(os.fsencode(path), size, C.byref(h)), f"rx_nfa_load_coe({path})")
        return cls(h)

    @classmethod
    def from_words(cls, words, size=0):
        w = np.ascontiguousarray(words, dtype=np.uint32)
        h = C.c_void_p()
        _chk(lib().rx_nfa_from_words(w.ctypes.data_as(C.c_void_p), w.size, size, C.byref(h)), "rx_nfa_from_words")
        return cls(h)

    @classmethod
    def compile(cls, patterns, icase=False, dotall=False):
        """rx_compile_patterns(): list of regex strings (bytes or str, optionally /re/flags) -> one automaton."""
        pats = [p if isinstance(p, bytes) else p.encode("latin-1") for p in patterns]
        arr = (C.c_char_p * len(pats))(*pats)
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = lib().rx_compile_patterns(arr, len(pats), (RE_ICASE if icase else 0) | (RE_DOTALL if dotall else 0),
                                       C.byref(h), err, 512)
        if rc:
            raise RxError(rc, f"rx_compile_patterns: {err.value.decode(errors='replace')}")
        return cls(h)

    def accept_pattern(self, state):
        out = C.c_int32(-1)
        _chk(lib().rx_nfa_accept_pattern(self._h, int(state), C.byref(out)), "rx_nfa_accept_pattern")
        return out.value

    def dfa_info(self, device=0):
        """(states, transitions) of the lazy-DFA cache built so far on `device`."""
        a, b = C.c_uint64(), C.c_uint64()
        _chk(lib().rx_nfa_dfa_info(self._h, device, C.byref(a), C.byref(b)), "rx_nfa_dfa_info")
        return a.value, b.value

    def dfa_reset(self, device=0):
        _chk(lib().rx_nfa_dfa_reset(self._h, device), "rx_nfa_dfa_reset")

    def save_coe(self, path):
        _chk(lib().rx_nfa_save_coe(self._h, os.fsencode(path)), f"rx_nfa_save_coe({path})")

    @property
    def words(self):
        n = C.c_size_t()
        p = lib().rx_nfa_words(self._h, C.byref(n))
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def close(self):
        if self._h:
            lib().rx_nfa_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def n_passes(stream_len, mode):
    if mode == MODE_TB_COMPAT:
        return max(stream_len - 1, 0)
    return stream_len + 1


def _mk_opts(device, mode, kernel, stream, k_base, collect_stats, group_lanes=0, flags=0):
    o = _Opts()
    o.group_lanes = group_lanes
    o.flags = flags
    o.struct_size = C.sizeof(_Opts)
    o.device, o.mode, o.kernel = device, mode, kernel
    o.stream = stream
    o.k_base = k_base
    o.collect_stats = int(collect_stats)  # 0 / 1 (True) / 2 = + testbench pair clock model
    return o


class _Out:
    """Caller-allocated output arrays for one rx_result."""

    def __init__(self, nfa, n_streams, stream_len, mode, events_cap, want_match_count, want_total, want_anymatch,
                 want_final, compact_final=0):
        self.npass = n_passes(stream_len, mode)
        self.ev = np.zeros(max(events_cap, 1), dtype=EVENT_DT) if events_cap else None
        self.mc = np.zeros((n_streams, nfa.size), np.uint32) if want_match_count else None
        self.tot = np.zeros(nfa.size, np.uint64) if want_total else None
        # (rows padded to a multiple of eight words: the plan's own pitch, so that the copies are flat; readers slice [:, :ceil(npass/32)])
        self.am_stride = (max((self.npass + 31) // 32, 1) + 7) & ~7
        self.am = np.zeros((n_streams, self.am_stride), np.uint32) if want_anymatch else None
        # compact_final = capacity (entries) of the list form of the final sets (rx_plan_run); the bitmask rows are then left out
        self.fin = np.zeros((n_streams, nfa.nw64), np.uint64) if (want_final and not compact_final) else None
        self.fst = np.zeros(compact_final, np.uint32) if compact_final else None
        self.foff = np.zeros(n_streams, np.uint32) if compact_final else None
        self.fcnt = np.zeros(n_streams, np.uint32) if compact_final else None
        r = _Result()
        r.struct_size = C.sizeof(_Result)
        if self.ev is not None:
            r.events, r.events_cap = self.ev.ctypes.data, events_cap
        if self.mc is not None:
            r.match_count = self.mc.ctypes.data
        if self.tot is not None:
            r.match_count_total = self.tot.ctypes.data
        if self.am is not None:
            r.anymatch, r.anymatch_stride = self.am.ctypes.data, self.am_stride
        if self.fin is not None:
            r.final_active = self.fin.ctypes.data
        if self.fst is not None:
            r.final_states, r.final_off, r.final_cnt = self.fst.ctypes.data, self.foff.ctypes.data, self.fcnt.ctypes.data
            r.final_states_cap = compact_final
        self.r = r

    def as_dict(self):
        r, s = self.r, self.r.stats
        return dict(events=self.ev[:r.n_events] if self.ev is not None else None, n_events=int(s.n_events),
                    events_overflow=bool(r.events_overflow), match_count=self.mc, match_count_total=self.tot,
                    anymatch=self.am[:, :max((self.npass + 31) // 32, 1)] if self.am is not None else None, final_active=self.fin,
                    final_states=self.fst[:r.n_final_states] if self.fst is not None else None, final_off=self.foff,
                    final_cnt=self.fcnt, final_states_overflow=bool(r.final_states_overflow),
                    stats=dict(n_passes=int(s.n_passes), n_events=int(s.n_events), sum_active=int(s.sum_active),
                               sum_edges=int(s.sum_edges), alg_bytes=int(s.alg_bytes), kernel_ms=s.kernel_ms,
                               h2d_ms=s.h2d_ms, d2h_ms=s.d2h_ms, kernel_used=int(s.kernel_used),
                               n_launches=int(s.n_launches), tb_cycles=int(s.tb_cycles), lanes_used=int(s.lanes_used),
                               variant=_variant_name(s)))


def expand_final(res, nw64):
    """Bitmask rows [n_streams][nw64] (uint64) from the list form of the final sets (Plan.run(compact_final=N))."""
    off, cnt, st = res["final_off"], res["final_cnt"], res["final_states"]
    rows = np.zeros((len(off), nw64), np.uint64)
    for s in range(len(off)):
        for t in st[off[s]:off[s] + cnt[s]]:
            rows[s, int(t) >> 6] |= np.uint64(1) << np.uint64(int(t) & 63)
    return rows


def _variant_name(s):
    """Which build of the kernel ran, e.g. 'S13', 'S32+fold', 'G4', 'S8+prune' ('' for the wave kernels)."""
    if not s.lanes_used and not s.variant:
        return ""
    v = (("G" if s.kernel_used == KERNEL_SYM_GROUP else "S") + str(s.lanes_used)) if s.lanes_used else "W1"
    for bit, name in ((1, "stats"), (2, "prune"), (4, "fold")):
        if s.variant & bit:
            v += "+" + name
    return v


def _as_rows(data):
    data = np.asarray(data, dtype=np.uint8)
    if data.ndim == 1:
        data = data[None, :]
    if data.ndim != 2:
        raise ValueError("data must be [n_streams, stream_len] uint8")
    if data.shape[1] and data.strides[1] != 1:
        data = np.ascontiguousarray(data)
    stride = data.strides[0] if data.shape[0] > 1 else max(data.shape[1], 1)
    if stride < data.shape[1]:
        data = np.ascontiguousarray(data)
        stride = data.shape[1]
    return data, stride


def match(nfa, data, mode=MODE_FULL, kernel=KERNEL_AUTO, device=-1, init_active=None, events_cap=1 << 20,
          want_match_count=False, want_total=True, want_anymatch=True, want_final=True, collect_stats=False,
          k_base=0, group_lanes=0, flags=0, compact_final=0):
    """rx_match(): one-shot match of uint8 [n_streams, stream_len] host rows on one GPU.  compact_final = N: the final
    sets as lists of at most N states in all (see Plan.run) instead of bitmask rows; streams from reset only."""
    data, stride = _as_rows(data)
    ns, sl = data.shape
    out = _Out(nfa, ns, sl, mode, events_cap, want_match_count, want_total, want_anymatch, want_final, compact_final)
    o = _mk_opts(device, mode, kernel, None, k_base, collect_stats, group_lanes, flags)
    ia = None
    if init_active is not None:
        ia = np.ascontiguousarray(init_active, dtype=np.uint64)
        if ia.shape != (ns, nfa.nw64):
            raise ValueError("init_active must be [n_streams, ceil(size/64)] uint64")
    _chk(lib().rx_match(nfa._h, data.ctypes.data, ns, sl, stride, ia.ctypes.data if ia is not None else None,
                        C.byref(o), C.byref(out.r)), "rx_match")
    return out.as_dict()


def match_sharded(nfa, data, devices, mode=MODE_FULL, kernel=KERNEL_AUTO, events_cap=1 << 20,
                  want_match_count=False, want_total=True, want_anymatch=True, want_final=True, collect_stats=False,
                  group_lanes=0, flags=0):
    """rx_match_sharded(): contiguous stream blocks over several GPUs of this process, no collective."""
    data, stride = _as_rows(data)
    ns, sl = data.shape
    out = _Out(nfa, ns, sl, mode, events_cap, want_match_count, want_total, want_anymatch, want_final)
    o = _mk_opts(-1, mode, kernel, None, 0, collect_stats, group_lanes, flags)
    devs = (C.c_int * len(devices))(*devices)
    _chk(lib().rx_match_sharded(nfa._h, data.ctypes.data, ns, sl, stride, devs, len(devices), C.byref(o),
                                C.byref(out.r)), "rx_match_sharded")
    return out.as_dict()


class Plan:
    """rx_plan: inputs stay resident in HBM across launches (serving / benchmarking)."""

    def __init__(self, nfa, max_streams, max_stream_len, mode=MODE_FULL, kernel=KERNEL_AUTO, device=-1, stream=None,
                 events_cap=1 << 20, want_match_count=False, want_anymatch=True, want_final=True, collect_stats=False,
                 k_base=0, group_lanes=0, flags=0):
        self.nfa, self.mode = nfa, mode
        self.events_cap = events_cap
        self.want = (want_match_count, want_anymatch, want_final)
        self._o = _mk_opts(device, mode, kernel, stream, k_base, collect_stats, group_lanes, flags)
        self._h = C.c_void_p()
        _chk(lib().rx_plan_create(nfa._h, C.byref(self._o), max_streams, max_stream_len, events_cap,
                                  int(want_match_count), int(want_anymatch), int(want_final), C.byref(self._h)),
             "rx_plan_create")
        self.n_streams = self.stream_len = 0
        self._keep = None

    def upload(self, data):
        data, stride = _as_rows(data)
        self.n_streams, self.stream_len = data.shape
        _chk(lib().rx_plan_upload(self._h, data.ctypes.data, self.n_streams, self.stream_len, stride),
             "rx_plan_upload")

    def set_device_input(self, dptr, n_streams, stream_len, stride, keepalive=None):
        """dptr: device address (e.g. torch_tensor.data_ptr()); keepalive pins the owner object."""
        self.n_streams, self.stream_len = n_streams, stream_len
        self._keep = keepalive
        _chk(lib().rx_plan_set_device_input(self._h, dptr, n_streams, stream_len, stride),
             "rx_plan_set_device_input")

    def set_init_active(self, init_active):
        if init_active is None:
            _chk(lib().rx_plan_set_init_active(self._h, None), "rx_plan_set_init_active")
            return
        ia = np.ascontiguousarray(init_active, dtype=np.uint64)
        _chk(lib().rx_plan_set_init_active(self._h, ia.ctypes.data), "rx_plan_set_init_active")

    def launch(self):
        _chk(lib().rx_plan_launch(self._h), "rx_plan_launch")

    def tune(self):
        """rx_plan_tune(): AUTO's probes now, decision pinned to the batch's shape (later launches only enqueue)."""
        _chk(lib().rx_plan_tune(self._h), "rx_plan_tune")

    def busy(self):
        """rx_plan_busy(): bit mask of the plan's streams that still have work queued (non-blocking)."""
        b = C.c_uint32()
        _chk(lib().rx_plan_busy(self._h, C.byref(b)), "rx_plan_busy")
        return b.value

    def sync(self):
        ms = C.c_double()
        _chk(lib().rx_plan_sync(self._h, C.byref(ms)), "rx_plan_sync")
        return ms.value

    def kernel_times(self):
        """-> (n_launches, sum_ms, min_ms, max_ms) of the launches since the previous call."""
        n, s, mn, mx = C.c_uint32(), C.c_double(), C.c_double(), C.c_double()
        _chk(lib().rx_plan_kernel_times(self._h, C.byref(n), C.byref(s), C.byref(mn), C.byref(mx)),
             "rx_plan_kernel_times")
        return n.value, s.value, mn.value, mx.value

    def download(self, want_total=True):
        wmc, wam, wfin = self.want
        out = _Out(self.nfa, self.n_streams, self.stream_len, self.mode, self.events_cap, wmc, want_total, wam, wfin)
        _chk(lib().rx_plan_download(self._h, C.byref(out.r)), "rx_plan_download")
        return out.as_dict()

    def run(self, data, want_total=True, register=True, compact_final=0):
        """rx_plan_run(): host rows in, host results out in one pipelined call (upload, kernel and download of blocks of
        streams overlap).  The output arrays live as long as the plan and are page-locked once (`register`), and so is
        `data` — pass the same array again and it moves by DMA.  Returns the same dict as download(); its arrays are
        overwritten by the next run().  compact_final = N: the final sets come as lists (final_states[final_off[s] ..
        + final_cnt[s]) per stream, at most N entries in all) instead of bitmask rows — a fraction of the bytes."""
        data, stride = _as_rows(data)
        ns, sl = data.shape
        key = (ns, sl, want_total, compact_final)
        if getattr(self, "_run_key", None) != key:
            self._release_run_buffers()
            wmc, wam, wfin = self.want
            self._run_out = _Out(self.nfa, ns, sl, self.mode, self.events_cap, wmc, want_total, wam, wfin, compact_final)
            self._run_key = key
            self._run_reg = []
            if register:
                o = self._run_out
                for arr in (o.ev, o.mc, o.am, o.fin, o.fst, o.foff, o.fcnt):
                    if arr is not None and arr.nbytes:
                        _chk(lib().rx_host_register(arr.ctypes.data, arr.nbytes), "rx_host_register")
                        self._run_reg.append(arr.ctypes.data)
        if register and data.nbytes and getattr(self, "_run_in", None) != (data.ctypes.data, data.nbytes):
            if getattr(self, "_run_in", None):
                lib().rx_host_unregister(self._run_in[0])
            _chk(lib().rx_host_register(data.ctypes.data, data.nbytes), "rx_host_register")
            self._run_in = (data.ctypes.data, data.nbytes)
            self._run_in_keep = data
        self.n_streams, self.stream_len = ns, sl
        out = self._run_out
        _chk(lib().rx_plan_run(self._h, data.ctypes.data, ns, sl, stride, C.byref(out.r)), "rx_plan_run")
        return out.as_dict()

    def _release_run_buffers(self):
        for addr in getattr(self, "_run_reg", []):
            lib().rx_host_unregister(addr)
        self._run_reg = []
        if getattr(self, "_run_in", None):
            lib().rx_host_unregister(self._run_in[0])
            self._run_in = None
            self._run_in_keep = None
        self._run_out = None
        self._run_key = None

    def close(self):
        if self._h:
            self._release_run_buffers()
            lib().rx_plan_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
