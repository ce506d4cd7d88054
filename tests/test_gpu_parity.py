"""GPU parity: every HIP kernel, called through the C-ABI (librxmatch.so), must reproduce the CPU
oracle BIT-EXACTLY (integer/byte/index work: no tolerance) — events, match counters, per-pass any-match
bitmap, final active sets and the algorithmic-byte statistics.  Run with `-m gpu` on an MI355X."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from nfa_util import blowup_nfa, build_words, convention_nfa, kat_ab, late_blowup_nfa, random_nfa

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(GOLDEN, "golden.json")))
N = 200000


@pytest.fixture(scope="module")
def kernels(rx):
    """Every kernel variant the C-ABI can launch, as rx_opts keyword sets."""
    return [dict(kernel=rx.KERNEL_CSR_WAVE), dict(kernel=rx.KERNEL_SYM_WAVE),
            dict(kernel=rx.KERNEL_SYM_GROUP, group_lanes=1), dict(kernel=rx.KERNEL_SYM_GROUP, group_lanes=2),
            dict(kernel=rx.KERNEL_SYM_GROUP, group_lanes=4), dict(kernel=rx.KERNEL_SYM_GROUP, group_lanes=8),
            dict(kernel=rx.KERNEL_SYM_GROUP, group_lanes=16), dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=4),
            dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=8), dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=13),
            dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=16), dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=24),
            dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=32), dict(kernel=rx.KERNEL_DFA), dict(kernel=rx.KERNEL_AUTO),
            # FOLD builds (always-on-state folding; plain pack kernel on automata without such a state)
            dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=8, flags=rx.host.OPT_FORCE_FOLD),
            dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=16, flags=rx.host.OPT_FORCE_FOLD | rx.host.OPT_FORCE_PRUNE),
            dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=24, flags=rx.host.OPT_FORCE_FOLD),
            dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=32, flags=rx.host.OPT_FORCE_FOLD),
            dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=48, flags=rx.host.OPT_FORCE_FOLD | rx.host.OPT_FORCE_PRUNE),
            dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=64, flags=rx.host.OPT_FORCE_FOLD),
            # register-resident one-wavefront-per-stream kernel, folded (default) and unfolded, with and without stepping over idle passes
            dict(kernel=rx.KERNEL_SYM_REG), dict(kernel=rx.KERNEL_SYM_REG, flags=rx.host.OPT_NO_FOLD),
            dict(kernel=rx.KERNEL_SYM_REG, flags=rx.host.OPT_REG_NO_SKIP), dict(kernel=rx.KERNEL_SYM_REG, flags=rx.host.OPT_REG_NO_SKIP | rx.host.OPT_NO_FOLD)]


@pytest.fixture(scope="module")
def gpu_nfas(rx, automata):
    return {name: rx.Nfa.from_words(W, size) for name, (W, size) in automata.items()}


def check_equal(rx, orx, got, ref, what, stats=True):
    assert got["n_events"] == ref["n_events"], what
    assert not got["events_overflow"], what
    assert np.array_equal(got["events"], ref["events"].astype(got["events"].dtype)), what
    for k in ("match_count", "match_count_total", "anymatch", "final_active"):
        if got.get(k) is not None and ref.get(k) is not None:
            g = got[k][:, :ref[k].shape[1]] if k == "anymatch" else got[k]  # binding keeps >= 1 word per row
            assert np.array_equal(g, ref[k]), (what, k)
    if stats:
        for k in ("n_passes", "n_events", "sum_active", "sum_edges", "alg_bytes"):
            assert got["stats"][k] == ref["stats"][k], (what, k, got["stats"][k], ref["stats"][k])


def test_loaded_native_library(rx):
    """The HIP path is the one that runs: the in-tree .so is loaded and a device is visible."""
    assert os.path.exists(rx.lib_path())
    assert rx.host.device_count() >= 1
    assert "gfx950" in rx.host.device_name(0)
    maps = open("/proc/self/maps").read()
    assert "librxmatch.so" in maps


def test_kat_ab(rx, orx, kernels):
    W, size = kat_ab()
    nfa = rx.Nfa.from_words(W)
    data = np.frombuffer(b"xabab", np.uint8)
    for mode in (rx.MODE_FULL, rx.MODE_TB_COMPAT):
        ref = orx.match_batch(W, size, data, mode=mode, want_match_count=True)
        for kern in kernels:
            got = rx.match(nfa, data, mode=mode, **kern, want_match_count=True, collect_stats=True)
            check_equal(rx, orx, got, ref, ("ab", mode, kern))
    got = rx.match(nfa, data, kernel=rx.KERNEL_AUTO, want_match_count=True)
    assert [(int(e["k"]), int(e["state"])) for e in got["events"]] == [(3, 3), (5, 3)]


@pytest.mark.parametrize("key", sorted(G["tb_compat"]))
def test_shipped_traces_single_stream(rx, orx, automata, traces, gpu_nfas, kernels, key):
    """BASELINE configs[0]/[1]: each shipped trace as ONE stream, tb-compat, vs golden digests + oracle."""
    name, lh = key.split(":")
    W, size = automata[name]
    data = traces[(name, lh)][:N]
    ref = orx.match_batch(W, size, data, mode=orx.MODE_TB_COMPAT, nthreads=1, want_match_count=True)
    g = G["tb_compat"][key]
    for kern in kernels:
        got = rx.match(gpu_nfas[name], data, mode=rx.MODE_TB_COMPAT, **kern, want_match_count=True,
                       collect_stats=True)
        check_equal(rx, orx, got, ref, (key, kern))
        assert orx.h_match_count(got["match_count"][0]) == g["H_mc"]
        assert orx.h_events(got["events"]) == g["H_ev"]
        assert orx.bits_to_states(got["final_active"][0]) == g["final_active"]
        assert got["stats"]["alg_bytes"] == g["alg_bytes"]


@pytest.mark.parametrize("name", ["l7", "snort_16"])
def test_testbench_pair_and_report(rx, orx, automata, traces, gpu_nfas, name):
    """What Blk_Mem_tb does: lo+hi in lock-step, full mode too, and the $display report text."""
    W, size = automata[name]
    lo, hi = traces[(name, "lo")], traces[(name, "hi")]
    r = rx.testbench.run(gpu_nfas[name], lo, hi)
    c = orx.tb_cycle(W, size, lo[:N + 1], hi[:N + 1], N, skip_idle=True)
    assert np.array_equal(r["match_count"][0], c["match_count"])
    assert np.array_equal(r["match_count"][1], c["match_count_2"])
    # "Total no. cycles" evaluated on the GPU == the clock-accurate model == the survey's prediction
    assert r["total_cycles"] == c["total_cycles"] == G["cycles"][name]["total_cycles"]
    want = rx.testbench.format_report(c["match_count"], c["match_count_2"], c["total_cycles"], 10 * c["total_cycles"] + 22)
    assert r["report"] == want and "match_count_2[" in want
    shown = ((c["total_cycles"] + 2**31) % 2**32) - 2**31  # `int cycles` is 32-bit signed in the testbench
    assert want.splitlines()[-1] == f"Total no. cycles: {shown:11d}"
    first = want.splitlines()[0]
    top = int(np.nonzero(c["match_count"])[0].max())
    assert first == f"match_count[{top:11d}] = {int(c['match_count'][top]) & 1023:4d}"
    rows = np.stack([lo[:N], hi[:N]])
    ref = orx.match_batch(W, size, rows, mode=orx.MODE_FULL, want_match_count=True)
    got = rx.match(gpu_nfas[name], rows, mode=rx.MODE_FULL, want_match_count=True, collect_stats=True)
    check_equal(rx, orx, got, ref, (name, "pair full"))


@pytest.mark.parametrize("workload", ["T", "U"])
def test_synthetic_batches(rx, orx, automata, traces, gpu_nfas, kernels, workload):
    """BASELINE configs[2] shape at oracle-sized scale: 1536 streams x 1024 B, both distributions."""
    W, size = automata["snort_16"]
    wl = rx.workloads
    if workload == "T":
        rows = wl.trace_windows(traces[("snort_16", "lo")], traces[("snort_16", "hi")], 1536, 1024, first=1000)
    else:
        rows = wl.uniform(1536, 1024, first=77)
    ref = orx.match_batch(W, size, rows, want_match_count=True)
    if workload == "T":
        assert ref["n_events"] > 100
    for kern in kernels:
        got = rx.match(gpu_nfas["snort_16"], rows, **kern, want_match_count=True, collect_stats=True)
        check_equal(rx, orx, got, ref, (workload, kern))
        got = rx.match(gpu_nfas["snort_16"], rows, **kern, want_match_count=True)  # the build without statistics
        check_equal(rx, orx, got, ref, (workload, kern, "plain"), stats=False)


def test_ragged_and_edge_shapes(rx, orx, automata, traces, gpu_nfas, kernels):
    """Empty / 1-byte / non-multiple-of-4 / chunk-boundary lengths, odd strides, unaligned rows."""
    W, size = automata["snort_16"]
    hi = traces[("snort_16", "hi")]
    for sl in (0, 1, 2, 3, 5, 255, 256, 257, 511, 513, 1023):
        for ns in (1, 3, 67):
            rows = np.stack([hi[(7 * s) % 900:(7 * s) % 900 + sl] for s in range(ns)]) if sl else np.zeros((ns, 0), np.uint8)
            for mode in (rx.MODE_FULL, rx.MODE_TB_COMPAT):
                ref = orx.match_batch(W, size, rows, mode=mode)
                for kern in kernels:
                    got = rx.match(gpu_nfas["snort_16"], rows, mode=mode, **kern, collect_stats=True)
                    check_equal(rx, orx, got, ref, (sl, ns, mode, kern))
                    got = rx.match(gpu_nfas["snort_16"], rows, mode=mode, **kern)
                    check_equal(rx, orx, got, ref, (sl, ns, mode, kern, "plain"), stats=False)
    # strided view: rows start at odd addresses (stride 301, offset 1)
    buf = np.zeros(40 * 301 + 8, np.uint8)
    buf[:] = np.resize(hi[:5000], buf.size)
    view = np.lib.stride_tricks.as_strided(buf[1:], shape=(40, 298), strides=(301, 1))
    ref = orx.match_batch(W, size, np.ascontiguousarray(view))
    for kern in kernels:
        got = rx.match(gpu_nfas["snort_16"], view, **kern, collect_stats=True)
        check_equal(rx, orx, got, ref, ("strided", kern))


def test_idle_stretches_are_stepped_over_exactly(rx, orx, automata, traces, gpu_nfas, kernels):
    """The FOLD builds of the pack kernel and the skipping build of the register kernel step over passes in which nothing is
    active.  Streams that are quiet except for short bursts planted around every boundary the skipping logic knows (the
    32-pass any-match word, the 64-byte window, the 256-byte chunk, the stream's first and last bytes) must give exactly
    the oracle's events, any-match bitmap, counters and final sets — on snort_16 (folded `.*` state), on a compiled automaton
    and on one without such a state (an empty set then stays empty to the end of the stream)."""
    hi = traces[("snort_16", "hi")]
    rng = np.random.default_rng(20261004)
    compiled = rx.Nfa.compile([b"needle", b"ab+c", b"x[0-9]*y", b"zz"])
    cases = [("snort_16", gpu_nfas["snort_16"], automata["snort_16"], [hi[o:o + 14] for o in (100, 3000, 50000, 120000, 199000)]),
             ("compiled", compiled, (compiled.words, compiled.size),
              [np.frombuffer(b, np.uint8) for b in (b"needle", b"abbbc", b"x0123y", b"zz", b"need", b"xab9y")])]
    marks = (0, 1, 29, 30, 31, 32, 33, 61, 62, 63, 64, 65, 126, 127, 128, 250, 254, 255, 256, 257, 510, 511, 512, 513, 1020, 1023, 1024, 1025)
    for name, nfa, (W, size), bursts in cases:
        for sl in (70, 300, 1500, 2111):
            for ns in (1, 3, 70):
                rows = np.full((ns, sl), 0x7E if name == "snort_16" else 0x2E, np.uint8)   # '~' / '.': starts nothing
                for s in range(ns):
                    if ns > 3 and s % 7 == 0:
                        continue                                    # quiet from the first byte to the last
                    picks = rng.choice(len(marks), size=int(rng.integers(1, 6)), replace=False)
                    for m in [marks[i] for i in picks] + ([sl - 5, sl - 1] if s % 3 == 1 else []):
                        b = bursts[int(rng.integers(len(bursts)))]
                        if 0 <= m < sl:
                            n = min(len(b), sl - m)
                            rows[s, m:m + n] = b[:n]
                for mode in (rx.MODE_FULL, rx.MODE_TB_COMPAT):
                    ref = orx.match_batch(W, size, rows, mode=mode, want_match_count=True)
                    if name == "compiled" and ns == 70 and sl == 2111:
                        assert ref["n_events"] > 20
                    for kern in kernels:
                        got = rx.match(nfa, rows, mode=mode, want_match_count=True, **kern)
                        check_equal(rx, orx, got, ref, (name, sl, ns, mode, kern), stats=False)
    # no folded state at all: after the first bytes nothing is active, and nothing ever will be
    W, size = kat_ab()
    nfa = rx.Nfa.from_words(W)
    rows = np.full((5, 1200), 0x71, np.uint8)
    rows[1, :2] = (0x61, 0x62)
    rows[3, 700:702] = (0x61, 0x62)
    ref = orx.match_batch(W, size, rows, want_match_count=True)
    for kern in kernels:
        got = rx.match(nfa, rows, want_match_count=True, **kern)
        check_equal(rx, orx, got, ref, ("ab", kern), stats=False)


def test_device_input_unaligned(rx, orx, automata, traces, gpu_nfas, kernels):
    """rx_plan_set_device_input with a caller-owned HBM buffer whose rows are NOT 4-byte aligned."""
    torch = pytest.importorskip("torch")
    W, size = automata["snort_16"]
    hi = traces[("snort_16", "hi")]
    ns, sl, stride = 50, 333, 335
    host = np.zeros(ns * stride + 3, np.uint8)
    rows = np.stack([hi[s * 11:s * 11 + sl] for s in range(ns)])
    for s in range(ns):
        host[1 + s * stride:1 + s * stride + sl] = rows[s]
    d = torch.from_numpy(host).cuda()
    ref = orx.match_batch(W, size, rows)
    for kern in kernels:
        p = rx.Plan(gpu_nfas["snort_16"], ns, sl, **kern, device=0, collect_stats=True)
        p.set_device_input(d.data_ptr() + 1, ns, sl, stride, keepalive=d)
        p.launch()
        p.launch()  # relaunching a resident plan gives the same answer
        got = p.download()
        check_equal(rx, orx, got, ref, ("device input", kern))
        n, s_ms, mn, mx = p.kernel_times()
        assert n == 2 and 0 < mn <= mx
        p.close()


def test_chunked_streaming_chain(rx, orx, automata, traces, gpu_nfas, kernels):
    """final_active -> init_active across calls + k_base == one uninterrupted run."""
    W, size = automata["snort_16"]
    hi, lo = traces[("snort_16", "hi")], traces[("snort_16", "lo")]
    rows = np.stack([hi[:6000], lo[:6000], hi[1000:7000]])
    whole = orx.match_batch(W, size, rows)
    cut = 2500
    for kern in kernels:
        a = rx.match(gpu_nfas["snort_16"], rows[:, :cut], **kern)
        b = rx.match(gpu_nfas["snort_16"], rows[:, cut:], **kern, init_active=a["final_active"], k_base=cut)
        ev_a = a["events"][a["events"]["k"] < cut]
        ev = np.concatenate([ev_a, b["events"]])
        ev = ev[np.lexsort((ev["state"], ev["k"], ev["stream"]))]
        assert np.array_equal(ev, whole["events"].astype(ev.dtype)), kern
        assert np.array_equal(b["final_active"], whole["final_active"]), kern


def test_pass_indices_up_to_the_32_bit_limit(rx, orx, automata, traces, gpu_nfas):
    """k_base at the top of the 32-bit range: event pass indices are k_base + k exactly, no wrap; one pass more is
    RX_EINVAL (checked before the launch)."""
    W, size = automata["snort_16"]
    rows = np.stack([traces[("snort_16", "hi")][:5000]] * 3)
    ref = orx.match_batch(W, size, rows)
    base = 2**32 - (5000 + 1)                      # FULL mode: passes 0..5000 -> the last index is 2^32 - 1
    for kern in (dict(kernel=rx.KERNEL_AUTO), dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=13), dict(kernel=rx.KERNEL_SYM_WAVE)):
        got = rx.match(gpu_nfas["snort_16"], rows, k_base=base, **kern)
        want = ref["events"].astype(got["events"].dtype).copy()
        assert int(want["k"].max()) + base < 2**32
        want["k"] = (want["k"].astype(np.uint64) + base).astype(np.uint32)
        assert np.array_equal(got["events"], want), kern
        with pytest.raises(rx.RxError) as e:
            rx.match(gpu_nfas["snort_16"], rows, k_base=base + 1, **kern)
        assert e.value.code == -1


@pytest.mark.parametrize("width", [300, 1300])
def test_active_set_larger_than_list_capacity(rx, orx, kernels, width):
    """|S_k| = 300 > the 256 entries a wave-kernel list holds when that kernel runs the batch (list form, 1 024 entries, in the
    launch that finishes the other kernels' hand-offs), |S_k| = 1 300 > both: the dense (bitmask-walk) form and the list form
    must give identical results."""
    W, size = blowup_nfa(width)
    nfa = rx.Nfa.from_words(W)
    rng = np.random.default_rng(5)
    rows = rng.choice(np.array([0x41, 0x42, 0x43, 0x44], np.uint8), size=(9, 64), p=[0.45, 0.05, 0.45, 0.05])
    rows[0, :6] = [0x43, 0x41, 0x41, 0x43, 0x42, 0x43]
    ref = orx.match_batch(W, size, rows, want_match_count=True)
    assert ref["stats"]["sum_active"] > width * 20 and ref["n_events"] >= 1 and ref["stats"]["max_active"] >= width
    for kern in kernels:
        got = rx.match(nfa, rows, **kern, want_match_count=True, collect_stats=True)
        check_equal(rx, orx, got, ref, ("blowup", width, kern))
        got = rx.match(nfa, rows, **kern, want_match_count=True)
        check_equal(rx, orx, got, ref, ("blowup", width, kern, "plain"), stats=False)


def test_handoff_in_the_middle_of_a_stream(rx, orx, kernels):
    """Accept pulses before AND after the pass at which a stream outgrows the group / pack kernel's list: the
    hand-off to the wave kernel must keep the earlier pulses, the partial any-match word, the pinned state and
    the statistics, for streams that blow up at different passes (or never) inside one wavefront."""
    W, size = late_blowup_nfa(220)
    nfa = rx.Nfa.from_words(W)
    base = b"xabxab..abYab"
    rows = np.zeros((40, 96), np.uint8)
    for s in range(40):
        txt = bytearray((base * 10)[:96])
        if s % 3 != 2:                       # every third stream never blows up
            at = 7 + (s * 5) % 60
            txt[at:at + 6] = b"ZYYBab"
        rows[s] = np.frombuffer(bytes(txt), np.uint8)
    for mode in (rx.MODE_FULL, rx.MODE_TB_COMPAT):
        ref = orx.match_batch(W, size, rows, mode=mode, want_match_count=True)
        assert ref["stats"]["max_active"] > 200 and ref["n_events"] > 100
        for kern in kernels:
            got = rx.match(nfa, rows, mode=mode, want_match_count=True, collect_stats=True, **kern)
            check_equal(rx, orx, got, ref, ("late handoff", mode, kern))
            got = rx.match(nfa, rows, mode=mode, want_match_count=True, **kern)
            check_equal(rx, orx, got, ref, ("late handoff", mode, kern, "plain"), stats=False)


def test_many_handoffs_take_one_wavefront_per_stream(rx, orx):
    """More handed-off streams than the workgroup-per-stream form of the finishing launch takes (4 096): that launch falls
    back to one wavefront per stream.  5 000 streams, every one of them outgrows the pack kernel's list."""
    W, size = late_blowup_nfa(220)
    nfa = rx.Nfa.from_words(W)
    base = b"xabxab..abYab"
    rows = np.zeros((5000, 48), np.uint8)
    for s in range(5000):
        txt = bytearray((base * 4)[:48])
        at = 3 + (s * 7) % 30
        txt[at:at + 6] = b"ZYYBab"
        rows[s] = np.frombuffer(bytes(txt), np.uint8)
    ref = orx.match_batch(W, size, rows, want_match_count=False)
    assert ref["stats"]["max_active"] > 200
    for kern in (dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=13), dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=16, flags=rx.host.OPT_FORCE_FOLD),
                 dict(kernel=rx.KERNEL_SYM_GROUP, group_lanes=4)):
        got = rx.match(nfa, rows, **kern)
        check_equal(rx, orx, got, ref, ("many hand-offs", kern), stats=False)


def test_random_automata(rx, orx, kernels):
    """Seeded random NFAs (unsorted rows, multi-target symbols, self loops, sinks) x random streams."""
    rng = np.random.default_rng(20261004)
    for trial in range(40):
        size = int(rng.integers(2, 400))
        alpha = int(rng.integers(2, 12))
        W, size = random_nfa(rng, size, max_deg=int(rng.integers(1, 20)), alphabet=alpha, dense_rows=int(rng.integers(0, 3)))
        nfa = rx.Nfa.from_words(W, size)
        ns, sl = int(rng.integers(1, 40)), int(rng.integers(0, 300))
        rows = rng.integers(0, alpha, size=(ns, sl), dtype=np.uint8)
        mode = int(trial & 1)
        ref = orx.match_batch(W, size, rows, mode=mode, want_match_count=True)
        for kern in kernels:
            got = rx.match(nfa, rows, mode=mode, **kern, want_match_count=True, collect_stats=True)
            check_equal(rx, orx, got, ref, ("random", trial, kern))
            got = rx.match(nfa, rows, mode=mode, **kern, want_match_count=True)
            check_equal(rx, orx, got, ref, ("random", trial, kern, "plain"), stats=False)


def test_automata_in_the_reference_convention(rx, orx, kernels):
    """Random automata built like the shipped tables (state 0 -> pinned `.*` state 1 on every byte, state 1 loops
    on every byte, both start patterns, nothing leads back): the shape the pinned-state logic of the group kernel
    and the list kernels see in practice."""
    rng = np.random.default_rng(1711)
    for trial in range(30):
        alpha = int(rng.integers(2, 10))
        Wc, sz = convention_nfa(rng, int(rng.integers(4, 200)), alphabet=alpha, n_first=int(rng.integers(1, 5)))
        nfa = rx.Nfa.from_words(Wc, sz)
        ns, sl = int(rng.integers(1, 70)), int(rng.integers(0, 260))
        rows = rng.integers(0, alpha, size=(ns, sl), dtype=np.uint8)
        mode = int(trial & 1)
        ref = orx.match_batch(Wc, sz, rows, mode=mode, want_match_count=True)
        for kern in kernels:
            got = rx.match(nfa, rows, mode=mode, **kern, want_match_count=True, collect_stats=True)
            check_equal(rx, orx, got, ref, ("convention", trial, kern))
            got = rx.match(nfa, rows, mode=mode, **kern, want_match_count=True)
            check_equal(rx, orx, got, ref, ("convention", trial, kern, "plain"), stats=False)


def test_lookahead_pruning_of_multi_target_rows(rx, orx, automata, traces, gpu_nfas):
    """Pack kernel without statistics = the build that really runs: rows with several targets on one byte insert only
    the targets that survive the stream's next byte (never at the stream's last byte).  Events, counts, bitmaps and
    final sets must not depend on it (rx_opts.flags RX_OPT_NO_PRUNE = the unpruned build), for stream lengths around the 64-byte
    window edges and in both modes."""
    cases = []
    W, size = automata["l7"]                                          # a multi-target row in nearly every pass
    for sl in (63, 64, 65, 129, 500):
        cases.append(("l7", gpu_nfas["l7"], W, size,
                      rx.workloads.trace_windows(traces[("l7", "lo")], traces[("l7", "hi")], 48, sl)))
    pats = rx.workloads.synthetic_ruleset(120)
    rs = rx.Nfa.compile(pats)
    cases.append(("ruleset", rs, rs.words, rs.size, rx.workloads.ruleset_traffic(pats, 40, 700)))
    rng = np.random.default_rng(4242)
    for trial in range(25):
        alpha = int(rng.integers(2, 6))
        Wr, sz = random_nfa(rng, int(rng.integers(3, 300)), max_deg=int(rng.integers(4, 24)), alphabet=alpha,
                            dense_rows=int(rng.integers(0, 3)))
        cases.append((("random", trial), rx.Nfa.from_words(Wr, sz), Wr, sz,
                      rng.integers(0, alpha, size=(int(rng.integers(1, 50)), int(rng.integers(0, 200))), dtype=np.uint8)))
    force = rx.host.OPT_FORCE_PRUNE  # these batches are too small for the probe that normally decides
    for n, (name, nfa, Wc, sz, rows) in enumerate(cases):
        mode = n & 1
        ref = orx.match_batch(Wc, sz, rows, mode=mode, want_match_count=True, events_cap=1 << 22)
        for lanes in (4, 8, 13, 16, 32):
            kern = dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=lanes)
            got = rx.match(nfa, rows, mode=mode, **kern, want_match_count=True, events_cap=1 << 22, flags=force)
            check_equal(rx, orx, got, ref, ("pruned", name, lanes), stats=False)
        got = rx.match(nfa, rows, mode=mode, kernel=rx.KERNEL_SYM_PACK, group_lanes=16, want_match_count=True, events_cap=1 << 22,
                       flags=rx.host.OPT_NO_PRUNE)
        check_equal(rx, orx, got, ref, ("unpruned", name), stats=False)


def test_more_than_65536_states(rx, orx, kernels):
    """State ids above 16 bits: the pack kernel's list entries then keep the full 24-bit state field and take the
    byte class from the stream window (its WIDE build); every kernel must agree with the oracle there too."""
    size = 70000
    hi = [65535, 65536, 66000, 69000, 69001, 69990]
    e = [(0, c, 1) for c in range(256)] + [(1, c, 1) for c in range(256)]
    for src in (0, 1):
        e += [(src, ord("a"), 69000), (src, ord("x"), 66000), (src, ord("q"), 65535)]
    e += [(69000, ord("a"), 69000), (69000, ord("b"), 69001)]                      # a+b -> accept 69001
    e += [(66000, ord("y"), t) for t in (65536, 69990, 300)]                     # several targets on one byte
    e += [(65536, ord("z"), 69999), (69990, ord("z"), 69999), (300, ord("z"), 2)]  # accepts 69999 and 2
    e += [(65535, ord("q"), 65535), (65535, ord("b"), 69001)]
    W = build_words(size, e)
    nfa = rx.Nfa.from_words(W, size)
    rng = np.random.default_rng(65536)
    rows = rng.choice(np.frombuffer(b"aabxyzq.", np.uint8), size=(40, 300))
    ref = orx.match_batch(W, size, rows, want_match_count=True)
    assert ref["n_events"] > 50 and {int(s) for s in ref["events"]["state"]} >= {2, 69001, 69999}
    for kern in kernels:
        got = rx.match(nfa, rows, **kern, want_match_count=True, collect_stats=True)
        check_equal(rx, orx, got, ref, ("wide", kern))


def test_pair_clock_model_on_random_automata(rx, orx):
    """rx_stats.tb_cycles (GPU) == clock-accurate restatement of FPGA.v + Blk_Mem_tb, many pairs per batch."""
    rng = np.random.default_rng(77)
    checked = 0
    for trial in range(16):
        size = int(rng.integers(2, 60))
        alpha = int(rng.integers(2, 9))
        W, size = random_nfa(rng, size, max_deg=int(rng.integers(1, 14)), alphabet=alpha, dense_rows=int(rng.integers(0, 2)))
        nfa = rx.Nfa.from_words(W, size)
        n_pairs, n = int(rng.integers(1, 20)), int(rng.integers(2, 150))
        rows = rng.integers(0, alpha, size=(2 * n_pairs, n), dtype=np.uint8)
        want = sum(orx.tb_cycle(W, size, rows[2 * q], rows[2 * q + 1], n, skip_idle=bool(q & 1))["total_cycles"]
                   for q in range(n_pairs))
        got = rx.match(nfa, rows, mode=rx.MODE_TB_COMPAT, collect_stats=2)
        if got["stats"]["tb_cycles"] == 0:
            continue  # some stream outgrew the pack kernel's list and was handed off: prediction unavailable
        checked += 1
        assert got["stats"]["tb_cycles"] == want, (trial, got["stats"]["tb_cycles"], want)
        with pytest.raises(rx.RxError):
            rx.match(nfa, rows[:1], mode=rx.MODE_TB_COMPAT, collect_stats=2)  # odd number of streams
    assert checked >= 10


def test_lazy_dfa_cache_is_persistent_and_resettable(rx, orx, automata, traces):
    """RX_KERNEL_DFA: the subset-construction cache grows on the device, persists across launches of the same
    automaton handle, gives identical results cold, warm and after a reset, also on streams it has never seen."""
    W, size = automata["snort_16"]
    nfa = rx.Nfa.from_words(W, size)
    wl = rx.workloads
    lo, hi = traces[("snort_16", "lo")], traces[("snort_16", "hi")]
    a = wl.trace_windows(lo, hi, 900, 700)
    b = wl.trace_windows(lo, hi, 900, 700, first=5000)
    ref_a, ref_b = orx.match_batch(W, size, a), orx.match_batch(W, size, b)
    assert nfa.dfa_info(0) == (0, 0)
    cold = rx.match(nfa, a, kernel=rx.KERNEL_DFA, collect_stats=True)
    s1, t1 = nfa.dfa_info(0)
    assert s1 > 100 and t1 > s1
    warm = rx.match(nfa, a, kernel=rx.KERNEL_DFA, collect_stats=True)
    assert nfa.dfa_info(0) == (s1, t1)                     # nothing new to build
    other = rx.match(nfa, b, kernel=rx.KERNEL_DFA, collect_stats=True)
    assert nfa.dfa_info(0)[0] >= s1
    nfa.dfa_reset(0)
    assert nfa.dfa_info(0) == (1, 0)
    again = rx.match(nfa, b, kernel=rx.KERNEL_DFA, collect_stats=True)
    for got, ref in ((cold, ref_a), (warm, ref_a), (other, ref_b), (again, ref_b)):
        check_equal(rx, orx, got, ref, "dfa")


def test_pipelined_host_to_host_run(rx, orx, automata, traces, gpu_nfas):
    """rx_plan_run (blocks of streams pipelined over HIP streams, page-locked caller buffers) == upload + launch +
    download == the oracle: several blocks with events on both sides of every block boundary, one block, ragged length,
    statistics and per-stream counters, repeated runs on one plan with the same and with a new input array."""
    W, size = automata["snort_16"]
    wl = rx.workloads
    lo, hi = traces[("snort_16", "lo")], traces[("snort_16", "hi")]
    for ns, sl, kw in ((40000, 300, dict()), (40000, 300, dict(collect_stats=True, want_match_count=True)),
                       (9000, 517, dict(kernel=rx.KERNEL_SYM_WAVE)), (700, 1024, dict(mode=rx.MODE_TB_COMPAT)),
                       (33000, 64, dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=32, flags=rx.host.OPT_FORCE_FOLD))):
        rows = wl.trace_windows(lo, hi, ns, sl, first=11)
        mode = kw.get("mode", rx.MODE_FULL)
        ref = orx.match_batch(W, size, rows, mode=mode, want_match_count=True, events_cap=1 << 22)
        p = rx.Plan(gpu_nfas["snort_16"], ns, sl, device=0, events_cap=1 << 21, **kw)
        for rep in range(3):
            data = rows if rep < 2 else rows.copy()          # same array twice (registered once), then a new one
            got = p.run(data)
            check_equal(rx, orx, got, ref, ("run", ns, sl, kw, rep), stats=bool(kw.get("collect_stats")))
        p.upload(rows)                                         # the step-by-step path still works on the same plan
        p.launch()
        check_equal(rx, orx, p.download(), ref, ("after run", ns, sl, kw), stats=bool(kw.get("collect_stats")))
        p.close()
    ev = ref["events"]
    assert len(ev) > 0


def test_final_sets_as_compact_lists(rx, orx, automata, traces, gpu_nfas):
    """rx_plan_run can return the final sets as lists (offset / count per stream into one array of states, ascending) instead
    of bitmask rows: expanded, they are the oracle's rows — several blocks of streams, one block, a batch whose streams end
    with many states active, a capacity that is too small (flagged, counts still exact), and rx_match's one-shot form."""
    wl = rx.workloads
    lo, hi = traces[("snort_16", "lo")], traces[("snort_16", "hi")]
    W, size = automata["snort_16"]
    nfa = gpu_nfas["snort_16"]
    for ns, sl, cap in ((70000, 200, 1 << 20), (700, 1024, 1 << 16), (40000, 300, 1 << 20)):
        rows = wl.trace_windows(lo, hi, ns, sl, first=5)
        ref = orx.match_batch(W, size, rows, events_cap=1 << 22)
        p = rx.Plan(nfa, ns, sl, device=0, events_cap=1 << 21)
        for rep in range(2):
            got = p.run(rows, compact_final=cap)
            assert got["final_active"] is None and not got["final_states_overflow"]
            assert int(got["final_cnt"].sum()) == len(got["final_states"])
            assert np.array_equal(rx.host.expand_final(got, nfa.nw64), ref["final_active"]), (ns, sl, rep)
            for s in (0, 1, ns // 2, ns - 1):                       # ascending within a stream
                seg = got["final_states"][got["final_off"][s]:got["final_off"][s] + got["final_cnt"][s]]
                assert np.all(np.diff(seg.astype(np.int64)) > 0)
            assert got["n_events"] == ref["n_events"]
        rows_form = p.run(rows)                                      # and the rows again on the same plan
        assert np.array_equal(rows_form["final_active"], ref["final_active"])
        p.close()
    # one-shot form; with a caller-supplied start set the lists are refused (that path downloads rows)
    rows = wl.trace_windows(lo, hi, 900, 333)
    ref = orx.match_batch(W, size, rows)
    got = rx.match(nfa, rows, compact_final=1 << 14)
    assert got["final_active"] is None and np.array_equal(rx.host.expand_final(got, nfa.nw64), ref["final_active"])
    with pytest.raises(rx.RxError):
        rx.match(nfa, rows, compact_final=1 << 14, init_active=ref["final_active"])
    # capacity too small: flagged; the counts are still what the sets hold
    ns, sl = 5000, 256
    rows = wl.trace_windows(lo, hi, ns, sl)
    ref = orx.match_batch(W, size, rows)
    want = np.array([bin(int(w)).count("1") for w in ref["final_active"].reshape(-1)]).reshape(ns, -1).sum(axis=1)
    p = rx.Plan(nfa, ns, sl, device=0)
    got = p.run(rows, compact_final=1000)
    assert got["final_states_overflow"] and np.array_equal(got["final_cnt"], want.astype(np.uint32)) and len(got["final_states"]) <= 1000
    p.close()


def test_events_capacity_overflow(rx, orx, automata, traces, gpu_nfas):
    W, size = automata["snort_16"]
    rows = np.stack([traces[("snort_16", "hi")][:4000]] * 8)
    ref = orx.match_batch(W, size, rows)
    got = rx.match(gpu_nfas["snort_16"], rows, events_cap=10)
    assert got["n_events"] == ref["n_events"] > 10 and got["events_overflow"] and len(got["events"]) == 10
    assert np.array_equal(got["match_count_total"], ref["match_count_total"])  # counters never overflow
    none = rx.match(gpu_nfas["snort_16"], rows, events_cap=0, want_anymatch=True)
    assert none["n_events"] == ref["n_events"] and np.array_equal(none["anymatch"], ref["anymatch"])


def test_full_size_config3_properties(rx, orx, automata, traces, gpu_nfas, kernels):
    """BASELINE configs[2] at full size (65 536 x 1 KB, distribution T): all kernels agree bit-for-bit on
    every output, a seeded sample of 3 072 streams equals the oracle, and size-independent invariants
    hold (event order, counters = histogram of events, any-match bits = event passes)."""
    W, size = automata["snort_16"]
    wl = rx.workloads
    ns, sl = 65536, 1024
    rows = wl.trace_windows(traces[("snort_16", "lo")], traces[("snort_16", "hi")], ns, sl)
    outs = [rx.match(gpu_nfas["snort_16"], rows, **k, events_cap=1 << 21, collect_stats=True) for k in kernels]
    a = outs[0]
    for b in outs[1:]:
        for k in ("events", "match_count_total", "anymatch", "final_active"):
            assert np.array_equal(a[k], b[k]), k
        assert {k: v for k, v in a["stats"].items() if k in ("n_events", "sum_active", "sum_edges", "alg_bytes")} == \
               {k: v for k, v in b["stats"].items() if k in ("n_events", "sum_active", "sum_edges", "alg_bytes")}
    # the builds that run when no statistics are asked for (what bench.py times), incl. AUTO's choice
    for k in (dict(kernel=rx.KERNEL_AUTO), dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=13),
              dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=16), dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=32)):
        b = rx.match(gpu_nfas["snort_16"], rows, **k, events_cap=1 << 21)
        for f in ("events", "match_count_total", "anymatch", "final_active"):
            assert np.array_equal(a[f], b[f]), (k, f)
    ev = a["events"]
    assert a["n_events"] == len(ev) > 10000
    order = np.lexsort((ev["state"], ev["k"], ev["stream"]))
    assert np.array_equal(order, np.arange(len(ev)))
    assert np.array_equal(np.bincount(ev["state"], minlength=size).astype(np.uint64), a["match_count_total"])
    bits = np.zeros_like(a["anymatch"])
    np.bitwise_or.at(bits, (ev["stream"], ev["k"] >> 5), (np.uint32(1) << (ev["k"] & 31)).astype(np.uint32))
    assert np.array_equal(bits, a["anymatch"])
    rng = np.random.default_rng(3)
    pick = np.sort(rng.choice(ns, size=3072, replace=False))
    ref = orx.match_batch(W, size, rows[pick])
    sel = ev[np.isin(ev["stream"], pick)]
    remap = np.searchsorted(pick, sel["stream"]).astype(np.uint32)
    sel = sel.copy()
    sel["stream"] = remap
    assert np.array_equal(sel, ref["events"].astype(sel.dtype))
    assert np.array_equal(a["final_active"][pick], ref["final_active"])
    assert np.array_equal(a["anymatch"][pick], ref["anymatch"])


def _size_independent_checks(rx, nfa_size, a, ns):
    """Invariants that need no oracle: canonical event order, counters = histogram of the events, any-match bits =
    the passes of the events, every event inside the batch."""
    ev = a["events"]
    assert a["n_events"] == len(ev) and not a["events_overflow"]
    order = np.lexsort((ev["state"], ev["k"], ev["stream"]))
    assert np.array_equal(order, np.arange(len(ev)))
    assert int(ev["stream"].max()) < ns
    assert np.array_equal(np.bincount(ev["state"], minlength=nfa_size).astype(np.uint64), a["match_count_total"])
    bits = np.zeros_like(a["anymatch"])
    np.bitwise_or.at(bits, (ev["stream"], ev["k"] >> 5), (np.uint32(1) << (ev["k"] & 31)).astype(np.uint32))
    assert np.array_equal(bits, a["anymatch"])


def _oracle_sample_checks(orx, W, size, rows, a, n_pick, seed):
    rng = np.random.default_rng(seed)
    pick = np.sort(rng.choice(rows.shape[0], size=n_pick, replace=False))
    ref = orx.match_batch(W, size, rows[pick], events_cap=1 << 22)
    ev = a["events"]
    sel = ev[np.isin(ev["stream"], pick)].copy()
    sel["stream"] = np.searchsorted(pick, sel["stream"]).astype(np.uint32)
    assert np.array_equal(sel, ref["events"].astype(sel.dtype))
    assert np.array_equal(a["final_active"][pick], ref["final_active"])
    assert np.array_equal(a["anymatch"][pick][:, :ref["anymatch"].shape[1]], ref["anymatch"])


@pytest.mark.parametrize("workload", ["T", "U"])
def test_baseline_configs3_per_gpu_shape(rx, orx, automata, traces, gpu_nfas, workload):
    """BASELINE configs[3] at its per-GPU shape (1 Mi streams over 8 GPUs = 131 072 x 1 KB each), both seeded
    distributions, generated as the block a middle rank owns: AUTO's choice == the wavefront-per-stream slice kernel ==
    the north-star CSR kernel on every output, a seeded sample of 2 048 streams == the oracle, and the
    size-independent invariants."""
    W, size = automata["snort_16"]
    wl = rx.workloads
    ns, sl, first = 131072, 1024, 3 * 131072
    rows = wl.trace_windows(traces[("snort_16", "lo")], traces[("snort_16", "hi")], ns, sl, first=first) if workload == "T" \
        else wl.uniform(ns, sl, first=first)
    a = rx.match(gpu_nfas["snort_16"], rows, kernel=rx.KERNEL_AUTO, events_cap=1 << 22)
    assert rx.host.KERNEL_NAMES[a["stats"]["kernel_used"]] == "sym_pack"
    for k in (dict(kernel=rx.KERNEL_SYM_WAVE), dict(kernel=rx.KERNEL_CSR_WAVE)):
        b = rx.match(gpu_nfas["snort_16"], rows, **k, events_cap=1 << 22)
        for f in ("events", "match_count_total", "anymatch", "final_active"):
            assert np.array_equal(a[f], b[f]), (k, f)
    if workload == "T":
        assert a["n_events"] > 100000
        _size_independent_checks(rx, size, a, ns)
    else:
        assert a["n_events"] == len(a["events"])
    _oracle_sample_checks(orx, W, size, rows, a, 2048, 31)


def test_baseline_configs4_snort16_on_4kb_windows(rx, orx, automata, traces, gpu_nfas):
    """BASELINE configs[4] at its per-GPU shape with SURVEY 8(d)-5's PRIMARY stand-in: the shipped snort_16 table
    (9 514 ~ 10k states) on 131 072 x 4 096-byte T windows (offset rule mod (200000 - 4096)), as rank 5 of 8 owns them.
    AUTO == the wavefront-per-stream kernel on every output, oracle sample, invariants."""
    W, size = automata["snort_16"]
    wl = rx.workloads
    ns, sl, first = 131072, 4096, 5 * 131072
    rows = wl.trace_windows(traces[("snort_16", "lo")], traces[("snort_16", "hi")], ns, sl, first=first)
    assert np.array_equal(rows[1], traces[("snort_16", "hi")][(((first + 1) >> 1) * 977) % (200000 - 4096):][:4096])
    a = rx.match(gpu_nfas["snort_16"], rows, kernel=rx.KERNEL_AUTO, events_cap=1 << 22)
    b = rx.match(gpu_nfas["snort_16"], rows, kernel=rx.KERNEL_SYM_WAVE, events_cap=1 << 22)
    for f in ("events", "match_count_total", "anymatch", "final_active"):
        assert np.array_equal(a[f], b[f]), f
    assert a["n_events"] > 500000
    _size_independent_checks(rx, size, a, ns)
    _oracle_sample_checks(orx, W, size, rows, a, 512, 41)


def test_baseline_configs4_ruleset_standin_per_gpu_shape(rx, orx):
    """BASELINE configs[4], second stand-in, at the same per-GPU shape: the compiled 700-pattern rule set (10 396 states)
    on 131 072 x 4 096 bytes of pseudo-traffic."""
    wl = rx.workloads
    pats = wl.synthetic_ruleset()
    nfa = rx.Nfa.compile(pats)
    ns, sl = 131072, 4096
    rows = wl.ruleset_traffic(pats, ns, sl, first=2 * ns, workers=12)
    assert np.array_equal(rows[:3], wl.ruleset_traffic(pats, 3, sl, first=2 * ns))
    a = rx.match(nfa, rows, kernel=rx.KERNEL_AUTO, events_cap=1 << 22)
    assert not a["events_overflow"]
    b = rx.match(nfa, rows, kernel=rx.KERNEL_SYM_WAVE, events_cap=1 << 22)
    for f in ("events", "match_count_total", "anymatch", "final_active"):
        assert np.array_equal(a[f], b[f]), f
    _size_independent_checks(rx, nfa.size, a, ns)
    _oracle_sample_checks(orx, nfa.words, nfa.size, rows, a, 128, 43)


def test_full_size_ruleset_standin(rx, orx):
    """BASELINE configs[4] stand-in at bench size (compiled 10 396-state rule set, 16 384 x 4 KB): AUTO's choice
    (pack kernel with look-ahead pruning) == the wavefront-per-stream kernel on every output, a seeded sample of
    streams == the oracle, and the size-independent invariants hold."""
    wl = rx.workloads
    pats = wl.synthetic_ruleset()
    nfa = rx.Nfa.compile(pats)
    ns, sl = 16384, 4096
    rows = wl.ruleset_traffic(pats, ns, sl)
    a = rx.match(nfa, rows, kernel=rx.KERNEL_AUTO, events_cap=1 << 23)
    assert rx.host.KERNEL_NAMES[a["stats"]["kernel_used"]] == "sym_pack" and not a["events_overflow"]
    b = rx.match(nfa, rows, kernel=rx.KERNEL_SYM_WAVE, events_cap=1 << 23)
    for f in ("events", "match_count_total", "anymatch", "final_active"):
        assert np.array_equal(a[f], b[f]), f
    ev = a["events"]
    assert np.array_equal(np.bincount(ev["state"], minlength=nfa.size).astype(np.uint64), a["match_count_total"])
    bits = np.zeros_like(a["anymatch"])
    np.bitwise_or.at(bits, (ev["stream"], ev["k"] >> 5), (np.uint32(1) << (ev["k"] & 31)).astype(np.uint32))
    assert np.array_equal(bits, a["anymatch"])
    rng = np.random.default_rng(9)
    pick = np.sort(rng.choice(ns, size=192, replace=False))
    ref = orx.match_batch(nfa.words, nfa.size, rows[pick], events_cap=1 << 22)
    sel = ev[np.isin(ev["stream"], pick)].copy()
    sel["stream"] = np.searchsorted(pick, sel["stream"]).astype(np.uint32)
    assert np.array_equal(sel, ref["events"].astype(sel.dtype))
    assert np.array_equal(a["final_active"][pick], ref["final_active"])
    assert np.array_equal(a["anymatch"][pick], ref["anymatch"])


def test_automaton_too_large_for_lds_fails_loudly(rx):
    """The wave kernels keep two size-bit bitmasks per stream in LDS (160 KB per CU): beyond 650 000 states the
    automaton is refused with RX_ECAPACITY (-8) when it is loaded — before any index is built — and just below
    that limit every kernel still runs."""
    e = [(0, c, 1) for c in range(256)] + [(1, c, 1) for c in range(256)]
    with pytest.raises(rx.RxError) as err:
        rx.Nfa.from_words(build_words(700_000, e + [(1, 97, 699_998), (699_998, 98, 699_999)]), 700_000)
    assert err.value.code == -8
    size = 640_000
    nfa = rx.Nfa.from_words(build_words(size, e + [(1, 97, size - 2), (size - 2, 98, size - 1)]), size)
    rows = np.frombuffer(b"xxabxxab", np.uint8)
    for kern in (dict(kernel=rx.KERNEL_AUTO), dict(kernel=rx.KERNEL_SYM_PACK), dict(kernel=rx.KERNEL_CSR_WAVE),
                 dict(kernel=rx.KERNEL_SYM_WAVE)):
        got = rx.match(nfa, rows, **kern)
        assert [(int(v["k"]), int(v["state"])) for v in got["events"]] == [(4, size - 1), (8, size - 1)], kern


def test_sharded_entry_point_single_device(rx, orx, automata, traces, gpu_nfas):
    """rx_match_sharded with the one visible device listed twice: the partition/merge path of the C-ABI."""
    W, size = automata["snort_16"]
    rows = rx.workloads.trace_windows(traces[("snort_16", "lo")], traces[("snort_16", "hi")], 301, 700)
    ref = orx.match_batch(W, size, rows, want_match_count=True)
    got = rx.match_sharded(gpu_nfas["snort_16"], rows, devices=[0, 0, 0], want_match_count=True, collect_stats=True)
    check_equal(rx, orx, got, ref, "sharded")


def test_report_cli(rx, orx, automata, traces):
    """rx_report (C++ caller of the C-ABI) prints the testbench's lines for the shipped l7 pair."""
    import subprocess
    from conftest import DATA
    exe = os.path.join(os.path.dirname(rx.lib_path()), "rx_report")
    out = subprocess.check_output([exe, os.path.join(DATA, "CSR_BlockMem.coe"),
                                   os.path.join(DATA, "input_trace_lo_l-7_filter.mem"),
                                   os.path.join(DATA, "input_trace_hi_l-7_filter.mem")], stderr=subprocess.DEVNULL).decode()
    W, size = automata["l7"]
    c = orx.tb_cycle(W, size, traces[("l7", "lo")][:N + 1], traces[("l7", "hi")][:N + 1], N, skip_idle=True)
    assert out.rstrip("\n") == rx.testbench.format_report(c["match_count"], c["match_count_2"], c["total_cycles"],
                                                          10 * c["total_cycles"] + 22)


def test_plain_c_caller(rx, tmp_path):
    """A C99 program that only includes include/rxmatch.h and links librxmatch.so runs the App. B.4 known answer."""
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "abi_kat")
    libdir = os.path.dirname(rx.lib_path())
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "abi_kat.c"), "-o", exe, "-L", libdir, "-lrxmatch",
                           f"-Wl,-rpath,{libdir}"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "abi_kat ok" in out.stdout, out.stdout + out.stderr


def test_randomized_differential_short(rx, orx):
    """A slice of tools/fuzz_gpu.py: seeded random automata (raw tables, blow-ups, compiled regexes) x random inputs,
    six randomly chosen kernel variants per case, everything compared with the oracle."""
    import importlib.util
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("fuzz_gpu", os.path.join(ROOT, "tools", "fuzz_gpu.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    for trial in range(400):
        err = fz.one_case(np.random.default_rng([7, trial]), trial)
        assert err is None, err
