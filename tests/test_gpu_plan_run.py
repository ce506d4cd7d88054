"""GPU tests of the resident-plan / pipelined-call machinery around the kernels (include/rxmatch.h): one capacity for
the whole rx_plan_run call, nothing in flight after an error return, tuned plans that only enqueue, struct_size = 0
frozen at the ABI-1 layouts.  Results are still compared with the CPU oracle, bit-exactly.  Run with `-m gpu`."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def snort(rx, automata):
    W, size = automata["snort_16"]
    return rx.Nfa.from_words(W, size), W, size


def _skewed_batch(rx, traces, ns, sl, busy):
    """ns streams of sl bytes whose accept events all lie in the first `busy` streams: trace windows there, then
    streams of one repeated byte (the `.*` state alone stays active: checked against the oracle by the caller)."""
    rows = np.full((ns, sl), 0x7E, np.uint8)
    rows[:busy] = rx.workloads.trace_windows(traces[("snort_16", "lo")], traces[("snort_16", "hi")], busy, sl)
    return rows


def test_one_capacity_for_the_whole_call(rx, orx, traces, snort):
    """131 072 streams (rx_plan_run cuts them into 4 blocks) whose accept events ALL lie in the first 10 000 streams,
    i.e. in block 0: with events_cap = the exact total no event is lost and events_overflow stays 0 (round 2 gave each
    block a quarter of the capacity); the same for the compact final sets with final_states_cap = the exact total."""
    nfa, W, size = snort
    ns, sl, busy = 131072, 1024, 10000
    rows = _skewed_batch(rx, traces, ns, sl, busy)
    ref = orx.match_batch(W, size, rows, events_cap=1 << 20)
    n_ev = ref["n_events"]
    assert n_ev > 5000 and int(ref["events"]["stream"].max()) < busy
    n_fin = int(sum(bin(int(w)).count("1") for w in ref["final_active"].ravel()[np.nonzero(ref["final_active"].ravel())[0]]))
    p = rx.Plan(nfa, ns, sl, events_cap=n_ev)
    got = p.run(rows, compact_final=n_fin)
    assert got["n_events"] == n_ev and not got["events_overflow"]
    assert np.array_equal(got["events"], ref["events"].astype(got["events"].dtype))
    assert not got["final_states_overflow"] and len(got["final_states"]) == n_fin
    assert np.array_equal(rx.host.expand_final(got, nfa.nw64), ref["final_active"])
    assert np.array_equal(got["anymatch"][:, :ref["anymatch"].shape[1]], ref["anymatch"])
    # bitmask rows through the same path, and the one-shot call (which runs through it since round 2)
    got = p.run(rows)
    assert np.array_equal(got["final_active"], ref["final_active"]) and got["n_events"] == n_ev and not got["events_overflow"]
    p.close()
    got = rx.match(nfa, rows, events_cap=n_ev)
    assert not got["events_overflow"] and np.array_equal(got["events"], ref["events"].astype(got["events"].dtype))
    # one event short: the overflow is reported, the count stays exact, what is kept is in canonical order
    got = rx.match(nfa, rows, events_cap=n_ev - 1)
    assert got["events_overflow"] and got["n_events"] == n_ev and len(got["events"]) == n_ev - 1
    ev = got["events"]
    assert np.array_equal(np.lexsort((ev["state"], ev["k"], ev["stream"])), np.arange(len(ev)))
    # a list capacity that is too small: reported, counts exact
    p = rx.Plan(nfa, ns, sl, events_cap=n_ev)
    got = p.run(rows, compact_final=n_fin - 7)
    assert got["final_states_overflow"] and int(got["final_cnt"].sum()) == n_fin
    p.close()


def test_error_return_leaves_nothing_in_flight(rx, orx, traces, snort):
    """rx_plan_run fails AFTER block 0's kernels and copies were enqueued (RX_OPT_INJECT_RUN_FAULT): when the error code
    is back no stream of the plan has work left, so the caller may poison / release its page-locked arrays at once; the
    plan stays usable."""
    nfa, W, size = snort
    ns, sl = 70000, 512
    rows = rx.workloads.trace_windows(traces[("snort_16", "lo")], traces[("snort_16", "hi")], ns, sl)
    ref = orx.match_batch(W, size, rows, events_cap=1 << 20)
    p = rx.Plan(nfa, ns, sl, events_cap=1 << 20, flags=rx.host.OPT_INJECT_RUN_FAULT)
    data = rows.copy()
    with pytest.raises(rx.RxError) as err:
        p.run(data)
    assert err.value.code == -7 and "injected" in str(err.value)
    assert p.busy() & 7 == 0
    out = p._run_out
    for arr in (data, out.ev, out.am, out.fin):  # what a caller that gives up does with its buffers
        arr.view(np.uint8).fill(0xAB)
    p.close()
    # an error found before anything is enqueued (odd stream count for the lock-step pair statistics)
    p = rx.Plan(nfa, ns, sl, events_cap=1 << 20, collect_stats=2)
    with pytest.raises(rx.RxError) as err:
        p.run(rows[:ns - 1])
    assert err.value.code == -1 and p.busy() & 7 == 0
    got = p.run(rows)  # the plan is still good
    assert got["n_events"] == ref["n_events"] and np.array_equal(got["events"], ref["events"].astype(got["events"].dtype))
    assert np.array_equal(got["final_active"], ref["final_active"])
    p.close()


def test_tuned_plan_only_enqueues(rx, orx, traces, snort):
    """rx_plan_tune once, then 40 launches on a plan created with RX_OPT_NO_PROBE: every launch returns while its
    kernel is still running (the stream is busy right after the call — the library neither probes nor synchronises),
    the choice is AUTO's (the pack kernel at the tuned streams per wavefront), results == oracle."""
    import torch
    nfa, W, size = snort
    ns, sl = 32768, 1024
    rows = rx.workloads.trace_windows(traces[("snort_16", "lo")], traces[("snort_16", "hi")], ns, sl)
    ref = orx.match_batch(W, size, rows[:2048], events_cap=1 << 20)
    st = torch.cuda.Stream()
    d_rows = torch.from_numpy(rows).cuda()
    torch.cuda.synchronize()
    p = rx.Plan(nfa, ns, sl, events_cap=1 << 20, stream=st.cuda_stream, flags=rx.host.OPT_NO_PROBE)
    p.set_device_input(d_rows.data_ptr(), ns, sl, sl, keepalive=d_rows)
    p.tune()
    assert p.busy() == 0
    still_running = 0
    for _ in range(40):
        p.set_device_input(d_rows.data_ptr(), ns, sl, sl, keepalive=d_rows)  # a new batch of the tuned shape
        p.launch()
        still_running += 0 if st.query() else 1
        st.synchronize()
    assert still_running >= 36, still_running  # (a 0.5 ms kernel cannot have finished when a 10 us enqueue returns)
    got = p.download()
    assert got["stats"]["kernel_used"] == rx.KERNEL_SYM_PACK and got["stats"]["lanes_used"] in (8, 11, 13, 16)
    ev = got["events"][got["events"]["stream"] < 2048]
    assert np.array_equal(ev, ref["events"].astype(ev.dtype))
    assert np.array_equal(got["final_active"][:2048], ref["final_active"])
    p.close()
    # an untuned shape under RX_OPT_NO_PROBE: the default choice, no probe, same results
    p = rx.Plan(nfa, 2048, sl, events_cap=1 << 20, stream=st.cuda_stream, flags=rx.host.OPT_NO_PROBE)
    p.upload(rows[:2048])
    p.launch()
    got = p.download()
    assert got["stats"]["kernel_used"] == rx.KERNEL_SYM_PACK and got["stats"]["lanes_used"] == 16
    assert np.array_equal(got["events"], ref["events"].astype(got["events"].dtype))
    p.close()


def test_struct_size_zero_is_the_abi1_layout(rx, orx, traces, snort):
    """A caller built against ABI 1 that left struct_size at 0: rx_opts is read up to `flags` only (garbage behind it is
    ignored) and rx_result is written up to stats.tb_cycles only (a canary behind it survives)."""
    nfa, W, size = snort
    h = rx.host
    L = h.lib()
    rows = np.ascontiguousarray(rx.workloads.trace_windows(traces[("snort_16", "lo")], traces[("snort_16", "hi")], 64, 512))
    ref = orx.match_batch(W, size, rows)
    o = h._mk_opts(-1, rx.MODE_FULL, rx.KERNEL_AUTO, None, 0, 0)
    o.struct_size = 0
    o.flags = 0xFFFFFFFF  # not part of the ABI-1 struct: must not be read (bit 128 would inject a fault)
    abi1 = h._Result.stats.offset + h._Stats.lanes_used.offset
    buf = (C.c_uint8 * C.sizeof(h._Result))()
    C.memset(buf, 0xC5, C.sizeof(h._Result))
    C.memset(buf, 0, abi1)
    r = h._Result.from_buffer(buf)
    ev = np.zeros(4096, h.EVENT_DT)
    r.struct_size = 0
    r.events, r.events_cap = ev.ctypes.data, len(ev)
    rc = L.rx_match(nfa._h, rows.ctypes.data, rows.shape[0], rows.shape[1], rows.shape[1], None, C.byref(o), C.byref(r))
    assert rc == 0
    assert r.n_events == ref["n_events"] and np.array_equal(ev[:r.n_events], ref["events"].astype(ev.dtype))
    assert int(r.stats.n_events) == ref["n_events"]
    assert bytes(buf)[abi1:] == b"\xC5" * (C.sizeof(h._Result) - abi1)


def test_evictions_rows_and_compact_lists(rx, orx):
    """Streams that outgrow the pack kernel's wave-wide list at different passes (or never) inside one wavefront: each
    leaves ALONE (its neighbours stay on the pack kernel and run the pass again), and the final sets — as rows written
    once from LDS, and as compact lists written by the match kernels themselves — equal the oracle's, as do events and
    the any-match bitmap.  Plain, pruned and FOLD builds, few and many streams per wavefront."""
    import sys
    sys.path.insert(0, __import__("os").path.dirname(__file__))
    from nfa_util import late_blowup_nfa
    W, size = late_blowup_nfa(220)
    nfa = rx.Nfa.from_words(W)
    base = b"xabxab..abYab"
    ns, sl = 333, 160
    rows = np.zeros((ns, sl), np.uint8)
    for s in range(ns):
        txt = bytearray((base * 16)[:sl])
        if s % 5 in (1, 3):                  # two in five blow up, at different passes; some come back down ('B')
            at = 5 + (s * 7) % 120
            txt[at:at + 6] = b"ZYYYab" if s % 2 else b"ZYYBab"
        rows[s] = np.frombuffer(bytes(txt), np.uint8)
    ref = orx.match_batch(W, size, rows)
    assert ref["stats"]["max_active"] > 200 and ref["n_events"] > 1000
    n_fin = int(np.unpackbits(ref["final_active"].view(np.uint8)).sum())
    H = rx.host
    for kw in (dict(kernel=rx.KERNEL_AUTO), dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=13), dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=4),
               dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=32), dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=16, flags=H.OPT_FORCE_FOLD),
               dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=64, flags=H.OPT_FORCE_FOLD | H.OPT_FORCE_PRUNE),
               dict(kernel=rx.KERNEL_SYM_PACK, group_lanes=8, flags=H.OPT_FORCE_PRUNE), dict(kernel=rx.KERNEL_SYM_WAVE),
               dict(kernel=rx.KERNEL_SYM_GROUP, group_lanes=4), dict(kernel=rx.KERNEL_SYM_REG)):
        p = rx.Plan(nfa, ns, sl, events_cap=1 << 20, **kw)
        got = p.run(rows)
        assert got["n_events"] == ref["n_events"] and np.array_equal(got["events"], ref["events"].astype(got["events"].dtype)), kw
        assert np.array_equal(got["final_active"], ref["final_active"]), kw
        assert np.array_equal(got["anymatch"][:, :ref["anymatch"].shape[1]], ref["anymatch"]), kw
        got = p.run(rows, compact_final=n_fin)
        assert not got["final_states_overflow"] and len(got["final_states"]) == n_fin, kw
        assert np.array_equal(H.expand_final(got, nfa.nw64), ref["final_active"]), kw
        for s in range(ns):  # ascending within every stream
            st = got["final_states"][got["final_off"][s]:got["final_off"][s] + got["final_cnt"][s]]
            assert np.all(st[1:] > st[:-1]), (kw, s)
        got = p.run(rows, compact_final=max(n_fin // 2, 1))  # capacity too small: counts exact, nothing written beyond it
        assert got["final_states_overflow"] and int(got["final_cnt"].sum()) == n_fin, kw
        p.close()
