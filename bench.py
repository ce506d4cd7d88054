#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric: input Gbit/s matched against the snort_16 NFA at N MI355X.

A "step" is one pass of the hot path over one resident batch: ONE launch of the match kernel over
`streams-per-gpu` streams x `stream-len` bytes that already sit in HBM.

    python bench.py                          # N=1: BASELINE configs[2], 65 536 x 1 KB, distribution T
    python bench.py --gpus 8                 # spawns 8 ranks itself; the SAME per-GPU shape (65 536 x 1 KB each): weak scaling
    python bench.py --gpus N --config 3      # N = 1, 2, 4, 8: the sweep at configs[3]'s per-GPU shape (131 072 x 1 KB each;
                                             # at N = 8 that is BASELINE configs[3] itself, 1 Mi streams)
    python bench.py --gpus 8 --config 4      # configs[4] stand-in: snort_16 on 4 KB T windows, 131 072 per GPU
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8   # what the driver does: same result

With N>1 every rank (one process per GPU) owns its own contiguous block of streams of the same generator —
weak scaling, no data-path collective; the only traffic is the barrier and the scalar all-reduces of the report.
Invoked directly with --gpus N>1 (no WORLD_SIZE in the environment) this script starts the N ranks itself as
children (`python -m torch.distributed.run`) BEFORE it touches torch.cuda / HIP and relays rank 0's line.

Prints ONE JSON line (rank 0).  `roofline`: `frac` = compulsory HBM bytes of one launch (input + outputs + table
once) / mean hipEvent duration of the kernel in the timed region / 8 TB/s — the kernel is nowhere near HBM-bound
(the table is cache-resident), which is what the figure says; what bounds it is in `roofline.limiter` and, when the
tracked rocprofv3 summary under profiles/ was taken from this very kernel source, `roofline.pmc`.  SURVEY.md §8(d)'s
FPGA-style algorithmic byte count (whole rows re-read per active state) is reported separately as
`roofline.effective_vs_fpga_row_bytes`; `north_star_form` times the wavefront-per-stream kernel that really reads
those rows from the unchanged CSR.
"""
import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

# BASELINE.json configs by index: (streams per GPU, stream length, workload, description)
CONFIGS = {
    2: (65536, 1024, "T", "BASELINE configs[2]"),
    3: (131072, 1024, "T", "BASELINE configs[3] (1 Mi streams over 8 GPUs = 131 072 per GPU)"),
    4: (131072, 4096, "T", "BASELINE configs[4] stand-in (SURVEY 8d-5: the shipped snort_16 table itself, 9 514 ~ 10k states, "
                           "on 4 096-byte T windows; no rule files exist offline)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=None,
                    help="BASELINE.json configs index, read as the PER-GPU shape at every --gpus N (default 2: a sweep over "
                         "N then keeps the per-GPU work fixed; --config 3 = 131 072 x 1 KB per GPU, configs[3] itself at N = 8)")
    ap.add_argument("--streams-per-gpu", type=int, default=None)
    ap.add_argument("--stream-len", type=int, default=None)
    ap.add_argument("--workload", choices=["T", "U", "R", "L"], default=None,
                    help="T trace windows (headline), U uniform bytes, R synthetic ~10k-state ruleset (second stand-in for "
                         "configs[4]), L the other shipped automaton (l7-filter) on windows of its own traces")
    ap.add_argument("--kernel", default="auto", choices=["auto", "csr_wave", "sym_wave", "sym_group", "sym_pack", "dfa", "sym_reg"])
    ap.add_argument("--group-lanes", type=int, default=0)
    ap.add_argument("--flags", type=int, default=0, help="rx_opts.flags (RX_OPT_* bits: A/B and diagnostic switches)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="cap on oracle threads (box share: 16 per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--all-kernels", action="store_true", help="also time the other kernels (extra keys)")
    ap.add_argument("--no-second-distribution", action="store_true",
                    help="skip the extra measurements (distribution U, single stream, north-star form, host-to-host)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL) for real multi-GPU runs; gloo only to rehearse ranks on fewer GPUs")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    a = ap.parse_args()
    cfg = a.config if a.config is not None else 2
    ns, sl, wl, desc = CONFIGS[cfg]
    a.config = cfg
    a.config_desc = desc if (a.streams_per_gpu is None and a.stream_len is None and a.workload is None) else \
        f"custom shape (preset {cfg} overridden)"
    a.streams_per_gpu = a.streams_per_gpu or ns
    a.stream_len = a.stream_len or sl
    a.workload = a.workload or wl
    return a


def spawn_ranks(a):
    """--gpus N>1 invoked directly: start the N ranks as CHILDREN before this process touches the GPU (a process that
    has initialised HIP must never exec), relay rank 0's JSON line, exit with the launcher's code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1])
    else:
        sys.stdout.write(proc.stdout)
    sys.exit(proc.returncode)


def kernel_source_sha16():
    h = hashlib.sha256()
    for f in ("rx_kernels.hip", "rx_internal.hpp"):
        h.update(open(os.path.join(ROOT, "regex-fpga_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def make_rows(rx, workload, first, count, stream_len, traces):
    wl = rx.workloads
    if workload in ("T", "L"):
        return wl.trace_windows(traces[0], traces[1], count, stream_len, first=first)
    if workload == "R":
        return wl.ruleset_traffic(traces, count, stream_len, first=first)
    return wl.uniform(count, stream_len, first=first)


def time_kernel(plan, steps):
    for _ in range(steps):
        plan.launch()
    n, s, mn, mx = plan.kernel_times()
    return s / max(n, 1), mn, mx


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)  # does not return
    import numpy as np
    import torch  # first: librxmatch must bind to the HIP runtime torch already loaded
    import torch.distributed as dist
    rx = importlib.import_module("regex-fpga_amd")
    rx.host.lib()  # fails loudly if the HIP extension is missing — there is no fallback

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.same_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rdev = dev if a.dist_backend == "nccl" else torch.device("cpu")  # where the report scalars are reduced
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")

    wl = rx.workloads
    kern = {"auto": rx.KERNEL_AUTO, "csr_wave": rx.KERNEL_CSR_WAVE, "sym_wave": rx.KERNEL_SYM_WAVE,
            "sym_group": rx.KERNEL_SYM_GROUP, "sym_pack": rx.KERNEL_SYM_PACK, "dfa": rx.KERNEL_DFA,
            "sym_reg": rx.KERNEL_SYM_REG}[a.kernel]
    if a.workload == "R":  # second stand-in for configs[4]: synthetic ruleset compiled by rx_compile_patterns
        traces = wl.synthetic_ruleset()
        nfa = rx.Nfa.compile(traces)
    elif a.workload == "L":
        nfa = rx.Nfa.load_coe(wl.L7_COE)
        traces = (rx.load_mem(wl.TRACES[("l7", "lo")]), rx.load_mem(wl.TRACES[("l7", "hi")]))
    else:
        nfa = rx.Nfa.load_coe(wl.SNORT_COE)
        traces = (rx.load_mem(wl.TRACES[("snort_16", "lo")]), rx.load_mem(wl.TRACES[("snort_16", "hi")]))
    ns, sl = a.streams_per_gpu, a.stream_len
    first = rank * ns  # contiguous block per rank (sharding.shard_range of world*ns streams)
    rows = make_rows(rx, a.workload, first, ns, sl, traces)
    ev_cap = 1 << 22 if sl <= 1024 else 1 << 24

    # inputs resident in HBM before the timed region (torch owns the buffer; plumbing only)
    torch.zeros(1, device=dev)  # context + allocator warm, so that the copy below times the copy
    torch.cuda.synchronize()
    t_h2d0 = time.perf_counter()
    d_rows = torch.from_numpy(rows).to(dev)  # pageable host memory, as a plain caller of rx_match() has
    torch.cuda.synchronize()
    h2d_s = time.perf_counter() - t_h2d0
    h_pin = torch.from_numpy(rows).pin_memory()
    t_h2d0 = time.perf_counter()
    d_rows.copy_(h_pin, non_blocking=True)
    torch.cuda.synchronize()
    h2d_pinned_s = time.perf_counter() - t_h2d0
    del h_pin
    stream = torch.cuda.current_stream().cuda_stream
    common = dict(mode=rx.MODE_FULL, kernel=kern, device=local, stream=stream, events_cap=ev_cap,
                  group_lanes=a.group_lanes, flags=a.flags)
    plan = rx.Plan(nfa, ns, sl, want_match_count=False, want_anymatch=True, want_final=True, **common)
    plan.set_device_input(d_rows.data_ptr(), ns, sl, sl, keepalive=d_rows)

    # algorithmic bytes of one launch: collect_stats build of the same kernel, untimed
    splan = rx.Plan(nfa, ns, sl, want_match_count=False, want_anymatch=True, want_final=True, collect_stats=True, **common)
    splan.set_device_input(d_rows.data_ptr(), ns, sl, sl, keepalive=d_rows)
    splan.launch()
    sres = splan.download()
    alg_bytes = sres["stats"]["alg_bytes"]
    n_events = sres["stats"]["n_events"]
    splan.close()

    for _ in range(a.warmup):
        plan.launch()
    plan.kernel_times()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        plan.launch()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    nk, ksum, kmin, kmax = plan.kernel_times()
    res = plan.download()
    assert res["stats"]["n_events"] == n_events, "timed kernel and stats kernel disagree"
    kernel_used = rx.host.KERNEL_NAMES[res["stats"]["kernel_used"]]
    variant = res["stats"].get("variant", "")
    sec, ev_total, bytes_total = rx.sharding.reduce_report(dist if world > 1 else None, rdev, t1 - t0, n_events,
                                                           ns * sl)
    kavg_ms = ksum / max(nk, 1)
    per_rank_ms = rx.sharding.gather_per_rank(dist if world > 1 else None, rdev, kavg_ms)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    gbit = 8.0 * bytes_total * a.steps / sec / 1e9
    npass = sl + 1
    compulsory = int(ns * sl + ns * ((npass + 31) // 32) * 4 + ns * nfa.nw64 * 8 + n_events * 12 + nfa.n_words * 4)
    hbm_gbs = compulsory / (kavg_ms * 1e-3) / 1e9
    eff_gbs = alg_bytes / (kavg_ms * 1e-3) / 1e9

    # HBM bytes per launch measured with rocprofv3 PMC passes (profiles/, tools/summarize_profile.py): valid only for
    # exactly this kernel source, kernel variant and shape — anything else is reported as null, loudly
    traffic, pmc, traffic_note = None, None, "no tracked rocprofv3 PMC summary for this kernel variant and shape"
    tkey = f"{kernel_used}{(':' + variant) if variant else ''}:{a.workload}:{ns}x{sl}"
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            ent = json.load(open(tpath)).get(tkey)
        except Exception:
            ent = None
        if ent is not None:
            if ent.get("src_sha16") == kernel_source_sha16():
                # FETCH_SIZE counts the input windows (16 B per lane, 64 B per stream and load) at exactly HALF their bytes on
                # gfx950 (tools/write_calib.hip: 32 780 KB for 64 MiB read in this very pattern; MI355X_MICROARCH.md says the same
                # for wide streaming reads) and WRITE_SIZE is exact for the kernel's stores: the input's other half is added back
                traffic, pmc = ent.get("hbm_bytes_per_launch") + (ns * sl) // 2, ent.get("pmc")
                traffic_note = (ent.get("source", "") + f"; raw counters: FETCH_SIZE {ent.get('fetch_bytes')} B + WRITE_SIZE "
                                f"{ent.get('write_bytes')} B; corrected = raw + half of the {ns * sl} input bytes, which the "
                                "counter tallies at 64 B per 128-B request (tools/write_calib.hip, profiles/r03_sym_pack_auto)")
            else:
                traffic_note = (f"profiles/traffic.json[{tkey}] was measured on another version of the kernel source "
                                f"({ent.get('src_sha16')}): STALE, not reported")
    shape_desc = f"{ns} x {sl} B streams per GPU"
    dist_desc = {"T": "windows of the reference snort_16 traces", "U": "splitmix64 uniform bytes"}.get(a.workload, "")
    if a.workload in ("T", "U"):
        workload = (f"snort_16 CSR NFA (9514 states, 79856 edges), {shape_desc}, distribution {a.workload} ({dist_desc}), "
                    f"full mode from reset, {a.config_desc}")
    elif a.workload == "L":
        workload = (f"l7-filter CSR NFA (the reference's other shipped table: {nfa.size} states, {nfa.nnz} edges), {shape_desc}: "
                    f"windows of its own lo/hi traces, full mode from reset")
    else:
        workload = (f"SECOND STAND-IN for configs[4]: synthetic 700-pattern ruleset compiled to one CSR NFA ({nfa.size} states, "
                    f"{nfa.nnz} edges), {shape_desc} of pseudo-traffic")
    out = {
        "metric": "input Gbit/s matched vs snort_16 NFA" if a.workload in ("T", "U") else
                  "input Gbit/s matched (NOT the headline automaton, see config.workload)",
        "value": round(gbit, 3), "unit": "Gbit/s",
        "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(sec / a.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": workload, "baseline_config_index": a.config, "kernel": kernel_used, "kernel_variant": variant,
                   "streams_per_gpu": ns, "stream_len": sl,
                   "parallelism": f"streams sharded over {a.gpus} GPU(s), contiguous blocks, no collective"},
        "roofline": {"bound": "hbm", "achieved": round(hbm_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(hbm_gbs / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_note": traffic_note,
                     "compulsory_hbm_bytes_per_launch": compulsory,
                     "kernel_ms_avg": round(kavg_ms, 4), "kernel_ms_min": round(kmin, 4), "kernel_ms_max": round(kmax, 4),
                     "limiter": "not HBM: five dependent memory round trips per pass (list entry, class byte, slice gather - ~900 cycles "
                                "outstanding, 528 of them inside the vector L1 although L2 answers in 138 - two filter atomics, appends) "
                                "with five in-order wavefronts per SIMD; 61 % of the wave-cycles wait on memory counters, no unit is "
                                "saturated on average (DESIGN.md 3.4, profiles/r03_issue_ceiling); compulsory HBM traffic is ~2 B per "
                                "input byte (input + final sets + bitmap)",
                     "pmc": pmc,
                     # SURVEY 8d's figure: what the FPGA design's row-by-row reads would have moved for the same work
                     "effective_vs_fpga_row_bytes": {"alg_bytes_per_launch": alg_bytes, "GBs": round(eff_gbs, 2),
                                                      "x_hbm_peak": round(eff_gbs / HBM_PEAK_GBS, 3),
                                                      "note": "FPGA-style whole-row bytes (1 + sum(8 + 4 deg) per pass + outputs) / "
                                                              "kernel time; this kernel reads one slice-index dword per active "
                                                              "state instead, so this is work done, not bytes moved"}},
        # every rank's mean kernel time (rank order) and the per-GPU rate it implies: scaling efficiency can be read off one line
        "per_rank_kernel_ms": [round(x, 4) for x in per_rank_ms],
        "per_rank_kernel_ms_min_max": [round(min(per_rank_ms), 4), round(max(per_rank_ms), 4)],
        "per_gpu_gbit_s": [round(8.0 * ns * sl / (x * 1e-3) / 1e9, 2) for x in per_rank_ms],
        "accept_events_per_launch": ev_total,
        "h2d_inclusive_gbit_s": round(8.0 * ns * sl / (h2d_s + kavg_ms * 1e-3) / 1e9, 3),
        "h2d_pinned_inclusive_gbit_s": round(8.0 * ns * sl / (h2d_pinned_s + kavg_ms * 1e-3) / 1e9, 3),
    }

    if world == 1 and not a.no_second_distribution:
        # host buffers in, host results out, through the library's own copies (pageable memory, resident plan): what a
        # caller of the C-ABI that does not keep its streams in HBM gets.  Reported beside `value`, never as `value`.
        hplan = rx.Plan(nfa, ns, sl, want_match_count=False, want_anymatch=True, want_final=True, **common)
        hplan.upload(rows)
        hplan.launch()
        hplan.download()  # warm: buffers allocated, AUTO decided
        t_h = time.perf_counter()
        hplan.upload(rows)
        hplan.launch()
        hres = hplan.download()
        serial_s = time.perf_counter() - t_h
        assert hres["stats"]["n_events"] == n_events
        # the same through rx_plan_run: blocks of streams pipelined over HIP streams, caller buffers page-locked once
        # (first call) and then moved by DMA without staging copies
        hplan.run(rows)
        hplan.run(rows)  # (the first call page-locks the buffers and lets AUTO probe; the second touches every page)
        host_to_host_s = 1e9
        for _ in range(3):
            t_h = time.perf_counter()
            hres = hplan.run(rows)
            host_to_host_s = min(host_to_host_s, time.perf_counter() - t_h)
        assert hres["stats"]["n_events"] == n_events and np.array_equal(hres["events"], res["events"])
        assert np.array_equal(hres["final_active"], res["final_active"]) and np.array_equal(hres["anymatch"], res["anymatch"])
        # every output again, the final sets as lists (offset + count per stream, states ascending) instead of 1.2 KB of
        # bitmask per stream: the same information in ~1 % of the bytes
        ccap = 1 << 22
        hplan.run(rows, compact_final=ccap)
        hplan.run(rows, compact_final=ccap)
        compact_s = 1e9
        for _ in range(3):
            t_h = time.perf_counter()
            cres = hplan.run(rows, compact_final=ccap)
            compact_s = min(compact_s, time.perf_counter() - t_h)
        assert cres["stats"]["n_events"] == n_events and not cres["final_states_overflow"]
        pick = np.arange(0, ns, max(ns // 512, 1))  # (the parity suite checks all of them; here a sample)
        sub = dict(final_off=cres["final_off"][pick], final_cnt=cres["final_cnt"][pick], final_states=cres["final_states"])
        assert np.array_equal(rx.host.expand_final(sub, nfa.nw64), res["final_active"][pick])
        out["host_to_host_compact_final_sets_gbit_s"] = round(8.0 * ns * sl / compact_s / 1e9, 3)
        out["compact_final_sets_bytes"] = int(cres["final_off"].nbytes + cres["final_cnt"].nbytes + cres["final_states"].nbytes)
        hplan.close()
        out["host_to_host_serial_pageable_gbit_s"] = round(8.0 * ns * sl / serial_s / 1e9, 3)
        # the link's floor for this call: input up + every output down over one PCIe link that carries ~56 GB/s in either
        # direction or both together (tools/copybw.py on the MI355X box); the final sets are 1.2 KB of bitmask per stream
        link_bytes = ns * sl + ns * nfa.nw64 * 8 + ns * ((npass + 31) // 32) * 4 + n_events * 12
        out["host_to_host_link_floor_gbit_s"] = round(8.0 * ns * sl / (link_bytes / 56e9) / 1e9, 1)
        nplan = rx.Plan(nfa, ns, sl, want_match_count=False, want_anymatch=True, want_final=False, **common)
        nplan.run(rows)
        nplan.run(rows)
        best = 1e9
        for _ in range(3):
            t_h = time.perf_counter()
            nres = nplan.run(rows)
            best = min(best, time.perf_counter() - t_h)
        out["host_to_host_no_final_sets_gbit_s"] = round(8.0 * ns * sl / best / 1e9, 3)
        assert nres["stats"]["n_events"] == n_events
        nplan.close()
        out["host_to_host_gbit_s"] = round(8.0 * ns * sl / host_to_host_s / 1e9, 3)

        # north-star form: ONE WAVEFRONT OWNS ONE STREAM and reads row_ptr pairs + whole rows from the unchanged CSR
        # (Design/FPGA.v:166-207, :227-714): here the algorithmic bytes ARE the bytes the kernel loads (from L1/L2)
        cp = rx.Plan(nfa, ns, sl, mode=rx.MODE_FULL, kernel=rx.KERNEL_CSR_WAVE, device=local, stream=stream, events_cap=ev_cap)
        cp.set_device_input(d_rows.data_ptr(), ns, sl, sl, keepalive=d_rows)
        time_kernel(cp, 1)
        cavg, cmin, cmax = time_kernel(cp, 3)
        cres = cp.download()
        assert cres["stats"]["n_events"] == n_events
        cp.close()
        out["north_star_form"] = {"kernel": "csr_wave (one wavefront per stream, unchanged CSR rows)",
                                  "kernel_ms_avg": round(cavg, 3), "gbit_s": round(8.0 * ns * sl / (cavg * 1e-3) / 1e9, 3),
                                  "csr_read_GBs": round(alg_bytes / (cavg * 1e-3) / 1e9, 1),
                                  "note": "row_ptr pairs + whole rows of every active state per pass, served by L1/L2 (table 357 KB); "
                                          "rocprofv3 summary: profiles/r02_csr_wave/"}

    if a.workload == "T" and world == 1 and not a.no_second_distribution:
        # SURVEY.md §8d asks for both seeded distributions; T above is the headline, U is reported beside it
        urows = make_rows(rx, "U", first, ns, sl, traces)
        d_u = torch.from_numpy(urows).to(dev)
        up = rx.Plan(nfa, ns, sl, collect_stats=True, **common)
        up.set_device_input(d_u.data_ptr(), ns, sl, sl, keepalive=d_u)
        up.launch()
        ualg = up.download()["stats"]["alg_bytes"]
        up.close()
        up = rx.Plan(nfa, ns, sl, **common)
        up.set_device_input(d_u.data_ptr(), ns, sl, sl, keepalive=d_u)
        time_kernel(up, 2)
        uavg, umin, umax = time_kernel(up, max(a.steps // 2, 3))
        ures = up.download()
        out["distribution_U"] = {"gbit_s": round(8.0 * ns * sl / (uavg * 1e-3) / 1e9, 3), "kernel_ms_avg": round(uavg, 4),
                                 "alg_bytes_per_launch": ualg, "eff_GBs": round(ualg / (uavg * 1e-3) / 1e9, 2),
                                 "kernel": rx.host.KERNEL_NAMES[ures["stats"]["kernel_used"]],
                                 "kernel_variant": ures["stats"].get("variant", ""),
                                 "accept_events_per_launch": ures["stats"]["n_events"]}
        up.close()
        del d_u
        # BASELINE configs[1]: ONE shipped trace as ONE stream, tb-compat (a single stream is a single dependency chain,
        # so this is a latency figure; the parity tests check its match vector)
        one = traces[1][:200000].reshape(1, -1)
        d_one = torch.from_numpy(one.copy()).to(dev)
        sp = rx.Plan(nfa, 1, one.shape[1], mode=rx.MODE_TB_COMPAT, kernel=kern, device=local, stream=stream,
                     events_cap=1 << 20, group_lanes=a.group_lanes, flags=a.flags)
        sp.set_device_input(d_one.data_ptr(), 1, one.shape[1], one.shape[1], keepalive=d_one)
        time_kernel(sp, 1)
        savg, _, _ = time_kernel(sp, 2)
        sres = sp.download()
        out["single_stream_config1"] = {"input": "input_trace_hi_snort_16.mem, 200 000 B, tb-compat", "kernel_ms": round(savg, 3),
                                        "mbit_s": round(8.0 * one.shape[1] / (savg * 1e-3) / 1e6, 2),
                                        "ns_per_pass": round(savg * 1e6 / (one.shape[1] - 1), 1),
                                        "accept_events": sres["stats"]["n_events"],
                                        "kernel": rx.host.KERNEL_NAMES[sres["stats"]["kernel_used"]],
                                        "kernel_variant": sres["stats"].get("variant", "")}
        sp.close()
        del d_one
        # small batches (a SIMD gets at most a few wavefronts): latency per pass is what counts; AUTO times the pack kernel
        # against one wavefront per stream on the batch itself
        small = {}
        for sn in (64, 1024, 4096):
            srows = make_rows(rx, "T", 0, sn, sl, traces)
            d_s = torch.from_numpy(srows).to(dev)
            bp = rx.Plan(nfa, sn, sl, **common)
            bp.set_device_input(d_s.data_ptr(), sn, sl, sl, keepalive=d_s)
            time_kernel(bp, 2)
            bavg, _, _ = time_kernel(bp, 5)
            bres = bp.download()
            small[f"{sn}x{sl}"] = {"kernel_ms_avg": round(bavg, 4), "gbit_s": round(8.0 * sn * sl / (bavg * 1e-3) / 1e9, 2),
                                   "kernel": rx.host.KERNEL_NAMES[bres["stats"]["kernel_used"]],
                                   "kernel_variant": bres["stats"].get("variant", "")}
            bp.close()
            del d_s
        out["small_batches_T"] = small
        # What RX_KERNEL_AUTO's probes cost a fresh plan (first launch of a shape: sample launches + timed candidates + stream
        # synchronisation), and what a launch costs once the shape is tuned (rx_plan_tune, RX_OPT_NO_PROBE): host wall clock
        # from the call to the completed kernel, 4 096 x 1 KB (a shape where AUTO times three candidates) and the headline shape
        probe = {}
        for sn in (4096, ns):
            srows = rows[:sn]
            d_s = torch.from_numpy(srows).to(dev)
            fp = rx.Plan(nfa, sn, sl, **common)
            fp.set_device_input(d_s.data_ptr(), sn, sl, sl, keepalive=d_s)
            torch.cuda.synchronize()
            t_p = time.perf_counter()
            fp.launch()
            fp.sync()
            first_ms = (time.perf_counter() - t_p) * 1e3
            fp.close()
            tp = rx.Plan(nfa, sn, sl, **dict(common, flags=a.flags | rx.host.OPT_NO_PROBE))
            tp.set_device_input(d_s.data_ptr(), sn, sl, sl, keepalive=d_s)
            tp.tune()
            tp.launch()
            tp.sync()
            t_p = time.perf_counter()
            tp.set_device_input(d_s.data_ptr(), sn, sl, sl, keepalive=d_s)
            tp.launch()
            tp.sync()
            tuned_ms = (time.perf_counter() - t_p) * 1e3
            tp.close()
            del d_s
            probe[f"{sn}x{sl}"] = {"first_launch_with_probes_ms": round(first_ms, 3), "launch_on_tuned_plan_ms": round(tuned_ms, 3)}
        out["auto_probe_cost"] = probe
        # Hand-off mix: what streams cost whose active set outgrows the pack kernel's wave-wide list.  One
        # stream in 64 enters a 220-state trap grafted onto the shipped table (workloads.table_with_trap; no input makes a
        # snort_16 stream do that on its own) and is finished by the wave kernel; since round 3 ONLY that stream leaves its
        # wavefront (before: all 13 streams of the wavefront).  Same table, same batch without the trapped streams beside it.
        tw, tsize = wl.table_with_trap(nfa.words, nfa.size)
        tnfa = rx.Nfa.from_words(tw, tsize)
        mix = {}
        for name, mrows in (("clean", rows), ("one_in_64_trapped", wl.handoff_mix(traces[0], traces[1], ns, sl, first=first))):
            d_m = torch.from_numpy(mrows).to(dev)
            mp = rx.Plan(tnfa, ns, sl, want_match_count=False, want_anymatch=True, want_final=True, **common)
            mp.set_device_input(d_m.data_ptr(), ns, sl, sl, keepalive=d_m)
            time_kernel(mp, 2)
            mavg, _, _ = time_kernel(mp, 5)
            mres = mp.download()
            mix[name] = {"kernel_ms_avg": round(mavg, 4), "gbit_s": round(8.0 * ns * sl / (mavg * 1e-3) / 1e9, 2),
                         "kernel_variant": mres["stats"].get("variant", ""), "accept_events": mres["stats"]["n_events"]}
            mp.close()
            del d_m
        mix["slowdown"] = round(mix["one_in_64_trapped"]["kernel_ms_avg"] / mix["clean"]["kernel_ms_avg"], 3)
        mix["note"] = ("snort_16 table + 222-state trap; the trapped streams (1.6 %) hold 222 states per pass and are finished behind "
                       "the pack kernel, one workgroup per stream; times are both launches together")
        out["handoff_mix_T"] = mix
        tnfa.close()

    if a.all_kernels:
        extra = {}
        for name, kid, gl in (("csr_wave", rx.KERNEL_CSR_WAVE, 0), ("sym_wave", rx.KERNEL_SYM_WAVE, 0),
                              ("sym_group1", rx.KERNEL_SYM_GROUP, 1), ("sym_group2", rx.KERNEL_SYM_GROUP, 2),
                              ("sym_group4", rx.KERNEL_SYM_GROUP, 4), ("sym_group8", rx.KERNEL_SYM_GROUP, 8),
                              ("sym_group16", rx.KERNEL_SYM_GROUP, 16), ("sym_pack4", rx.KERNEL_SYM_PACK, 4),
                              ("sym_pack8", rx.KERNEL_SYM_PACK, 8), ("sym_pack13", rx.KERNEL_SYM_PACK, 13),
                              ("sym_pack16", rx.KERNEL_SYM_PACK, 16), ("sym_pack24", rx.KERNEL_SYM_PACK, 24),
                              ("sym_pack32", rx.KERNEL_SYM_PACK, 32), ("dfa_warm", rx.KERNEL_DFA, 0)):
            p2 = rx.Plan(nfa, ns, sl, mode=rx.MODE_FULL, kernel=kid, device=local, stream=stream, events_cap=ev_cap,
                         group_lanes=gl, flags=a.flags)
            p2.set_device_input(d_rows.data_ptr(), ns, sl, sl, keepalive=d_rows)
            time_kernel(p2, 2)
            avg, mn, mx = time_kernel(p2, max(a.steps // 2, 3))
            extra[name] = {"kernel_ms_avg": round(avg, 4), "gbit_s": round(8.0 * ns * sl / (avg * 1e-3) / 1e9, 3)}
            p2.close()
        out["kernels"] = extra

    if not a.no_cpu_baseline and world == 1:  # the CPU baseline is reported at N=1 only (rank 0 is the only rank there)
        from oracle import orx  # checker / reported CPU baseline only
        W = nfa.words if a.workload == "R" else orx.load_coe(wl.L7_COE if a.workload == "L" else wl.SNORT_COE)
        size = nfa.size if a.workload == "R" else orx.infer_size(W)
        # threads = this process's CPU share (capped), sample sized for ~15 s of CPU work
        nthr = max(1, min(len(os.sched_getaffinity(0)), a.cpu_threads))
        t = time.perf_counter()
        orx.match_batch(W, size, rows[:256], mode=orx.MODE_FULL, nthreads=1, want_final=False)
        per_stream_s = (time.perf_counter() - t) / 256
        nsamp = int(min(ns, max(nthr * 64, 15.0 / per_stream_s)))
        t = time.perf_counter()
        ref = orx.match_batch(W, size, rows[:nsamp], mode=orx.MODE_FULL, nthreads=nthr, want_final=False,
                              events_cap=ev_cap)
        cpu_s = time.perf_counter() - t
        # cross-check the GPU's events on the sample
        gev = res["events"]
        gev = gev[gev["stream"] < nsamp]
        ok = bool(np.array_equal(gev, ref["events"].astype(gev.dtype)))
        out["cpu_baseline"] = {"value": round(8.0 * nsamp * sl / cpu_s / 1e9, 4), "unit": "Gbit/s",
                               "cores": ref["threads"], "kind": "port",
                               "sample": f"first {nsamp} of {ns} streams x {sl} B of the same batch, functional C "
                                         f"oracle (oracle/rx_oracle.c), {ref['threads']} threads, {cpu_s:.2f} s wall "
                                         f"(~{cpu_s * ref['threads']:.0f} core-s)",
                               "events_match_gpu_on_sample": ok}
        # RTL-equivalent baseline: the clock-accurate restatement of FPGA.v + Blk_Mem_tb on the reference's
        # OWN run — the whole shipped trace pair, every clock simulated (no idle fast-forward), 1 core — so the line carries
        # the number the testbench prints (`Total no. cycles`, testbench_BLK_Mem.sv:84).  ~20 s for snort_16.
        if a.workload == "R":
            m = min(3000, sl)
            pair = (rows[0], rows[1] if ns > 1 else rows[0])
            what, expect = "first %d bytes of streams 0+1" % m, None
        else:
            m = 200000
            pair = (traces[0][:m + 1], traces[1][:m + 1])
            what = ("the shipped l7-filter lo+hi pair" if a.workload == "L" else "the shipped snort_16 lo+hi pair") + \
                   ", all 199 999 passes Blk_Mem_tb runs before $finish"
            expect = 617518104 if a.workload == "L" else 2188184738
        t = time.perf_counter()
        cyc = orx.tb_cycle(W, size, pair[0], pair[1], m, skip_idle=False)
        cyc_s = time.perf_counter() - t
        out["cpu_baseline"]["rtl_model"] = {
            "kind": "clock-accurate C restatement of FPGA.v (no Verilator in the image)", "cores": 1,
            "total_cycles": cyc["total_cycles"], "total_cycles_expected": expect,
            "clocks_per_s": round(cyc["total_cycles"] / cyc_s), "wall_s": round(cyc_s, 2),
            "input_bit_s": round(2 * (m - 1) * 8 / cyc_s), "sample": what}
        assert expect is None or cyc["total_cycles"] == expect, "clock model disagrees with the survey's Total no. cycles"
        assert ok, "GPU events differ from the oracle on the CPU-baseline sample"
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
