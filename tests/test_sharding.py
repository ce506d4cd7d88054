"""Multi-GPU path, rehearsed on CPU: 2 processes, gloo.  Each rank builds and matches ONLY its
contiguous block of streams (no data-path collective); the scalar report reduces over ranks.  The
matcher plugged in here is the oracle (this is a test of partitioning/merging, not of the kernel)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_shard_ranges(rx):
    sr = rx.sharding.shard_range
    for n in (0, 1, 7, 8, 9, 65536, 1048576 + 3):
        for world in (1, 2, 3, 8):
            blocks = [sr(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == n
            for (f0, c0), (f1, _) in zip(blocks, blocks[1:]):
                assert f0 + c0 == f1                      # contiguous
            counts = [c for _, c in blocks]
            assert max(counts) - min(counts) <= 1 and counts == sorted(counts, reverse=True)  # remainder to low ranks
    with pytest.raises(ValueError):
        sr(10, 2, 2)


def _worker(rank, world, port, n_streams, stream_len, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import torch
    import torch.distributed as dist
    from oracle import orx
    rx = importlib.import_module("regex-fpga_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wl = rx.workloads
    W = orx.load_coe(wl.SNORT_COE)
    size = orx.infer_size(W)
    lo, hi = orx.load_mem(wl.TRACES[("snort_16", "lo")]), orx.load_mem(wl.TRACES[("snort_16", "hi")])
    first, count, res = rx.sharding.run_sharded(
        lambda rows: orx.match_batch(W, size, rows, nthreads=1),
        lambda f, c: wl.trace_windows(lo, hi, c, stream_len, first=f), n_streams, rank, world)
    dist.barrier()
    sec, ev, nbytes = rx.sharding.reduce_report(dist, torch.device("cpu"), 0.5 + rank, res["n_events"], count * stream_len)
    q.put((rank, first, count, res["events"], res["final_active"], sec, ev, nbytes))
    dist.destroy_process_group()


def test_two_rank_gloo_equals_single_process(rx, orx):
    import torch.multiprocessing as mp
    n_streams, stream_len, world = 301, 200, 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_streams, stream_len, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    wl = rx.workloads
    W = orx.load_coe(wl.SNORT_COE)
    size = orx.infer_size(W)
    lo, hi = orx.load_mem(wl.TRACES[("snort_16", "lo")]), orx.load_mem(wl.TRACES[("snort_16", "hi")])
    ref = orx.match_batch(W, size, wl.trace_windows(lo, hi, n_streams, stream_len), nthreads=2)
    assert [(g[1], g[2]) for g in got] == [(0, 151), (151, 150)]
    assert np.array_equal(np.concatenate([g[3] for g in got]), ref["events"])
    assert np.array_equal(np.concatenate([g[4] for g in got]), ref["final_active"])
    for g in got:  # every rank sees max time and global sums
        assert g[5] == 1.5 and g[6] == ref["n_events"] and g[7] == n_streams * stream_len
    assert ref["n_events"] > 0


def test_workload_generators_are_shardable(rx, traces):
    wl = rx.workloads
    lo, hi = traces[("snort_16", "lo")], traces[("snort_16", "hi")]
    full = wl.trace_windows(lo, hi, 64, 1024)
    assert np.array_equal(wl.trace_windows(lo, hi, 20, 1024, first=30), full[30:50])
    assert np.array_equal(full[5], hi[(2 * 977):(2 * 977) + 1024]) and np.array_equal(full[4], lo[(2 * 977):(2 * 977) + 1024])
    u = wl.uniform(16, 100)
    assert np.array_equal(wl.uniform(6, 100, first=7), u[7:13])
    # splitmix64 known answer: state0 = 20261004, first output little-endian
    st = (20261004 + 0x9E3779B97F4A7C15) & (2**64 - 1)
    z = st
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
    z ^= z >> 31
    assert u[0, :8].tobytes() == z.to_bytes(8, "little")
    assert abs(float(u.mean()) - 127.5) < 12
