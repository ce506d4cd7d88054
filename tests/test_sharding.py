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


def _worker(rank, world, port, n_streams, stream_len, q, use_hip=False):
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
    if use_hip:  # the product path: every rank loads the automaton file itself, uploads its own table, owns its plan
        nfa = rx.Nfa.load_coe(wl.SNORT_COE)
        matcher = lambda rows: rx.match(nfa, rows, device=0, kernel=rx.KERNEL_AUTO)  # noqa: E731
    else:
        matcher = lambda rows: orx.match_batch(W, size, rows, nthreads=1)  # noqa: E731
    first, count, res = rx.sharding.run_sharded(
        matcher, lambda f, c: wl.trace_windows(lo, hi, c, stream_len, first=f), n_streams, rank, world)
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


@pytest.mark.gpu
def test_two_rank_gloo_with_the_hip_matcher_on_one_gpu(rx, orx):
    """The N>1 path of bench.py with the REAL matcher: two processes (gloo for the report scalars), both on GPU 0, each
    loading the automaton file, uploading its own table copy and matching only its contiguous block through the C-ABI;
    the concatenation must equal the single-process oracle result.  (Two GPUs are not available to the tests; what can
    be covered — two processes sharing one automaton file and one device, plan lifetimes, shard arithmetic — is.)"""
    import torch.multiprocessing as mp
    n_streams, stream_len, world = 4099, 700, 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_streams, stream_len, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    wl = rx.workloads
    W = orx.load_coe(wl.SNORT_COE)
    size = orx.infer_size(W)
    lo, hi = orx.load_mem(wl.TRACES[("snort_16", "lo")]), orx.load_mem(wl.TRACES[("snort_16", "hi")])
    ref = orx.match_batch(W, size, wl.trace_windows(lo, hi, n_streams, stream_len))
    assert [(g[1], g[2]) for g in got] == [(0, 2050), (2050, 2049)]
    ev = np.concatenate([g[3] for g in got])
    assert np.array_equal(ev, ref["events"].astype(ev.dtype))
    assert np.array_equal(np.concatenate([g[4] for g in got]), ref["final_active"])
    for g in got:
        assert g[5] == 1.5 and g[6] == ref["n_events"] > 0 and g[7] == n_streams * stream_len


def _bench_self_launch(rx):
    """`python bench.py --gpus 2` invoked directly (no torchrun around it) must start its ranks itself, before any
    GPU call.  Here there is no GPU: the ranks start, fail at torch.cuda.set_device, and the launcher's failure comes
    back — what matters is that the parent did not raise SystemExit('launch with torch.distributed.run')."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--streams-per-gpu", "64", "--stream-len", "64", "--dist-backend", "gloo", "--same-device",
                        "--no-cpu-baseline", "--no-second-distribution"], env=env, capture_output=True, text=True, timeout=600)
    blob = r.stdout + r.stderr
    assert "launch with" not in blob
    try:
        have_gpu = rx.host.device_count() > 0
    except rx.RxError:
        have_gpu = False
    if have_gpu:
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        import json
        d = json.loads(line)
        assert r.returncode == 0 and d["n_gpus"] == 2 and d["value"] > 0
    else:
        assert r.returncode != 0 and ("torch.distributed" in blob or "ChildFailedError" in blob or "rank" in blob.lower())


def test_bench_starts_its_own_ranks(rx):
    _bench_self_launch(rx)


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_on_the_gpu(rx):
    _bench_self_launch(rx)


@pytest.mark.gpu
def test_rccl_report_path_with_one_rank():
    """The collective calls of bench.py's N>1 path on RCCL itself (backend "nccl"), as a world of one rank on the box's GPU:
    init with device_id, barrier, the report's all-reduces and all_gather, destroy."""
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_one_rank.py")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "nccl ok 1.5 3 4 [0.75]" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_bench_config3_point_at_one_gpu():
    """`bench.py --gpus 1 --config 3`: the N = 1 point of the configs[3] sweep (131 072 x 1 KB per GPU) exists and its line
    carries the per-rank figures a sweep's efficiency is read from."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "3", "--steps", "5", "--warmup", "2",
                        "--no-cpu-baseline", "--no-second-distribution"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["baseline_config_index"] == 3 and d["config"]["streams_per_gpu"] == 131072 and d["n_gpus"] == 1
    assert len(d["per_rank_kernel_ms"]) == 1 and d["per_gpu_gbit_s"][0] > 100 and d["value"] > 100


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
