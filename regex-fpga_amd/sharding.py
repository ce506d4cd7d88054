"""Multi-GPU partitioning of a stream batch: contiguous blocks, remainder to the low ranks, NO
data-path collective (streams are independent units — SURVEY.md §8e; the reference itself has one
device).  One process per GPU (`torch.distributed`); the only traffic is a barrier and a few scalar
all-reduces for reporting (max time over ranks, event totals)."""


def shard_range(n_streams, rank, world):
    """-> (first, count) of the contiguous block owned by `rank`."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    per, rem = divmod(n_streams, world)
    first = rank * per + min(rank, rem)
    return first, per + (1 if rank < rem else 0)


def run_sharded(match_fn, make_rows, n_streams, rank, world):
    """Each rank builds ONLY its rows (make_rows(first, count)) and matches them with `match_fn`.
    Returns (first, count, result); stream ids in result['events'] are rebased to global ids."""
    first, count = shard_range(n_streams, rank, world)
    rows = make_rows(first, count)
    res = match_fn(rows)
    ev = res.get("events")
    if ev is not None and len(ev):
        ev = ev.copy()
        ev["stream"] += first
        res = dict(res, events=ev)
    return first, count, res


def reduce_report(dist, device, seconds, n_events, n_bytes):
    """max(seconds) and sum(events, bytes) over ranks via all_reduce; returns python numbers."""
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    c = torch.tensor([n_events, n_bytes], dtype=torch.int64, device=device)
    if dist is not None and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), int(c[0].item()), int(c[1].item())


def gather_per_rank(dist, device, value):
    """One float per rank, in rank order, on every rank (all_gather of a scalar): the per-GPU kernel times of the report."""
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist is None or not dist.is_initialized():
        return [float(value)]
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]
