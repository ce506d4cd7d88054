"""RCCL, once, on the one GPU the test box has: the exact calls of bench.py's N>1 report path — init_process_group
(backend "nccl" = RCCL, device_id), barrier, reduce_report's two all-reduces, gather_per_rank's all_gather, destroy — as a
world of ONE rank.  Run as a child process by tests/test_sharding.py (`-m gpu`); prints "nccl ok ..." on success."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

rx = importlib.import_module("regex-fpga_amd")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", device_id=dev)
dist.barrier()
sec, ev, nbytes = rx.sharding.reduce_report(dist, dev, 1.5, 3, 4)
per_rank = rx.sharding.gather_per_rank(dist, dev, 0.75)
torch.cuda.synchronize()
assert (sec, ev, nbytes) == (1.5, 3, 4) and per_rank == [0.75], (sec, ev, nbytes, per_rank)
print("nccl ok", sec, ev, nbytes, per_rank, flush=True)
dist.destroy_process_group()
