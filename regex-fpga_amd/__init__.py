"""regex-fpga_amd — MI355X-native CSR-NFA multi-pattern matcher (host-side Python mirror).

The product is `librxmatch.so` (C-ABI in include/rxmatch.h; gfx950 kernels in csrc/).  This package
is the thin Python host over that C-ABI: `host` (ctypes binding), `testbench` (what the reference's
Blk_Mem_tb reports), `workloads` (BASELINE.json's synthetic stream batches), `sharding`
(one-process-per-GPU stream partitioning).  It never computes a match on the CPU.

The directory name has a hyphen, so import it with
    rx = importlib.import_module("regex-fpga_amd")
"""
from . import host, sharding, testbench, workloads  # noqa: F401
from .host import (MODE_FULL, MODE_TB_COMPAT, KERNEL_AUTO, KERNEL_CSR_WAVE, KERNEL_SYM_WAVE,  # noqa: F401
                   KERNEL_SYM_GROUP, KERNEL_SYM_PACK, KERNEL_DFA, KERNEL_SYM_REG, Nfa, Plan, RxError, load_mem, match, match_sharded, lib_path)
