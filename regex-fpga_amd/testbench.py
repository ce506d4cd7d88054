"""What the reference's only caller, `Blk_Mem_tb`, does with the matcher — on the GPU path.

Simulation/testbench_BLK_Mem.sv: two traces are $readmemh'ed (:34-35), fed in lock-step as
input_char / input_char_2 (:53-59), every accept pulse increments a 10-bit counter per state and
stream (:21-22, :61-69), and when m == 200000 the non-zero counters are printed, highest state first,
stream 1 then stream 2 (:71-86).  Here the two traces are simply two rows of one rx_match() batch in
RX_MODE_TB_COMPAT (passes 0..N-2).
"""
import numpy as np

from . import host

TB_M_STOP = 200000  # testbench_BLK_Mem.sv:71


def run(nfa, lo, hi, m_stop=TB_M_STOP, kernel=host.KERNEL_AUTO, device=-1, collect_stats=2):
    """-> dict(match_count[2][size] (raw), events, report (str), total_cycles, stats).
    collect_stats=2 also evaluates the FPGA's clock count of the run on the GPU."""
    lo = np.asarray(lo, np.uint8)
    hi = np.asarray(hi, np.uint8)
    if lo.size < m_stop or hi.size < m_stop:
        raise ValueError(f"traces must hold at least m_stop={m_stop} bytes")
    rows = np.stack([lo[:m_stop], hi[:m_stop]])
    r = host.match(nfa, rows, mode=host.MODE_TB_COMPAT, kernel=kernel, device=device, want_match_count=True,
                   events_cap=1 << 20, collect_stats=collect_stats)
    cyc = r["stats"]["tb_cycles"] or None
    r["total_cycles"] = cyc
    # $time at the $display: last posedge at 10*cycles ns, then #1, #1 (byte load) and #20 (:52-73)
    r["report"] = format_report(r["match_count"][0], r["match_count"][1], cyc, 10 * cyc + 22 if cyc else None)
    return r


def format_report(mc1, mc2, total_cycles=None, time_ns=None):
    """The $display lines of testbench_BLK_Mem.sv:75-84.  %d pads to the operand's widest decimal:
    11 columns for the 32-bit `int` index p, 4 for the 10-bit counters (which wrap mod 1024)."""
    lines = []
    for name, mc in (("match_count", mc1), ("match_count_2", mc2)):
        mc = np.asarray(mc, dtype=np.uint64) & np.uint64(1023)  # logic [9:0]
        for p in range(len(mc) - 1, -1, -1):                      # foreach over [size-1:0] descends
            if mc[p] != 0:
                lines.append(f"{name}[{p:11d}] = {int(mc[p]):4d}")
    if total_cycles is not None:
        # `int cycles` is a 32-bit signed SystemVerilog int (:19): snort_16's 2 188 184 738 clocks wrap to
        # -2 106 782 558 in the real testbench's printout, and so do they here
        wrapped = ((int(total_cycles) + 2**31) % 2**32) - 2**31
        lines.append(f"{(time_ns if time_ns is not None else 0):20d}")
        lines.append(f"Total no. cycles: {wrapped:11d}")
    return "\n".join(lines)
