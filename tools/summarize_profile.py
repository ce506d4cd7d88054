#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into the files kept under profiles/<name>/:
kernel_stats.csv (rocprofv3 --kernel-trace --stats) and pmc_summary.json (per-launch means of every --pmc
counter for the dominant kernel); optionally refresh profiles/traffic.json, which bench.py reads for
roofline.traffic / roofline.pmc — keyed "<kernel>[:<variant>]:<workload>:<streams>x<len>" and stamped with the sha256
of the kernel sources, so that bench.py reports it only for exactly the code that was profiled.
usage: summarize_profile.py gpurun_out/prof_<tag> profiles/<name> [traffic-key]"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    key = sys.argv[3] if len(sys.argv) > 3 else None
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv"))
    if not stats:
        sys.exit("no kernel_stats.csv under " + src)
    shutil.copy(stats[0], os.path.join(dst, "kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats[0])))
    # (AUTO's probe runs the pack kernel's statistics build <S, true, ...> a few times: never the kernel that was timed)
    match = [r for r in rows if "rx_" in r["Name"] and not re.search(r"rx_sym_pack_kernel<\d+, true", r["Name"])]
    main_k = max(match, key=lambda r: float(r["TotalDurationNs"]))["Name"]
    short = main_k.replace("void (anonymous namespace)::", "").replace("(RxParams)", "")
    counters = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith(main_k.split("(RxParams)")[0]):
                counters[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"command": "tools/profile.sh (rocprofv3 --kernel-trace --stats, then separate --pmc passes, each: -- python3 "
                      "bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-second-distribution)",
           "kernel": short, "kernel_trace_avg_ns": float(next(r for r in rows if r["Name"] == main_k)["AverageNs"]),
           "counters": {k: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for k, v in sorted(counters.items())}}
    c = out["counters"]
    if "SQ_WAVES" in c and "SQ_INSTS_VALU" in c:
        out["per_wave"] = {k: c[k]["mean_per_launch"] / c["SQ_WAVES"]["mean_per_launch"]
                           for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES") if k in c}
    json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
    print(short, "avg", out["kernel_trace_avg_ns"] / 1e6, "ms;", {k: round(v["mean_per_launch"]) for k, v in c.items()})
    # figures bench.py copies into roofline.pmc (only while the kernel source is the one that was profiled)
    t_s = out["kernel_trace_avg_ns"] * 1e-9
    # shader clock under this load, measured in the kernel (s_memtime / s_memrealtime): 2.37 GHz.  (SQ_WAVE_CYCLES
    # undercounts by ~23 %: do not derive the clock from it.)
    n_simd, clk = 1024, 2.37e9
    pmc = {}
    if "SQ_INSTS_VALU" in c:
        # lower bound: 2.1 cycles per plain VALU instruction; forms with an SGPR operand / result or three sources take 4
        pmc["valu_issue_frac"] = round(c["SQ_INSTS_VALU"]["mean_per_launch"] * 2.1 / (n_simd * clk * t_s), 4)
    if "SQ_INSTS_SALU" in c:
        # one SALU instruction per 4 cycles per SIMD (tools/issue_bench); branches are not in SQ_INSTS_SALU
        pmc["salu_issue_frac"] = round(c["SQ_INSTS_SALU"]["mean_per_launch"] * 4.0 / (n_simd * clk * t_s), 4)
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
        pmc["wave_cycles_waiting_frac"] = round(c["SQ_WAIT_ANY"]["mean_per_launch"] / c["SQ_WAVE_CYCLES"]["mean_per_launch"], 4)
    # (SQ_LDS_BANK_CONFLICT and SQ_LDS_IDX_ACTIVE count LDS-array cycles; SQ_ACTIVE_INST_LDS counts quad-cycles of waves
    # and must not be the denominator — profiles/r02_lds_ab/README.md)
    if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c:
        pmc["lds_conflict_frac_of_lds_array_cycles"] = round(c["SQ_LDS_BANK_CONFLICT"]["mean_per_launch"] / c["SQ_LDS_IDX_ACTIVE"]["mean_per_launch"], 4)
        pmc["lds_array_utilisation"] = round(c["SQ_LDS_IDX_ACTIVE"]["mean_per_launch"] / (256 * clk * t_s), 4)
    if "TCP_TCC_READ_REQ_sum" in c and "SQ_INSTS_VMEM_RD" in c:
        pmc["l2_requests_per_vmem_read_instruction"] = round(c["TCP_TCC_READ_REQ_sum"]["mean_per_launch"] / c["SQ_INSTS_VMEM_RD"]["mean_per_launch"], 3)
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        pmc["l2_hit_rate"] = round(c["TCC_HIT_sum"]["mean_per_launch"] / (c["TCC_HIT_sum"]["mean_per_launch"] + c["TCC_MISS_sum"]["mean_per_launch"]), 4)
    out["derived"] = pmc
    json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
    if key and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        tj_path = os.path.join(os.path.dirname(os.path.abspath(dst)), "traffic.json")
        tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
        fetch = c["FETCH_SIZE"]["mean_per_launch"] * 1024.0   # counters are in KB
        write = c["WRITE_SIZE"]["mean_per_launch"] * 1024.0
        import hashlib
        h = hashlib.sha256()
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for f in ("rx_kernels.hip", "rx_internal.hpp"):
            h.update(open(os.path.join(root, "regex-fpga_amd", "csrc", f), "rb").read())
        tj[key] = {"hbm_bytes_per_launch": round(fetch + write), "fetch_bytes": round(fetch), "write_bytes": round(write),
                   "src_sha16": h.hexdigest()[:16], "kernel_ms": round(out["kernel_trace_avg_ns"] / 1e6, 4), "pmc": pmc,
                   "source": f"{dst}/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, kernel {short} "
                             f"only; memset and resume kernels excluded)",
                   "note": "FETCH_SIZE/WRITE_SIZE are in KB. The gfx950 x2 correction of MI355X_MICROARCH.md applies to 16 B/lane "
                           "streaming reads; this kernel reads 16 B/lane only for the input windows and gathers dwords otherwise "
                           "(uncalibrated per the guide), so the raw counter is reported. Upper bound with x2 on the read side: "
                           f"{round(2 * fetch + write)}"}
        json.dump(tj, open(tj_path, "w"), indent=1)


if __name__ == "__main__":
    main()
