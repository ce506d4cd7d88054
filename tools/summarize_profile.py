#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into the files kept under profiles/<name>/:
kernel_stats.csv (rocprofv3 --kernel-trace --stats) and pmc_summary.json (per-launch means of every --pmc
counter for the dominant kernel); optionally refresh profiles/traffic.json, which bench.py reads for
roofline.traffic.   usage: summarize_profile.py gpurun_out/prof_<tag> profiles/<name> [traffic-key]"""
import collections
import csv
import glob
import json
import os
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
    match = [r for r in rows if "rx_" in r["Name"]]
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
    if key and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        tj_path = os.path.join(os.path.dirname(os.path.abspath(dst)), "traffic.json")
        tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
        fetch = c["FETCH_SIZE"]["mean_per_launch"] * 1024.0   # counters are in KB
        write = c["WRITE_SIZE"]["mean_per_launch"] * 1024.0
        tj[key] = {"hbm_bytes_per_launch": round(fetch + write), "fetch_bytes": round(fetch), "write_bytes": round(write),
                   "source": f"{dst}/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, kernel {short} "
                             f"only; memset and resume kernels excluded)",
                   "note": "FETCH_SIZE/WRITE_SIZE are in KB. The gfx950 x2 correction of MI355X_MICROARCH.md applies to 16 B/lane "
                           "streaming reads; this kernel reads 16 B/lane only for the input windows and gathers dwords otherwise "
                           "(uncalibrated per the guide), so the raw counter is reported. Upper bound with x2 on the read side: "
                           f"{round(2 * fetch + write)}"}
        json.dump(tj, open(tj_path, "w"), indent=1)


if __name__ == "__main__":
    main()
