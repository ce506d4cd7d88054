import importlib, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
rx = importlib.import_module("regex-fpga_amd"); wl = rx.workloads
import torch
snort = rx.Nfa.load_coe(wl.SNORT_COE)
lo, hi = rx.load_mem(wl.TRACES[("snort_16", "lo")]), rx.load_mem(wl.TRACES[("snort_16", "hi")])
for ns in (65536, 131072, 262144, 524288):
    rows = wl.trace_windows(lo, hi, ns, 1024)
    d = torch.from_numpy(rows).cuda()
    for kern, name in ((rx.KERNEL_DFA, "dfa"), (rx.KERNEL_SYM_PACK, "pack16")):
        plan = rx.Plan(snort, ns, 1024, kernel=kern, device=0, events_cap=1 << 23, want_final=True)
        plan.set_device_input(d.data_ptr(), ns, 1024, 1024, keepalive=d)
        for i in range(3): plan.launch()
        plan.kernel_times()
        for i in range(5): plan.launch()
        n, s, mn, mx = plan.kernel_times()
        print(f"{ns} streams {name}: {s/n:.3f} ms -> {8*ns*1024/(s/n)/1e6:.1f} Gbit/s", snort.dfa_info(0) if kern == rx.KERNEL_DFA else "", flush=True)
        plan.close()
    del d
