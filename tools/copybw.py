#!/usr/bin/env python3
"""PCIe copy rates on this box: pageable / registered (hipHostRegister) / pinned (hipHostMalloc) host memory, both directions,
alone and both at once."""
import time
import numpy as np
import torch
n = 64 << 20
dev = torch.device("cuda", 0)
d = torch.empty(n, dtype=torch.uint8, device=dev)
d2 = torch.empty(n, dtype=torch.uint8, device=dev)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
page = torch.from_numpy(np.random.default_rng(0).integers(0, 255, n, dtype=np.uint8))
page2 = torch.empty(n, dtype=torch.uint8)
reg = torch.from_numpy(np.random.default_rng(1).integers(0, 255, n, dtype=np.uint8))
reg2 = torch.empty(n, dtype=torch.uint8)
rt = torch.cuda.cudart()
for x in (reg, reg2):
    r = rt.cudaHostRegister(x.data_ptr(), x.numel(), 0)
    print("hostRegister ->", r)
pin = page.pin_memory(); pin2 = torch.empty(n, dtype=torch.uint8).pin_memory()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for name, h_in, h_out in (("pageable", page, page2), ("registered", reg, reg2), ("pinned", pin, pin2)):
    h2d = t(lambda: d.copy_(h_in, non_blocking=True))
    d2h = t(lambda: h_out.copy_(d, non_blocking=True))
    def both():
        with torch.cuda.stream(s1): d.copy_(h_in, non_blocking=True)
        with torch.cuda.stream(s2): h_out.copy_(d2, non_blocking=True)
    bi = t(both)
    print(f"{name:10s} H2D {n/h2d/1e9:6.1f} GB/s  D2H {n/d2h/1e9:6.1f} GB/s  both at once {2*n/bi/1e9:6.1f} GB/s total ({bi*1e3:.2f} ms)", flush=True)
# does a copy overlap with a kernel on another stream?
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
def mm():
    with torch.cuda.stream(s1):
        for _ in range(4): torch.mm(a, a)
def cp():
    with torch.cuda.stream(s2): pin2.copy_(d2, non_blocking=True)
tm, tc = t(mm), t(cp)
def both2():
    mm(); cp()
tb = t(both2)
print(f"kernel alone {tm*1e3:.2f} ms, D2H alone {tc*1e3:.2f} ms, both on two streams {tb*1e3:.2f} ms", flush=True)
def cp_h2d():
    with torch.cuda.stream(s2): d.copy_(pin, non_blocking=True)
def both3():
    mm(); cp_h2d()
print(f"kernel + H2D on two streams {t(both3)*1e3:.2f} ms (H2D alone {t(cp_h2d)*1e3:.2f})", flush=True)
