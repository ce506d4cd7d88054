#!/usr/bin/env python3
"""Count the instructions of the largest loop of every kernel in a gfx950 .s file, by issue category.
usage: count_loop_insts.py file.s [name-filter]"""
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\n(_Z\w+):\s*; @\1\n(.*?)\.Lfunc_end\d+:", s, re.S):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    labels, ins = {}, []
    for l in body.split("\n"):
        l = l.strip()
        if not l or l.startswith(";"):
            continue
        lm = re.match(r"(\.LBB\d+_\d+|\d+):", l)
        if lm:
            labels.setdefault(lm.group(1), len(ins))
            continue
        if l.startswith(".") or l.endswith(":"):
            continue
        ins.append(l.split(";")[0].strip())
    best = None
    for idx, l in enumerate(ins):
        bm = re.match(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if bm and bm.group(1) in labels and labels[bm.group(1)] < idx:
            span = idx - labels[bm.group(1)] + 1
            if not best or span > best[0]:
                best = (span, labels[bm.group(1)], idx)
    if not best:
        continue
    span, a, b = best
    cat = {}
    for l in ins[a:b + 1]:
        op = l.split()[0]
        c = ("VALU" if op.startswith("v_") else "LDS" if op.startswith("ds_") else
             "WAIT" if op.startswith(("s_waitcnt", "s_nop")) else "BRANCH" if op.startswith(("s_cbranch", "s_branch")) else
             "VMEM" if op.startswith(("global_", "buffer_", "flat_")) else "SMEM" if op.startswith(("s_load", "s_memtime", "s_memrealtime")) else "SALU")
        cat[c] = cat.get(c, 0) + 1
    print(name, span, dict(sorted(cat.items())))
