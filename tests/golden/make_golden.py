#!/usr/bin/env python3
"""Regenerates tests/golden/golden.json.

The reference ships NO expected outputs (its testbench has no assertions and no golden file), and
its RTL cannot be simulated in this image, so these vectors are produced by the CPU oracle
(oracle/rx_oracle.c functional model; oracle/rx_cycle.c clock-accurate model for the cycle totals)
from the reference's own input files in data/.  They coincide with the digests the survey computed
with an independent Python model.  Parity is therefore "unpinned by the
reference"; these fixtures pin the oracle against regressions and the GPU path against the oracle.
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orx  # noqa: E402

DATA = os.path.join(ROOT, "data")
N = 200000


def main():
    out = {"_provenance": __doc__.strip(), "files": {}, "tables": {}, "tb_compat": {}, "full": {}, "cycles": {}}
    for f in sorted(os.listdir(DATA)):
        if f.endswith((".coe", ".mem")):
            out["files"][f] = hashlib.sha256(open(os.path.join(DATA, f), "rb").read()).hexdigest()
    for name, coe, tag in (("l7", "CSR_BlockMem.coe", "l-7_filter"), ("snort_16", "CSR_BlockMem_snort_16.coe", "snort_16")):
        W = orx.load_coe(os.path.join(DATA, coe))
        size = orx.infer_size(W)
        out["tables"][name] = dict(n_words=int(W.size), size=size, nnz=int(W[size]),
                                   sha256_words_le=hashlib.sha256(W.astype("<u4").tobytes()).hexdigest())
        tr = {}
        for lh in ("lo", "hi"):
            tr[lh] = orx.load_mem(os.path.join(DATA, f"input_trace_{lh}_{tag}.mem"))
            for mode, key in ((orx.MODE_TB_COMPAT, "tb_compat"), (orx.MODE_FULL, "full")):
                r = orx.match_batch(W, size, tr[lh][:N], mode=mode, nthreads=1, want_match_count=True)
                mc = r["match_count"][0]
                out[key][f"{name}:{lh}"] = dict(
                    n_events=r["n_events"], sum_active=r["stats"]["sum_active"], sum_edges=r["stats"]["sum_edges"],
                    alg_bytes=r["stats"]["alg_bytes"], H_mc=orx.h_match_count(mc), H_ev=orx.h_events(r["events"]),
                    first_events=[[int(e["k"]), int(e["state"])] for e in r["events"][:8]],
                    nonzero_counts={str(i): int(mc[i]) for i in mc.nonzero()[0][:16]},
                    final_active=orx.bits_to_states(r["final_active"][0]))
        c = orx.tb_cycle(W, size, tr["lo"][:N + 1], tr["hi"][:N + 1], N, skip_idle=True)
        out["cycles"][name] = dict(total_cycles=c["total_cycles"], passes=c["passes"],
                                   H_mc=orx.h_match_count(c["match_count"]), H_mc_2=orx.h_match_count(c["match_count_2"]))
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote golden.json")


if __name__ == "__main__":
    main()
