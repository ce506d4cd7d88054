import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
DATA = os.path.join(ROOT, "data")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orx():
    """CPU oracle (test infrastructure): oracle/liborx.so via oracle/orx.py."""
    from oracle import orx as m
    m.lib()
    return m


@pytest.fixture(scope="session")
def rx():
    """The product package; loading it requires the in-tree librxmatch.so (built by build())."""
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, g.PKG, "librxmatch.so")):
        g.build()
    m = importlib.import_module(g.PKG)
    m.host.lib()
    return m


@pytest.fixture(scope="session")
def automata(orx):
    """name -> (words, size) decoded by the oracle's .coe parser."""
    out = {}
    for name, f in (("l7", "CSR_BlockMem.coe"), ("snort_16", "CSR_BlockMem_snort_16.coe")):
        W = orx.load_coe(os.path.join(DATA, f))
        out[name] = (W, orx.infer_size(W))
    return out


@pytest.fixture(scope="session")
def traces(orx):
    out = {}
    for name, tag in (("l7", "l-7_filter"), ("snort_16", "snort_16")):
        for lh in ("lo", "hi"):
            out[(name, lh)] = orx.load_mem(os.path.join(DATA, f"input_trace_{lh}_{tag}.mem"))
    return out
