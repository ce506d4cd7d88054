"""The C-ABI boundary: librxmatch.so loads and exports exactly what include/rxmatch.h declares.
No compute calls here (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def declared_functions():
    src = open(os.path.join(ROOT, "include", "rxmatch.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rx_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(rx):
    names = declared_functions()
    assert len(names) >= 20
    L = C.CDLL(rx.lib_path())
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/rxmatch.h but not exported"
    assert sorted(rx.host.ABI_SYMBOLS) == names  # the Python binding covers every entry point
    assert L.rx_abi_version() == 1


def test_header_is_plain_c(tmp_path):
    """The boundary header compiles as C99 (no C++/torch types in any signature)."""
    import subprocess
    c = tmp_path / "t.c"
    c.write_text('#include "rxmatch.h"\nint main(void){ rx_opts o; rx_result r; (void)o; (void)r; return sizeof(rx_event)==12 ? 0 : 1; }\n')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0


def test_struct_layouts_match_binding(rx, tmp_path):
    import subprocess
    c = tmp_path / "s.c"
    c.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rxmatch.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu\\n",'
                 'sizeof(rx_opts),sizeof(rx_result),sizeof(rx_stats),sizeof(rx_nfa_info),offsetof(rx_result,stats),offsetof(rx_opts,k_base));return 0;}\n')
    exe = tmp_path / "s"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    h = rx.host
    assert got == [C.sizeof(h._Opts), C.sizeof(h._Result), C.sizeof(h._Stats), C.sizeof(h._Info),
                   h._Result.stats.offset, h._Opts.k_base.offset]


def test_error_strings(rx):
    L = rx.host.lib()
    for code in range(0, -10, -1):
        assert L.rx_strerror(code)
    assert b"no CPU fallback" in L.rx_strerror(-6)


def test_product_does_not_link_the_oracle(rx):
    """The shipped library must not reference oracle code (no CPU fallback, oracle is test-only)."""
    import subprocess
    syms = subprocess.check_output(["nm", "-D", rx.lib_path()]).decode()
    assert "orx_" not in syms
    ldd = subprocess.check_output(["ldd", rx.lib_path()]).decode()
    assert "liborx" not in ldd and "libamdhip64" in ldd


def test_no_device_fails_loudly(rx):
    """Without a HIP device every compute entry point returns RX_ENODEVICE — never a CPU result."""
    import numpy as np
    try:
        n = rx.host.device_count()
    except rx.RxError:
        n = 0
    if n > 0:
        pytest.skip("a GPU is present")
    nfa = rx.Nfa.from_words(np.array([0, 1, 1, 0x61000001], np.uint32))
    with pytest.raises(rx.RxError) as e:
        rx.match(nfa, np.zeros((1, 4), np.uint8))
    assert e.value.code == -6
    with pytest.raises(rx.RxError) as e:
        rx.Plan(nfa, 1, 4)
    assert e.value.code == -6
