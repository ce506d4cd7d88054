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
    assert L.rx_abi_version() == 3


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
                 'sizeof(rx_opts),sizeof(rx_result),sizeof(rx_stats),sizeof(rx_nfa_info),offsetof(rx_result,stats),offsetof(rx_opts,k_base));'
                 'printf("%zu\\n", offsetof(rx_opts,flags));return 0;}\n')
    exe = tmp_path / "s"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    h = rx.host
    assert got == [C.sizeof(h._Opts), C.sizeof(h._Result), C.sizeof(h._Stats), C.sizeof(h._Info),
                   h._Result.stats.offset, h._Opts.k_base.offset, h._Opts.flags.offset]


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


def test_pass_index_range_is_checked_before_any_device_work(rx):
    """rx_event.k is 32 bits: k_base + passes of the batch beyond 2^32 is RX_EINVAL (-1), never a silent wrap —
    checked before the device is touched, so it shows here without a GPU."""
    import numpy as np
    nfa = rx.Nfa.from_words(np.array([0, 1, 1, 0x61000001], np.uint32))
    rows = np.zeros((1, 16), np.uint8)
    for k_base in (2**32 - 16, 2**32, 2**40):
        with pytest.raises(rx.RxError) as e:
            rx.match(nfa, rows, k_base=k_base)
        assert e.value.code == -1, k_base
        with pytest.raises(rx.RxError) as e:
            rx.Plan(nfa, 1, 16, k_base=k_base)
        assert e.value.code == -1, k_base
    try:  # the largest base that still fits: passes 0..16 -> k up to 2^32 - 1
        rx.match(nfa, rows, k_base=2**32 - 17)
    except rx.RxError as e:
        assert e.code == -6  # no device here; on a GPU box it runs


def test_options_struct_of_an_older_caller(rx):
    """A caller built against ABI 1 passes an rx_opts that ends before `flags` (struct_size = 40): the library must
    not read past it."""
    import numpy as np
    h = rx.host
    nfa = rx.Nfa.from_words(np.array([0, 1, 1, 0x61000001], np.uint32))
    o = h._Opts()
    o.struct_size = h._Opts.flags.offset
    o.device, o.mode, o.kernel = -1, 5, 0        # invalid mode: the options WERE read
    o.flags = 0xFFFFFFFF                         # garbage behind the caller's struct
    p = C.c_void_p()
    assert h.lib().rx_plan_create(nfa._h, C.byref(o), 1, 4, 0, 0, 0, 0, C.byref(p)) == -1


def test_list_form_of_final_sets_expands_to_rows(rx):
    """host.expand_final: (offset, count, states) per stream -> the bitmask rows rx_result.final_active would hold."""
    import numpy as np
    res = dict(final_off=np.array([0, 2, 2, 5], np.uint32), final_cnt=np.array([2, 0, 3, 1], np.uint32),
               final_states=np.array([1, 64, 0, 63, 130, 7], np.uint32))
    rows = rx.host.expand_final(res, 3)
    want = np.zeros((4, 3), np.uint64)
    want[0, 0] = 1 << 1
    want[0, 1] = 1 << 0
    want[2, 0] = (1 << 0) | (1 << 63)
    want[2, 2] = 1 << 2
    want[3, 0] = 1 << 7
    assert rows.dtype == np.uint64 and np.array_equal(rows, want)
