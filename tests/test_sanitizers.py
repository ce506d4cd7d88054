"""AddressSanitizer + UBSan over the host-side C/C++ (parsers, index builder, regex compiler) and both
oracle models.  CPU-only (GPU sanitizers are unavailable on this pool)."""
import os
import subprocess

import pytest

from conftest import DATA, ROOT


def test_host_and_oracle_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "sanitize_main")
    csrc = os.path.join(ROOT, "regex-fpga_amd", "csrc")
    orc = os.path.join(ROOT, "oracle")
    objs = []
    for src in (os.path.join(orc, "rx_oracle.c"), os.path.join(orc, "rx_cycle.c")):
        o = str(tmp_path / (os.path.basename(src) + ".o"))
        subprocess.check_call(["gcc", "-std=gnu11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                               "-pthread", "-c", src, "-o", o])
        objs.append(o)
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-pthread", os.path.join(ROOT, "tests", "native", "sanitize_main.cpp"),
           os.path.join(csrc, "rx_host.cpp"), os.path.join(csrc, "rx_compile.cpp")] + objs + ["-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in (r.stderr or "") and "cannot find" in r.stderr:
        pytest.skip("libasan/libubsan not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    for coe, tag in (("CSR_BlockMem_snort_16.coe", "snort_16"), ("CSR_BlockMem.coe", "l-7_filter")):
        out = subprocess.run([exe, os.path.join(DATA, coe), os.path.join(DATA, f"input_trace_lo_{tag}.mem"),
                              os.path.join(DATA, f"input_trace_hi_{tag}.mem")], capture_output=True, text=True,
                             env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
        assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
        assert "sanitize ok" in out.stdout and "runtime error" not in out.stderr
