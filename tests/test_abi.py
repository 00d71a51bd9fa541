"""The C-ABI library loads without a GPU and exports every symbol include/gtx.h declares."""
import os
import re

from graphtyper_amd import lib as gtx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header():
    gtx.build()
    L = gtx.lib()
    header = open(os.path.join(ROOT, "include", "gtx.h")).read()
    declared = set(re.findall(r"\b(gtx_[a-z0-9_]+)\s*\(", header))
    assert declared == set(gtx.EXPORTS), declared ^ set(gtx.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def test_strerror():
    L = gtx.lib()
    assert L.gtx_strerror(0) == b"ok"
    assert b"no HIP device" in L.gtx_strerror(2)


def test_binding_declares_argument_types_for_every_entry_point():
    """ctypes passes an undeclared Python int as a 32-bit C int: a device pointer would be truncated"""
    L = gtx.lib()
    missing = [n for n in gtx.EXPORTS if getattr(L, n).argtypes is None and n not in ("gtx_last_error",)]
    assert missing == []


def test_header_is_plain_c(tmp_path):
    """the boundary is a C ABI: include/gtx.h has to compile as C99 (no C++ types in the signatures) and link against
    the library"""
    import subprocess
    src = tmp_path / "use_gtx.c"
    src.write_text('#include "gtx.h"\nint main(void) { gtx_params p; gtx_score_buffers b; (void)p; (void)b; '
                   'return gtx_strerror(0) == 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-c",
                           str(src), "-o", str(tmp_path / "use_gtx.o")])
