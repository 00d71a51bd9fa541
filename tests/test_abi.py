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
