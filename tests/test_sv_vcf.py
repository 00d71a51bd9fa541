"""SV post-processing of the calls (oracle/gto_sv.hpp restating src/graph/sv.cpp:117-655 and src/typer/sample_call.cpp:189-385)
on hand-worked cases -- the reference holds no vector for this path, so the numbers below were worked from its text -- and the
product (gtx_vcf_records on an SV graph) against the oracle through the whole pipeline on a small cfg5-like input."""
import ctypes as C

import numpy as np
import pytest

import harness
from graphtyper_amd import lib as gtx
from oracle_lib import lib as oracle_lib


@pytest.fixture(scope="module", autouse=True)
def _built():
    gtx.build()


def sv_line(kind, begin, size, model="BREAKPOINT"):
    cols = [kind, "chrS", begin, size, size, begin + size, 0, -1, -1, -1, -1, model] + ["."] * 8
    return "\t".join(str(c) for c in cols) + "\n"


def coverage_call(table, depth, offset=1):
    L = oracle_lib()
    out = (C.c_uint32 * 5)()
    d = np.ascontiguousarray(depth, np.uint16)
    rc = L.gto_coverage_call(table.encode(), C.c_long(0), d.ctypes.data_as(C.c_void_p), C.c_long(len(d)), C.c_uint32(offset), out)
    if rc != 0:
        L.gto_last_error.restype = C.c_char_p
        raise RuntimeError(L.gto_last_error().decode())
    return list(out)


def test_coverage_model_hand_worked():
    """make_call_based_on_coverage (sample_call.cpp:255-385): medians of the depth at 101 points inside the deletion (20 bp off
    its ends) and at 51 + 50 points every 20 bp on either side; "coverage" = (inside, outside - inside); PL = 12 per read
    against a homozygous genotype, 3 per read against the heterozygous one, lowest subtracted; x 2/3 up to 100 bp, x 3/2 above 1 kb"""
    depth = np.full(20000, 30, np.uint16)
    depth[5000:6000] = 15  # (index = position - offset; offset 1: position 5001 .. 6000)
    # a heterozygous 1 000-bp deletion at 5001: inside 15, outside 30 -> AD 15,15; 180 / 90 / 180 -> 90,0,90; 1000 is not > 1000
    assert coverage_call(sv_line("DEL", 5001, 1000), depth) == [15, 15, 90, 0, 90]
    # homozygous, 2 000 bp: inside 0 -> AD 0,30; 360 / 90 / 0; x 3/2 -> 540 (written 255), 135, 0
    depth2 = np.full(20000, 30, np.uint16)
    depth2[5000:7000] = 0
    assert coverage_call(sv_line("DEL", 5001, 2000), depth2) == [0, 30, 255, 135, 0]
    # no deletion, 80 bp: inside = outside = 30 -> AD 30,0; 0 / 90 / 360 -> x 2/3 -> 0,60,240
    assert coverage_call(sv_line("DEL", 5001, 80), np.full(20000, 30, np.uint16)) == [30, 0, 0, 60, 240]
    # homozygous, 80 bp: AD 0,30; 360 / 90 / 0, x 2/3 each -> 240, 60, 0
    depth4 = np.full(20000, 30, np.uint16)
    depth4[5000:5080] = 0
    assert coverage_call(sv_line("DEL", 5001, 80), depth4) == [0, 30, 240, 60, 0]
    # the points beside the SV are 20 bp apart, 51 in front and 50 behind: with reads only within 400 bp of it most of them see none
    depth5 = np.zeros(20000, np.uint16)
    depth5[4600:6400] = 30
    depth5[5000:6000] = 15
    assert coverage_call(sv_line("DEL", 5001, 1000), depth5)[:2] == [15, 0]
    # more reads inside than outside never gives a negative count
    depth3 = np.full(20000, 10, np.uint16)
    depth3[5000:5500] = 14
    assert coverage_call(sv_line("DEL", 5001, 500), depth3)[:2] == [14, 0]
    # an SV of 40 bases has no point inside: the reference takes the median of an empty vector; here it is refused
    with pytest.raises(RuntimeError):
        coverage_call(sv_line("DEL", 5001, 40), depth)


def test_small_sv_graph_vcf_equals_the_oracle(tmp_path):
    """the product's SV post-processing (gtx_vcf.cpp: sv_graph_records) against the oracle's on a small cfg5-like input, through
    the host emulation of the kernels (the `-m gpu` suite runs the same case through libgtx at cfg5's size)"""
    from test_emu_parity import cfg5_case
    cfg5_case(harness.EmuBackend, tmp_path, n_ref=60000, n_del=8, n_ins=4, n_samples=5, pairs_per_sv=30, background_pairs=200)
    lines = cfg5_case.vcf.decode().split("\n")[1:-1]
    by_pos = {}
    for l in lines:
        f = l.split("\t")
        by_pos.setdefault(f[1], []).append(f)
    # the records of one SV share its position and type: the second and third carry ".0", ".1" behind the ID (vcf.cpp:1240-1275)
    some = [v for v in by_pos.values() if len(v) == 3]
    assert some, "no SV with three records (AGGREGATED, BREAKPOINT*, COVERAGE)"
    for v in some:
        ids = [f[2] for f in v]
        assert ids[1] == ids[0] + ".0" and ids[2] == ids[0] + ".1", ids
        assert all(f[3] == "N" and f[4].startswith("<") and f[8] == "GT:FT:AD:MD:DP:RA:PP:GQ:PL" for f in v)
    # every record names its model in the allele and in INFO, END is behind POS, SV_ID points into the SV table
    for f in lines:
        f = f.split("\t")
        info = dict(kv.split("=", 1) if "=" in kv else (kv, "") for kv in f[7].split(";"))
        assert info["SVMODEL"] in f[4] and int(info["END"]) >= int(f[1]) and "SV_ID" in info and info["SVTYPE"] in ("DEL", "INS")
