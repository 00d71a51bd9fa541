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


def test_coverage_model_hand_worked_duplications_sizes_and_points():
    """the rest of make_call_based_on_coverage by hand: <DUP> / <INV> (coverage from the medians' mean and difference,
    sample_call.cpp:315-338), the sizes at which the likelihoods are scaled (<= 100, > 1000), the 190 000-base limit behind which no
    point is taken after the SV, and WHICH positions the points are (20 bases off the ends, every 20 bases outside: 51 in front, 50
    behind; 101 inside at i x (size - 40) / 102)"""
    flat = lambda inside, outside, size, n=20000: (lambda d: (d.__setitem__(slice(5000, 5000 + size), inside), d)[1])(np.full(n, outside, np.uint16))
    # no duplication: medians equal -> the mean 30 for the reference, 0; 0 / 90 / 360, x 3/2 for 2 000 bases -> 0, 135, 540 (written 255)
    assert coverage_call(sv_line("DUP", 5001, 2000), flat(30, 30, 2000)) == [30, 0, 0, 135, 255]
    # heterozygous: inside 45, outside 30: mean 37.5, difference 15 = half of the outside: (1 - 0.5) x 37.5 = 18.75 -> 19, and 37.5 - 19 -> 18;
    # 216 / 111 / 228 -> 105, 0, 117 -> x 3/2 (integer) -> 157, 0, 175
    assert coverage_call(sv_line("DUP", 5001, 2000), flat(45, 30, 2000)) == [19, 18, 157, 0, 175]
    assert coverage_call(sv_line("INV", 5001, 2000), flat(45, 30, 2000)) == [19, 18, 157, 0, 175]
    # homozygous: inside 60 = twice the outside: the fraction is 1 -> 0 and 45; 540 / 135 / 0 -> 810 (255), 202, 0
    assert coverage_call(sv_line("DUP", 5001, 2000), flat(60, 30, 2000)) == [0, 45, 255, 202, 0]
    # fewer reads inside than outside: the mean for the reference, 0
    assert coverage_call(sv_line("DUP", 5001, 2000), flat(20, 30, 2000)) == [25, 0, 0, 112, 255]
    # one read more inside than outside (11 and 10): mean 10.5, a tenth of the outside -> 0.9 x 10.5 = 9.45 -> 9 and 10.5 - 9 -> 1;
    # 12 / 30 / 108 -> 0, 18, 96 -> x 3/2
    assert coverage_call(sv_line("DUP", 5001, 2000), flat(11, 10, 2000)) == [9, 1, 0, 27, 144]
    # no read beside the duplication and 8 inside: the reference divides by the outside's median here; what its machines make of that
    # is 0 and the mean (4): 48 / 12 / 0 -> x 3/2
    assert coverage_call(sv_line("DUP", 5001, 2000), flat(8, 0, 2000)) == [0, 4, 72, 18, 0]
    assert coverage_call(sv_line("DUP", 5001, 2000), flat(8, 1, 2000)) == [0, 4, 72, 18, 0]   # (one read outside: (1 - 7 / 1) x 4.5 is negative -> 0, and 4.5 -> 4)
    # the scaling by size: x 2/3 up to 100 bases, nothing up to 1 000, x 3/2 above (a heterozygous deletion: 90, 0, 90 unscaled)
    assert coverage_call(sv_line("DEL", 5001, 100), flat(15, 30, 100)) == [15, 15, 60, 0, 60]
    assert coverage_call(sv_line("DEL", 5001, 101), flat(15, 30, 101)) == [15, 15, 90, 0, 90]
    assert coverage_call(sv_line("DEL", 5001, 1001), flat(15, 30, 1001)) == [15, 15, 135, 0, 135]
    # 190 000 bases and more: the end is taken 190 000 behind the begin and NO point is taken after the SV.  In front: the 26 points
    # next to the SV see 30 reads, the 25 beyond them 10; behind the SV 10.  With the 50 points behind, 75 of 101 are 10 -> outside 10,
    # AD 5,5, 60 / 30 / 60 -> 30, 0, 30 -> x 3/2; without them the median of the 51 in front is 30 -> AD 5,25, 300 / 90 / 60 -> 240, 30, 0 -> x 3/2
    big = np.full(400000, 10, np.uint16)
    big[5000 - 20 * 26 - 1:5000] = 30
    for size, want in ((189999, [5, 5, 45, 0, 45]), (190000, [5, 25, 255, 45, 0]), (250000, [5, 25, 255, 45, 0])):
        d = big.copy()
        d[5000:5000 + size] = 5
        d[5000 + size:] = 10
        assert coverage_call(sv_line("DEL", 5001, size), d) == want, size
    # the points outside: 20, 40, ... bases off the ends.  51 in front of which only the FIRST (begin - 20) sees reads, 50 behind which
    # all do: 51 of 101 values are 30 -> the median is 30 (AD 0,30); a point at begin - 21 instead, or one point more on either side, and it is 0
    d = np.zeros(20000, np.uint16)
    d[5001 - 20 - 1:5000] = 30          # positions begin - 20 .. begin - 1
    d[6000:7001] = 30                   # the end is begin + size = 6001: the 50 points behind it are 6021 .. 7001 (indices 6020 .. 7000)
    assert coverage_call(sv_line("DEL", 5001, 1000), d)[:2] == [0, 30]
    d2 = d.copy()
    d2[5001 - 20 - 1] = 0               # nobody at begin - 20: 50 of 101
    assert coverage_call(sv_line("DEL", 5001, 1000), d2)[:2] == [0, 0]
    # the points inside: begin + 20 + i x (size - 40) / 102.  size 1 060: begin + 20 + 10 i, i = 1 .. 101; the depth grows by one every
    # ten bases (position begin + k -> k // 10), so point i sees 2 + i reads and the median is point 51's: 53
    d = np.full(20000, 200, np.uint16)
    d[5000:5000 + 1060] = np.arange(1060) // 10
    assert coverage_call(sv_line("DEL", 5001, 1060), d)[:2] == [53, 147]


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
