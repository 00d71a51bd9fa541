"""gtx_graph_from_files against the known answers of the reference's constructor tests
(test/graph/test_constructor.cpp, extracted by tests/golden/make_constructor_vectors.py) on the reference's own
index_test.fa / index_test.vcf (committed verbatim under tests/golden)."""
import gzip
import json
import os
import shutil

import numpy as np
import pytest

from graphtyper_amd import lib as gtx

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
CASES = json.load(open(os.path.join(GOLDEN, "constructor_cases.json")))


@pytest.fixture(scope="module", autouse=True)
def _built():
    gtx.build()


def check_case(case, g):
    dna = g["dna"].tobytes().decode()
    seq = {"ref": [dna[o:o + n] for o, n in zip(g["ref_dna_off"], g["ref_len"])],
           "var": [dna[o:o + n] for o, n in zip(g["var_dna_off"], g["var_len"])]}
    order = {"ref": g["ref_order"], "var": g["var_order"]}
    eo, ev = g["event_off"], g["event_val"]
    sets = {"events": lambda v: set(ev[eo[2 * v]:eo[2 * v + 1]].tolist()),
            "anti_events": lambda v: set(ev[eo[2 * v + 1]:eo[2 * v + 2]].tolist())}
    ctx = gtx.Context(g, device=-1, is_sv_graph=case["is_sv_graph"])
    ref_reach, actual = ctx.special_positions()
    special = {"ref_reach_poses": ref_reach, "actual_poses": actual}
    for c in case["checks"]:
        kind = c[0]
        if kind == "count":
            assert len(order[c[1]]) == c[2], c
        elif kind == "out_degree":
            got = int(g["ref_nvar"][c[2]]) if c[1] == "ref" else 1  # a var node has one successor, the next ref node
            assert got == c[3], c
        elif kind == "var_index":
            assert int(g["ref_first_var"][c[1]]) + c[2] == c[3], c
        elif kind == "out_ref":
            assert int(g["var_out_ref"][c[1]]) == c[2], c
        elif kind == "order":
            assert int(order[c[1]][c[2]]) == c[3], c
        elif kind == "dna":
            assert seq[c[1]][c[2]] == c[3], c
        elif kind == "set_size":
            assert len(sets[c[2]](c[1])) == c[3], c
        elif kind == "set_count":
            assert (c[3] in sets[c[2]](c[1])) == bool(c[4]), c
        elif kind == "n_special":
            assert len(actual) == c[1], c
        elif kind == "special":
            assert int(special[c[1]][c[2]]) == c[3], c
        elif kind == "n_special_keys":
            assert len(set(ref_reach.tolist())) == c[1], c
        elif kind == "special_key_count":
            assert (c[1] in set(ref_reach.tolist())) == bool(c[2]), c
        else:
            raise AssertionError("unknown check %r" % (c,))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_reference_constructor_case(case):
    g, _ = gtx.graph_from_files(os.path.join(GOLDEN, case["fasta"]), os.path.join(GOLDEN, case["vcf"]), case["region"],
                                add_all_variants=case["add_all_variants"], is_sv_graph=case["is_sv_graph"])
    check_case(case, g)


def test_gzip_vcf_and_unindexed_fasta_give_the_same_graphs(tmp_path):
    """the reference reads index_test.vcf.gz through tabix and the FASTA through its .fai; the library also takes a
    gzip VCF (multi-member, like bgzip writes) and a FASTA without index (scan)"""
    fa = tmp_path / "t.fa"
    shutil.copy(os.path.join(GOLDEN, "index_test.fa"), fa)  # no .fai next to it
    vz = tmp_path / "t.vcf.gz"
    lines = open(os.path.join(GOLDEN, "index_test.vcf"), "rb").read().splitlines(keepends=True)
    with open(vz, "wb") as f:  # two gzip members
        f.write(gzip.compress(b"".join(lines[:len(lines) // 2])))
        f.write(gzip.compress(b"".join(lines[len(lines) // 2:])))
    for case in CASES:
        a, sa = gtx.graph_from_files(os.path.join(GOLDEN, case["fasta"]), os.path.join(GOLDEN, case["vcf"]), case["region"],
                                     add_all_variants=case["add_all_variants"], is_sv_graph=case["is_sv_graph"])
        b, sb = gtx.graph_from_files(fa, vz, case["region"], add_all_variants=case["add_all_variants"],
                                     is_sv_graph=case["is_sv_graph"])
        assert sa == sb
        assert all(np.array_equal(a[k], b[k]) for k in a)


def test_sv_alleles_and_bad_input_are_refused_loudly():
    fa, vcf = os.path.join(GOLDEN, "index_test.fa"), os.path.join(GOLDEN, "index_test.vcf")
    g, _ = gtx.graph_from_files(fa, vcf, "chr6", is_sv_graph=True)  # <DUP>, <INV>: two breakpoint records each
    assert g["dna"].tobytes().count(b"<SV:") == 4 and len(g["ref_order"]) == 5
    g, _ = gtx.graph_from_files(fa, vcf, "chr7", is_sv_graph=True)  # <INS:ME:ALU> is skipped like the reference skips it
    assert len(g["var_order"]) == 0
    with pytest.raises(gtx.GtxError, match="non-SV graph"):
        gtx.graph_from_files(fa, vcf, "chr6", is_sv_graph=False)
    with pytest.raises(gtx.GtxError, match="not found"):
        gtx.graph_from_files(fa, vcf, "chrZ")
    with pytest.raises(gtx.GtxError, match="cannot open"):
        gtx.graph_from_files(fa + ".missing", vcf, "chr1")
    g, span = gtx.graph_from_files(fa, None, "chr1:11-20")  # reference only
    assert span == (10, 20) and len(g["var_order"]) == 0 and g["dna"].tobytes() == b"AGGTTTCCCC"


def test_sv_deletion_graph_equals_the_oracles():
    """chr5 of the fixture: `<DEL>` with SVSIZE=70.  gtx_graph_from_files synthesises the alleles (add_sv_deletion,
    constructor.cpp:478-514); graph and index have to equal the oracle's, which is pinned on the reference's known answers
    for this contig (tests/test_oracle_pinned.py::test_index_chr5 <- test/index/test_index.cpp:246-312)"""
    from fixtures import sv_contig
    from oracle_lib import Oracle
    fa, vcf = os.path.join(GOLDEN, "index_test.fa"), os.path.join(GOLDEN, "index_test.vcf")
    g, span = gtx.graph_from_files(fa, vcf, "chr5", is_sv_graph=True)
    assert span == (0, 280)
    ref, recs = sv_contig("chr5")
    o = Oracle(ref, recs, is_sv_graph=True)
    og = o.graph()
    for k in ("ref_order", "ref_len", "ref_nvar", "var_order", "var_len", "var_out_ref"):
        assert np.array_equal(g[k], og[k]), k
    assert g["dna"].tobytes() == og["dna"].tobytes()
    assert g["dna"].tobytes().decode().endswith("<SV:0000000>" + "C" * 70 + "G" * 70 + "T" * 70) or b"<SV:0000000>" in g["dna"].tobytes()
    c = gtx.Context(g, device=-1, is_sv_graph=True)
    rr, ap = c.special_positions()
    assert np.array_equal(rr, og["ref_reach_poses"]) and np.array_equal(ap, og["actual_poses"])
    k1, c1, l1 = o.index_dump()
    k2, c2, l2 = c.index_dump()
    assert np.array_equal(k1, k2) and np.array_equal(c1, c2) and np.array_equal(l1, l2)


def test_damaged_input_files_are_refused_not_crashed_on(tmp_path):
    """a few seeds of tests/fuzz_files.py: damaged VCF lines and fields, FASTA index and FASTA bytes, truncated files end in
    GTX_ERR_* (or in the graph of what is still readable), never in a crash of the calling process"""
    import fuzz_files
    assert fuzz_files.run(0, 20, tmp=str(tmp_path)) == 0
