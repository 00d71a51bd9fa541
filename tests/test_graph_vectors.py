"""Graph construction (record filtering + the merge rules of Graph::add_genomic_region / VarRecord::merge*) against the
known answers of the reference's own tests: test/graph/test_graph.cpp and test_haplotypes.cpp, extracted as data into
tests/golden/graph_cases.json by tests/golden/make_graph_vectors.py."""
import json
import os

import numpy as np
import pytest

from graphtyper_amd import lib as gtx
from oracle_lib import Oracle

CASES = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "graph_cases.json")))


def records_of(case):
    out = []
    for r in case["records"]:
        info = []
        if r["ref_events"]:
            info.append("RE=" + ",".join(map(str, r["ref_events"])))
        if r["ref_anti"]:
            info.append("RA=" + ",".join(map(str, r["ref_anti"])))
        for i, a in enumerate(r["alts"]):
            if a["events"]:
                info.append("E%d=%s" % (i, ",".join(map(str, a["events"]))))
            if a["anti_events"]:
                info.append("A%d=%s" % (i, ",".join(map(str, a["anti_events"]))))
        out.append((r["pos"], r["ref"], [a["seq"] for a in r["alts"]], ";".join(info) or None))
    return out


def node_tables(g):
    dna = g["dna"].tobytes().decode()
    ref_dna, var_dna = [], []
    off = 0
    for n in g["ref_len"]:
        ref_dna.append(dna[off:off + int(n)])
        off += int(n)
    for n in g["var_len"]:
        var_dna.append(dna[off:off + int(n)])
        off += int(n)
    return ref_dna, var_dna


def check_expectations(case, g):
    e = case["expect"]
    ref_dna, var_dna = node_tables(g)
    if "n_ref" in e and e["n_ref"] is not None:
        assert len(ref_dna) == e["n_ref"]
    if "n_var" in e and e["n_var"] is not None:
        assert len(var_dna) == e["n_var"]
    for i, s in e["ref_dna"].items():
        assert ref_dna[int(i)] == s, ("ref", i)
    for i, s in e["var_dna"].items():
        assert var_dna[int(i)] == s, ("var", i)
    for i, o in e["ref_order"].items():
        assert int(g["ref_order"][int(i)]) == o
    for i, o in e["var_order"].items():
        assert int(g["var_order"][int(i)]) == o
    for s in e["contains"]:
        assert s in var_dna, s
    for i, d in e["out_degree"].items():
        assert int(g["ref_nvar"][int(i)]) == d
    for k, v in e["var_index"].items():
        r, j = (int(x) for x in k.split(","))
        assert int(g["ref_first_var"][r]) + j == v
    for i, r in e["out_ref"].items():
        assert int(g["var_out_ref"][int(i)]) == r
    if "n_haplotypes" in e:
        assert len(ref_dna) - 1 == e["n_haplotypes"]  # one haplotype per variant site (graph.cpp:680-704)
    if "hap0_num" in e:
        assert int(g["ref_nvar"][0]) == e["hap0_num"]


@pytest.mark.parametrize("case", CASES, ids=[c["name"][:60] for c in CASES])
def test_oracle_graph_matches_reference_tests(case):
    o = Oracle(case["reference"], records_of(case), region_begin=case["region_begin"], add_all_variants=case["add_all_variants"],
               extend_prefix=case["extend_prefix"])
    if "all_ref" in case["expect"]:
        assert o.all_ref() == case["expect"]["all_ref"]
    check_expectations(case, o.graph())


def test_variant_overlapping_an_n():  # test/graph/test_graph.cpp:1436-1519 (three graphs in one TEST_CASE)
    ref = "GCTGCGGCGGGCGTCGCGGCCGCCCCCGGGGAGCCCGGCGGGCGCCGGCGCGNCCCCCCCCCCACCCCACGTCTCGTCGCGCGCGC"
    g = Oracle(ref, [(51, "GN", ["GA"], None)], add_all_variants=True).graph()
    assert node_tables(g) == ([ref], [])
    g = Oracle(ref, [(51, "G", ["GN", "GA"], None)], add_all_variants=True).graph()
    r, v = node_tables(g)
    assert len(r) == 2 and v == ["G", "GA"]
    g = Oracle(ref, [(51, "G", ["GN", "GNN"], None)], add_all_variants=True).graph()
    assert node_tables(g) == ([ref], [])


@pytest.mark.parametrize("case", CASES, ids=[c["name"][:60] for c in CASES])
def test_product_graph_matches_reference_tests(case):
    gtx.build()
    g = gtx.graph_from_records(case["reference"], records_of(case), region_begin=case["region_begin"],
                               add_all_variants=case["add_all_variants"], extend_prefix=case["extend_prefix"])
    check_expectations(case, g)


def _random_records(rng, ref, n_sites, with_events):
    """clusters of SNPs / insertions / deletions, some overlapping, some adjacent, some far apart"""
    recs = []
    pos = int(rng.integers(3, 12))
    ev = 1
    while len(recs) < n_sites and pos + 12 < len(ref):
        kind = int(rng.integers(0, 4))
        if kind == 0:
            r = ref[pos]
            alts = sorted({"ACGT"[(("ACGT".index(r)) + int(k)) % 4] for k in rng.integers(1, 4, size=int(rng.integers(1, 3)))})
        elif kind == 1:
            r = ref[pos]
            alts = [r + "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=int(rng.integers(1, 5))))]
        elif kind == 2:
            d = int(rng.integers(1, 7))
            r = ref[pos:pos + d + 1]
            alts = [ref[pos]]
        else:
            r = ref[pos:pos + 2]
            alts = sorted({ref[pos] + "ACGT"[int(rng.integers(4))] for _ in range(2)} - {r})
            if not alts:
                alts = [ref[pos]]
        info = None
        if with_events and len(alts) == 1 and rng.random() < 0.5:
            info = "GT_ID=%d" % ev
            if ev > 1 and rng.random() < 0.5:
                info += ";GT_ANTI_HAPLOTYPE=%d" % int(rng.integers(1, ev))
            ev += 1
        recs.append((pos, r, alts, info))
        pos += int(rng.choice([0, 1, 1, 2, 3, 5, 9, 14, 40]))
        if recs and pos < recs[-1][0]:
            pos = recs[-1][0]
    return recs


@pytest.mark.parametrize("add_all", [False, True])
@pytest.mark.parametrize("with_events", [False, True])
def test_product_graph_equals_oracle_on_random_records(add_all, with_events):
    gtx.build()
    rng = np.random.default_rng(17 + 2 * add_all + with_events)
    n_checked = 0
    for trial in range(150):
        ref = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=int(rng.integers(80, 400))))
        recs = _random_records(rng, ref, int(rng.integers(1, 14)), with_events)
        begin = int(rng.integers(0, 3)) * 1000
        recs = [(p + begin, r, a, i) for p, r, a, i in recs]
        try:
            og = Oracle(ref, recs, region_begin=begin, add_all_variants=add_all, extend_prefix=True).graph()
        except RuntimeError:
            continue  # e.g. duplicated alts after prefix extension: the reference aborts on those
        g = gtx.graph_from_records(ref, recs, region_begin=begin, add_all_variants=add_all, extend_prefix=True)
        for k in ("ref_order", "ref_len", "ref_nvar", "var_order", "var_len", "var_out_ref"):
            assert np.array_equal(g[k], og[k]), (trial, k)
        assert node_tables(g) == node_tables(og), trial
        # events per variant node
        oe = og["events"]
        pos_ = 0
        for v in range(len(g["var_order"])):
            ne, na = int(oe[pos_]), int(oe[pos_ + 1])
            want_e = sorted(int(x) for x in oe[pos_ + 2:pos_ + 2 + ne])
            want_a = sorted(int(x) for x in oe[pos_ + 2 + ne:pos_ + 2 + ne + na])
            pos_ += 2 + ne + na
            o0, o1, o2 = (int(x) for x in g["event_off"][2 * v:2 * v + 3])
            assert [int(x) for x in g["event_val"][o0:o1]] == want_e, (trial, v)
            assert [int(x) for x in g["event_val"][o1:o2]] == want_a, (trial, v)
        n_checked += 1
    assert n_checked > 100


# test/graph/test_graph.cpp:1436-1519 "Variant overlapping a N on the reference genome": three graphs in one test case
# (the extractor takes one graph per case, so these are written out here)
N_REFERENCE = "GCTGCGGCGGGCGTCGCGGCCGCCCCCGGGGAGCCCGGCGGGCGCCGGCGCGNCCCCCCCCCCACCCCACGTCTCGTCGCGCGCGC"


@pytest.mark.parametrize("records,n_ref,n_var,var_dna", [
    ([(51, "GN", ["GA"], None)], 1, 0, []),               # the reference allele has an N: nothing is added
    ([(51, "G", ["GN", "GA"], None)], 2, 2, ["G", "GA"]),  # an alternative allele has an N: that allele is dropped
    ([(51, "G", ["GN", "GNN"], None)], 1, 0, []),         # every alternative allele has an N: the variant is removed
])
@pytest.mark.parametrize("builder", ["oracle", "product"])
def test_variant_overlapping_an_n(records, n_ref, n_var, var_dna, builder):
    assert len(N_REFERENCE) == 86
    if builder == "oracle":
        g = Oracle(N_REFERENCE, records, add_all_variants=True).graph()
    else:
        g = gtx.graph_from_records(N_REFERENCE, records, add_all_variants=True)
    got_ref, got_var = node_tables(g)
    assert len(got_ref) == n_ref and len(got_var) == n_var
    assert got_var == var_dna
    if n_ref == 1:
        assert got_ref[0] == N_REFERENCE


def test_merged_nodes_carry_the_union_of_events():
    """test/graph/test_graph.cpp:2330-2431: three adjacent SNPs with parity events merge (add_all_variants) into one site;
    the reference allele CAG has events {-1,-2,-3} and anti-events {2,3}, the alternative TGA events {1,2,3} and
    anti-events {-2,-3} -- in the oracle's graph and in the product's event tables"""
    case = [c for c in CASES if c["name"].startswith("parity events test case 2")][0]
    recs = records_of(case)
    want = [({-1, -2, -3}, {2, 3}), ({1, 2, 3}, {-2, -3})]
    og = Oracle(case["reference"], recs, add_all_variants=True, extend_prefix=case["extend_prefix"]).graph()
    ev, at = [int(x) for x in og["events"]], 0
    for events, anti in want:  # per var node: n_events, n_anti, then the values
        ne, na = ev[at], ev[at + 1]
        assert set(ev[at + 2:at + 2 + ne]) == events and set(ev[at + 2 + ne:at + 2 + ne + na]) == anti
        at += 2 + ne + na
    assert at == len(ev)
    g = gtx.graph_from_records(case["reference"], recs, add_all_variants=True, extend_prefix=case["extend_prefix"])
    _, var_dna = node_tables(g)
    assert var_dna == ["CAG", "TGA"]
    off, val = [int(x) for x in g["event_off"]], [int(x) for x in g["event_val"]]
    for v, (events, anti) in enumerate(want):  # slots 2v = events, 2v+1 = anti-events of var node v
        assert set(val[off[2 * v]:off[2 * v + 1]]) == events
        assert set(val[off[2 * v + 1]:off[2 * v + 2]]) == anti


def test_graph_build_validates_its_records():
    """records out of order, or running past the reference they came with, are an error -- not a graph with lost sites or
    empty nodes"""
    ref = "ACGT" * 50
    with pytest.raises(gtx.GtxError, match="not sorted"):
        gtx.graph_from_records(ref, [(50, "G", ["T"], None), (10, "G", ["A"], None)])
    with pytest.raises(gtx.GtxError, match="leaves the reference"):
        gtx.graph_from_records(ref, [(198, "GTAC", ["G"], None)])
    with pytest.raises(gtx.GtxError, match="leaves the reference"):
        gtx.graph_from_records(ref, [(500, "A", ["C"], None)])
    g = gtx.graph_from_records(ref, [(10, "G", ["A"], None), (50, "G", ["T"], None), (196, "ACGT", ["A"], None)])
    assert len(g["ref_order"]) == 4
