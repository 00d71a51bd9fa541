"""Graph construction (record filtering + the merge rules of Graph::add_genomic_region / VarRecord::merge*) against the
known answers of the reference's own tests: test/graph/test_graph.cpp and test_haplotypes.cpp, extracted as data into
tests/golden/graph_cases.json by tests/golden/make_graph_vectors.py."""
import json
import os

import numpy as np
import pytest

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
