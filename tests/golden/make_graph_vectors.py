#!/usr/bin/env python3
"""Extracts the known-answer vectors of the reference's graph-construction tests (test/graph/test_graph.cpp,
test/graph/test_haplotypes.cpp) into tests/golden/graph_cases.json: inputs (reference string, variant records with
events, region begin, add_all_variants) and the expected node tables the REQUIREs assert.  Only data is extracted --
run in the development container where /root/reference exists:  python tests/golden/make_graph_vectors.py"""
import json
import os
import re
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/test/graph/test_graph.cpp"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph_cases.json")


def strip_comments(t):
    t = re.sub(r"/\*.*?\*/", "", t, flags=re.S)
    return re.sub(r"//[^\n]*", "", t)


def chars(s):
    m = re.search(r'to_vec\("([^"]*)"\)', s)
    return m.group(1) if m else "".join(re.findall(r"'(.)'", s))


ADD_ALL = [False]  # Options::add_all_variants is global state that leaks from one TEST_CASE into the next


def parse_case(name, body):
    m = re.search(r'testdata(?:\[\])?\s*=\s*"([^"]*)"', body)
    has_graph = bool(m) and "add_genomic_region" in body
    cut = body.index("add_genomic_region") if has_graph else len(body)
    for a in re.finditer(r"add_all_variants = (true|false)", body[:cut]):
        ADD_ALL[0] = a.group(1) == "true"
    add_all_here = ADD_ALL[0]
    for a in re.finditer(r"add_all_variants = (true|false)", body[cut:]):
        ADD_ALL[0] = a.group(1) == "true"
    if not has_graph:
        return None
    if body.count("add_genomic_region") > 1:
        return None  # several graphs in one TEST_CASE: restated by hand in tests/test_graph_vectors.py
    case = dict(name=name, reference=m.group(1), add_all_variants=add_all_here, records=[],
                extend_prefix="add_reference_to_record_if_they_have_a_matching_prefix" in body,
                region_begin=0, expect=dict(ref_dna={}, var_dna={}, ref_order={}, var_order={}, contains=[], out_degree={},
                                            var_index={}, out_ref={}))
    r = re.search(r'GenomicRegion\("chr1:(\d+)"\)', body)
    if r:
        case["region_begin"] = int(r.group(1)) - 1
    cur = dict(pos=0, ref="", alts=[], ref_events=[], ref_anti=[])
    rec_part = body[:cut]
    pat = (r"record\.(pos|ref|alts)(?:\[(\d+)\])?"
           r"(?:\.(events|anti_events)(?:\.(?:emplace|insert)\((-?\d+)\)|\s*=\s*\{([^}]*)\}))?"
           r"\s*(?:=\s*([^;]*))?;|records\.push_back\(record\)|record\.clear\(\)")
    for st in re.finditer(pat, rec_part):
        if st.group(0).startswith("records.push_back"):
            case["records"].append(json.loads(json.dumps(cur)))
            continue
        if st.group(0).startswith("record.clear"):
            cur = dict(pos=0, ref="", alts=[], ref_events=[], ref_anti=[])
            continue
        field, idx, ev, evval, evlist, val = st.groups()
        if ev:
            vals = [int(evval)] if evval is not None else [int(x) for x in evlist.replace(" ", "").split(",") if x]
            if field == "alts":
                tgt = cur["alts"][int(idx)]
                key = ev
            else:
                tgt = cur
                key = "ref_events" if ev == "events" else "ref_anti"
            if evval is not None:
                tgt[key].extend(vals)
            else:
                tgt[key] = vals
        elif field == "pos":
            cur["pos"] = int(val)
        elif field == "ref":
            cur["ref"] = chars(val)
            cur["ref_events"], cur["ref_anti"] = [], []
        elif field == "alts":
            if "to_vec" in val:
                cur["alts"] = [dict(seq=a, events=[], anti_events=[]) for a in re.findall(r'to_vec\("([^"]*)"\)', val)]
            else:
                cur["alts"] = [dict(seq=chars(a), events=[], anti_events=[]) for a in re.findall(r"\{((?:'.'(?:,\s*)?)*)\}", val)]
    e = case["expect"]
    for m in re.finditer(r"REQUIRE\((?:graph\.)?(ref|var)_nodes\.size\(\) == (\d+)\)", body):
        e["n_" + m.group(1)] = int(m.group(2))
    for m in re.finditer(r'REQUIRE\((ref|var)_nodes\[(\d+)\]\.get_label\(\)\.dna == gyper::to_vec\("([^"]*)"\)\)', body):
        e[m.group(1) + "_dna"][m.group(2)] = m.group(3)
    for m in re.finditer(r"REQUIRE\((ref|var)_nodes\[(\d+)\]\.get_label\(\)\.order == ([0-9+ ]+)\)", body):
        e[m.group(1) + "_order"][m.group(2)] = sum(int(x) for x in m.group(3).split("+"))
    for m in re.finditer(r'std::find\(var_dna\.c?begin\(\), var_dna\.c?end\(\), gyper::to_vec\("([^"]*)"\)\) != var_dna\.end\(\)', body):
        e["contains"].append(m.group(1))
    for m in re.finditer(r'std::find\(var_dna\.c?begin\(\), var_dna\.c?end\(\), gyper::to_vec\("([^"]*)"\)\) == var_dna\.begin\(\)', body):
        e["var_dna"]["0"] = m.group(1)
    for m in re.finditer(r"REQUIRE\(ref_nodes\[(\d+)\]\.out_degree\(\) == (\d+)\)", body):
        e["out_degree"][m.group(1)] = int(m.group(2))
    for m in re.finditer(r"REQUIRE\(ref_nodes\[(\d+)\]\.get_var_index\((\d+)\) == (\d+)\)", body):
        e["var_index"]["%s,%s" % (m.group(1), m.group(2))] = int(m.group(3))
    for m in re.finditer(r"REQUIRE\(var_nodes\[(\d+)\]\.get_out_ref_index\(\) == (\d+)\)", body):
        e["out_ref"][m.group(1)] = int(m.group(2))
    m = re.search(r"REQUIRE\(haps\.size\(\) == (\d+)\)", body)
    if m:
        e["n_haplotypes"] = int(m.group(1))
    m = re.search(r"REQUIRE\(haps\[0\]\.get_genotype_num\(\) == (\d+)\)", body)
    if m:
        e["hap0_num"] = int(m.group(1))
    m = re.search(r'graph\.get_all_ref\(\) == gyper::to_vec\("([^"]*)"\)', body)
    if m:
        e["all_ref"] = m.group(1)
    return case


def main():
    cases = []
    for src in (SRC, os.path.join(os.path.dirname(SRC), "test_haplotypes.cpp")):
        text = strip_comments(open(src).read())
        parts = re.split(r"TEST_CASE\(", text)[1:]
        for p in parts:
            name = re.match(r'\s*"([^"]*)"', p).group(1)
            c = parse_case(name, p)
            if c:
                c["source"] = os.path.basename(src)
                cases.append(c)
    json.dump(cases, open(OUT, "w"), indent=1)
    print("wrote %d cases to %s" % (len(cases), OUT))
    for c in cases:
        e = c["expect"]
        print("  %-60.60s recs=%d add_all=%d nref=%s nvar=%s dna=%d contains=%d" % (c["name"], len(c["records"]), c["add_all_variants"],
              e.get("n_ref"), e.get("n_var"), len(e["ref_dna"]) + len(e["var_dna"]), len(e["contains"])))


if __name__ == "__main__":
    main()
