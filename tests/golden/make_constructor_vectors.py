"""Extracts the known answers of the reference's constructor tests into constructor_cases.json.

Run in the build container only:  python tests/golden/make_constructor_vectors.py
Input : /root/reference/test/graph/test_constructor.cpp (the active TEST_CASEs; the SV and b37 cases are commented out
        upstream).  Each case calls create_test_graph(fasta, vcf, region, flag) -- `flag` lands in construct_graph's
        is_sv_graph parameter (test/help_functions.hpp:12-31) -- and REQUIREs node counts, connectivity, orders, bases,
        events.
Output: tests/golden/constructor_cases.json: per case the call arguments and a list of assertions
        [kind, node index, (argument), expected value]; tests/test_constructor_vectors.py replays them on
        gtx_graph_from_files.  Data only -- no reference source text is kept."""
import json
import os
import re

SRC = "/root/reference/test/graph/test_constructor.cpp"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "constructor_cases.json")

text = open(SRC).read()
text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)  # disabled cases
text = re.sub(r"//[^\n]*", "", text)
cases = []
add_all = False
for m in re.finditer(r'TEST_CASE\("([^"]+)"\)\s*\{', text):
    start = m.end()
    nxt = text.find("TEST_CASE(", start)
    body = text[start:nxt if nxt >= 0 else len(text)]
    call = re.search(r'create_test_graph\("([^"]+)",\s*"([^"]+)",\s*"([^"]+)"(?:,\s*(true|false))?\)', body)
    if not call:
        continue
    before = body[:call.start()]
    for f in re.finditer(r"add_all_variants\s*=\s*(true|false)", before):
        add_all = f.group(1) == "true"
    case = dict(name=m.group(1), fasta=os.path.basename(call.group(1)), vcf=os.path.basename(call.group(2)).replace(".vcf.gz", ".vcf"),
                region=call.group(3), is_sv_graph=(call.group(4) or "true") == "true", add_all_variants=add_all, checks=[])
    for r in re.finditer(r"REQUIRE\((.*?)\);", body[call.end():], flags=re.S):
        e = " ".join(r.group(1).split())
        e = e.replace("gyper::", "").replace("graph.ref_nodes", "ref_nodes").replace("graph.var_nodes", "var_nodes")
        pats = [
            (r"^(ref|var)_nodes\.size\(\) == (\d+)$", lambda g: ["count", g[0], int(g[1])]),
            (r"^(ref|var)_nodes\[(\d+)\]\.out_degree\(\) == (\d+)$", lambda g: ["out_degree", g[0], int(g[1]), int(g[2])]),
            (r"^ref_nodes\[(\d+)\]\.get_var_index\((\d+)\) == (\d+)$", lambda g: ["var_index", int(g[0]), int(g[1]), int(g[2])]),
            (r"^var_nodes\[(\d+)\]\.get_out_ref_index\(\) == (\d+)$", lambda g: ["out_ref", int(g[0]), int(g[1])]),
            (r"^(ref|var)_nodes\[(\d+)\]\.get_label\(\)\.order == (\d+)$", lambda g: ["order", g[0], int(g[1]), int(g[2])]),
            (r'^(ref|var)_nodes\[(\d+)\]\.get_label\(\)\.dna == to_vec\("([A-Z]*)"\)$', lambda g: ["dna", g[0], int(g[1]), g[2]]),
            (r"^var_nodes\[(\d+)\]\.(events|anti_events)\.size\(\) == (\d+)$", lambda g: ["set_size", int(g[0]), g[1], int(g[2])]),
            (r"^var_nodes\[(\d+)\]\.(events|anti_events)\.count\((-?\d+)\) == (\d+)$", lambda g: ["set_count", int(g[0]), g[1], int(g[2]), int(g[3])]),
            (r"^graph\.actual_poses\.size\(\) == (\d+)$", lambda g: ["n_special", int(g[0])]),
            (r"^graph\.(actual_poses|ref_reach_poses)\.size\(\) == (\d+)$", lambda g: ["n_special", int(g[1])]),
            (r"^graph\.(actual_poses|ref_reach_poses)\[(\d+)\] == (\d+)$", lambda g: ["special", g[0], int(g[1]), int(g[2])]),
            (r"^std::distance\(graph\.ref_reach_to_special_pos\.begin\(\), graph\.ref_reach_to_special_pos\.end\(\)\) == (\d+)$",
             lambda g: ["n_special_keys", int(g[0])]),
            (r"^graph\.ref_reach_to_special_pos\.count\((\d+)\) == (\d+)$", lambda g: ["special_key_count", int(g[0]), int(g[1])]),
        ]
        for pat, mk in pats:
            mm = re.match(pat, e)
            if mm:
                case["checks"].append(mk(mm.groups()))
                break
        else:
            raise SystemExit("unparsed assertion in %r: %s" % (case["name"], e))
    # flags set after the call (restored for the next case)
    for f in re.finditer(r"add_all_variants\s*=\s*(true|false)", body[call.end():]):
        add_all = f.group(1) == "true"
    cases.append(case)
json.dump(cases, open(OUT, "w"), indent=1)
print(len(cases), "cases,", sum(len(c["checks"]) for c in cases), "assertions ->", OUT)
