#!/usr/bin/env python3
"""Mechanical mutation audit of the oracle's unpinned half: EVERY relational operator, EVERY integer literal and EVERY && / || in the
functions the reference holds no vector for (RANGES below) is changed once -- < <-> <=, > <-> >=, == <-> !=, N -> N + 1, && <-> || --,
the changed oracle is compiled and the ground-truth suite (run_audit.KILL_SUITE: the tests that hold the oracle to hand-worked or
simulated truth, never to the product) is run against it.  A mutant no test fails on SURVIVES: either the two texts mean the same
there (a bound that is never met, a reserve() size) or the suite does not pin that token.  audit_auto.json (committed) holds the
counts per file and function and every survivor with its line, so that the next hand-worked vector can be aimed.

    python tests/oracle_mutants/run_auto.py [-j 5] [--file gto.hpp] [--limit N] [--every K]"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import run_audit  # noqa: E402

# (file, first line marker, last line marker): the text between the first line that holds `first` and the first line after it that
# holds `last` -- the unpinned functions, by what they restate
RANGES = [
    ("gto.hpp", "Graph::get_locations_of_an_actual_position(uint32_t pos, Path const & path, bool is_special) const", "struct ReferenceDepth"),
    ("gto.hpp", "void genotype_only(ReadRecord const & rec, bool update_prev)", "std::vector<std::vector<Call>> sample_calls() const"),
    ("gto.hpp", "std::map<PhKey, std::map<PhKey, int8_t>> phase_flags() const", "} // namespace gto"),
    ("gto_vcf.hpp", "void scan_calls() // variant.cpp:230-428", "struct WriteOptions"),
    ("gto_sv.hpp", "inline SampleCall make_bi_allelic_call(", "inline void remove_common_prefix("),
    ("gto_sv.hpp", "inline void update_per_allele_stats(", "inline void write_records(std::ostream & out"),
]
REL = {"<": "<=", "<=": "<", ">": ">=", ">=": ">", "==": "!=", "!=": "=="}


def strip_comment(line):
    out, in_str, i = [], False, 0
    while i < len(line):
        c = line[i]
        if c == '"' and (i == 0 or line[i - 1] != "\\"):
            in_str = not in_str
        if not in_str and line.startswith("//", i):
            break
        out.append(" " if in_str and c != '"' else c)  # (string contents are not code)
        i += 1
    return "".join(out)


def mutants_of(path_rel):
    lines = open(os.path.join(ROOT, "oracle", path_rel)).read().split("\n")
    spans = []
    for f, first, last in RANGES:
        if f != path_rel:
            continue
        a = next(i for i, l in enumerate(lines) if first in l)
        b = next(i for i in range(a + 1, len(lines)) if last in lines[i])
        spans.append((a, b))
    out = []
    func = "?"
    for a, b in spans:
        for ln in range(a, b):
            raw = lines[ln]
            m = re.match(r"^\s*(?:inline |static |template.*|)[A-Za-z_0-9:<>,&\* ]*?([A-Za-z_][A-Za-z_0-9:]*)\(.*\)\s*(?:const)?\s*(?://.*)?$", raw)
            if not m:  # a signature that goes on in the next line
                m = re.match(r"^(?:inline |static )[A-Za-z_0-9:<>,&\* ]*?([A-Za-z_][A-Za-z_0-9:]*)\([^()]*,\s*$", raw)
            if m and not raw.strip().startswith(("if", "for", "while", "return", "else", "switch")) and ";" not in strip_comment(raw):
                func = m.group(1)
            code = strip_comment(raw)
            if code.lstrip().startswith("#") or "static_assert" in code:
                continue
            for mm in re.finditer(r" (<=|>=|==|!=|<|>) ", code):
                out.append(dict(file=path_rel, line=ln + 1, col=mm.start(1), find=mm.group(1), replace=REL[mm.group(1)], kind="relational", func=func))
            for mm in re.finditer(r"(&&|\|\|)", code):
                out.append(dict(file=path_rel, line=ln + 1, col=mm.start(1), find=mm.group(1), replace="||" if mm.group(1) == "&&" else "&&", kind="logical", func=func))
            for mm in re.finditer(r"(?<![A-Za-z_0-9\.])(\d+)(?![\.\dxXa-fA-F])(u|l|ul|ull|lu|)\b", code):
                if mm.start(1) > 0 and code[mm.start(1) - 1] in "xX":
                    continue
                out.append(dict(file=path_rel, line=ln + 1, col=mm.start(1), find=mm.group(1), replace=str(int(mm.group(1)) + 1), kind="literal", func=func))
    for k, m in enumerate(out):
        m["id"] = "%s:%d:%d:%s" % (m["file"], m["line"], m["col"], m["kind"])
        m["text"] = lines[m["line"] - 1].strip()[:160]
    return out


def run_one(m):
    tmp = tempfile.mkdtemp(prefix="gto_auto_")
    try:
        work = os.path.join(tmp, "oracle")
        shutil.copytree(os.path.join(ROOT, "oracle"), work, ignore=shutil.ignore_patterns("*.so", "_ref"))
        path = os.path.join(work, m["file"])
        lines = open(path).read().split("\n")
        l = lines[m["line"] - 1]
        assert l[m["col"]:m["col"] + len(m["find"])] == m["find"], m
        lines[m["line"] - 1] = l[:m["col"]] + m["replace"] + l[m["col"] + len(m["find"]):]
        open(path, "w").write("\n".join(lines))
        so = os.path.join(tmp, "libgto_mutant.so")
        cc = subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-w", "-shared", "-o", so, os.path.join(work, "gto_capi.cpp")], capture_output=True, text=True)
        if cc.returncode != 0:
            return dict(m, status="does not compile")
        env = dict(os.environ, GTO_LIB=so)
        try:
            t = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + run_audit.KILL_SUITE, cwd=ROOT, env=env, capture_output=True,
                               text=True, timeout=300)
        except subprocess.TimeoutExpired:
            return dict(m, status="killed", by="timeout (a loop that no longer ends)")
        if t.returncode == 0:
            return dict(m, status="SURVIVED")
        killers = [x.split(" ")[1] for x in t.stdout.splitlines() if x.startswith("FAILED ") or x.startswith("ERROR ")]
        return dict(m, status="killed", by=(killers[:1] or ["crash"])[0])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-j", type=int, default=5)
    ap.add_argument("--file", nargs="*", default=["gto.hpp", "gto_vcf.hpp", "gto_sv.hpp"])
    ap.add_argument("--limit", type=int, default=0)
    ap.add_argument("--every", type=int, default=1, help="take every K-th mutant (a sample)")
    ap.add_argument("--lines", default="", help="A-B: only the mutants on these lines (a look at one function)")
    ap.add_argument("--merge", action="store_true", help="put this run's results into audit_auto.json in place of the same mutants' old ones")
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--ids", default="", help="a file with mutant ids, one per line: only those")
    a = ap.parse_args()
    mutants = [m for f in a.file for m in mutants_of(f)][::a.every]
    if a.lines:
        lo, hi = (int(x) for x in a.lines.split("-"))
        mutants = [m for m in mutants if lo <= m["line"] <= hi]
    if a.ids:
        wanted = set(open(a.ids).read().split())
        mutants = [m for m in mutants if m["id"] in wanted]
    if a.limit:
        mutants = mutants[:a.limit]
    if a.list:
        for m in mutants:
            print(m["id"], m["func"], "|", m["text"])
        print(len(mutants), "mutants")
        return
    base = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + run_audit.KILL_SUITE, cwd=ROOT, capture_output=True, text=True)
    if base.returncode != 0:
        raise SystemExit("the kill suite fails on the unmodified oracle:\n" + base.stdout[-2000:])
    t0 = time.time()
    results = []
    with ThreadPoolExecutor(a.j) as pool:
        for k, r in enumerate(pool.map(run_one, mutants)):
            results.append(r)
            if (k + 1) % 50 == 0:
                sys.stderr.write("%d / %d  (%.0f s)\n" % (k + 1, len(mutants), time.time() - t0))
    by_file = {}
    for r in results:
        d = by_file.setdefault(r["file"], {}).setdefault(r["func"], dict(mutants=0, killed=0, survived=0, does_not_compile=0))
        d["mutants"] += 1
        d[{"killed": "killed", "SURVIVED": "survived"}.get(r["status"], "does_not_compile")] += 1
    survivors = [dict(id=r["id"], func=r["func"], change="%s -> %s" % (r["find"], r["replace"]), text=r["text"]) for r in results if r["status"] == "SURVIVED"]
    out = dict(kill_suite=run_audit.KILL_SUITE, every=a.every, total=len(results), killed=sum(r["status"] == "killed" for r in results),
               survived=len(survivors), does_not_compile=sum(r["status"] == "does not compile" for r in results), per_function=by_file, survivors=survivors)
    if a.merge:  # the mutants of this run replace their entries in audit_auto.json (after a new vector: only its function is run again)
        full = json.load(open(os.path.join(HERE, "audit_auto.json")))
        done = {r["id"] for r in results}
        before = {s["id"] for s in full["survivors"]}
        full["survivors"] = sorted([s for s in full["survivors"] if s["id"] not in done] + survivors, key=lambda s: (s["id"].split(":")[0], int(s["id"].split(":")[1]), int(s["id"].split(":")[2])))
        for r in results:
            d = full["per_function"][r["file"]][r["func"]]
            was = "survived" if r["id"] in before else None
            now = {"killed": "killed", "SURVIVED": "survived"}.get(r["status"])
            if was == "survived" and now == "killed":
                d["survived"] -= 1
                d["killed"] += 1
            elif was is None and now == "survived":  # (was killed or did not compile)
                d["survived"] += 1
                d["killed"] -= 1
        full["killed"] = sum(c["killed"] for f in full["per_function"].values() for c in f.values())
        full["survived"] = len(full["survivors"])
        full["kill_suite"] = run_audit.KILL_SUITE
        json.dump(full, open(os.path.join(HERE, "audit_auto.json"), "w"), indent=1)
        open(os.path.join(HERE, "audit_auto.json"), "a").write("\n")
        print("merged into audit_auto.json: %d killed, %d survived" % (full["killed"], full["survived"]))
    name = "audit_auto.json" if a.every == 1 and not a.limit and not a.lines and not a.ids and len(a.file) == 3 else "audit_auto_partial.json"
    if a.merge:
        name = "audit_auto_partial.json"
    json.dump(out, open(os.path.join(HERE, name), "w"), indent=1)
    open(os.path.join(HERE, name), "a").write("\n")
    print("%d mutants: %d killed, %d survived, %d do not compile  (%.0f s) -> %s" % (out["total"], out["killed"], out["survived"], out["does_not_compile"], time.time() - t0, name))


if __name__ == "__main__":
    main()
