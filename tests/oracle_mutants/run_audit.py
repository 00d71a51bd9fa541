#!/usr/bin/env python3
"""Mutation audit of the oracle's UNPINNED half (seed chaining with variants, walks, filters, orientation / pair selection,
explain_to_score, coverage, connections, the SV coverage model): does the ground-truth suite -- the tests that hold the oracle
to hand-worked or simulated truth, NOT to the product -- notice a one-token misreading of the reference?

Every entry of mutants.json is one such misreading: a unique piece of oracle text and what it is replaced by.  For each, the
oracle is copied, changed, compiled (g++ -O1) and the kill suite is run against it (GTO_LIB); a mutant that no test fails on
SURVIVES.  Results go to audit.json (committed; tests/test_oracle_mutants.py checks it against mutants.json and re-runs a sample).

    python tests/oracle_mutants/run_audit.py [-j 6] [--only ID ...]"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
KILL_SUITE = ["tests/test_oracle_truth.py", "tests/test_oracle_handworked.py", "tests/test_oracle_pinned.py", "tests/test_sv_vcf.py::test_coverage_model_hand_worked", "tests/test_sv_vcf.py::test_coverage_model_hand_worked_duplications_sizes_and_points",
              "tests/test_oracle_vcf_truth.py", "tests/test_oracle_handworked_pairs.py", "tests/test_oracle_merge.py", "tests/test_oracle_truth_walks.py", "tests/test_oracle_truth_pairs.py"]


def apply(mutant, oracle_dir):
    path = os.path.join(oracle_dir, mutant["file"])
    text = open(path).read()
    n = text.count(mutant["find"])
    if n != 1:
        raise SystemExit("mutant %s: its text occurs %d times in %s (must be 1)" % (mutant["id"], n, mutant["file"]))
    open(path, "w").write(text.replace(mutant["find"], mutant["replace"], 1))


def run_one(mutant):
    tmp = tempfile.mkdtemp(prefix="gto_mutant_")
    try:
        work = os.path.join(tmp, "oracle")
        shutil.copytree(os.path.join(ROOT, "oracle"), work, ignore=shutil.ignore_patterns("*.so", "_ref"))
        apply(mutant, work)
        so = os.path.join(tmp, "libgto_mutant.so")
        cc = subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-w", "-shared", "-o", so, os.path.join(work, "gto_capi.cpp")], capture_output=True, text=True)
        if cc.returncode != 0:
            return dict(id=mutant["id"], status="does not compile", detail=cc.stderr[-300:])
        env = dict(os.environ, GTO_LIB=so)
        t = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + KILL_SUITE, cwd=ROOT, env=env, capture_output=True, text=True)
        if t.returncode == 0:
            return dict(id=mutant["id"], status="SURVIVED")
        killers = [l.split(" ")[1] for l in t.stdout.splitlines() if l.startswith("FAILED ") or l.startswith("ERROR ")]
        return dict(id=mutant["id"], status="killed", by=killers[:1] or ["?"])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-j", type=int, default=6)
    ap.add_argument("--only", nargs="*")
    a = ap.parse_args()
    mutants = json.load(open(os.path.join(HERE, "mutants.json")))
    if a.only:
        mutants = [m for m in mutants if m["id"] in a.only]
    # the unmodified oracle has to pass the suite first
    base = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + KILL_SUITE, cwd=ROOT, capture_output=True, text=True)
    if base.returncode != 0:
        raise SystemExit("the kill suite fails on the unmodified oracle:\n" + base.stdout[-2000:])
    with ThreadPoolExecutor(a.j) as pool:
        results = list(pool.map(run_one, mutants))
    for r in results:
        print("%-34s %s %s" % (r["id"], r["status"], r.get("by", r.get("detail", ""))))
    killed = sum(r["status"] == "killed" for r in results)
    print("%d of %d mutants killed" % (killed, len(results)))
    if not a.only:
        json.dump(dict(kill_suite=KILL_SUITE, killed=killed, total=len(results), results=results), open(os.path.join(HERE, "audit.json"), "w"), indent=1)
        open(os.path.join(HERE, "audit.json"), "a").write("\n")


if __name__ == "__main__":
    main()
