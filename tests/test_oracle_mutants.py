"""The committed mutation audit of the oracle's unpinned half (tests/oracle_mutants/): audit.json has to cover every mutant of
mutants.json, each killed by a ground-truth / hand-worked test unless the list itself says why it is expected to survive; and a
sample of them is re-run here (compile the changed oracle, run the test that is recorded as its killer) so that the record
cannot go stale silently.  The full audit: python tests/oracle_mutants/run_audit.py."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "oracle_mutants"))


def _load():
    mutants = json.load(open(os.path.join(HERE, "oracle_mutants", "mutants.json")))
    audit = json.load(open(os.path.join(HERE, "oracle_mutants", "audit.json")))
    return mutants, audit


def test_the_audit_covers_the_mutants_and_they_die():
    mutants, audit = _load()
    res = {r["id"]: r for r in audit["results"]}
    assert set(res) == {m["id"] for m in mutants} and len(mutants) >= 30
    for m in mutants:
        # the text a mutant changes must still be in the oracle, once
        text = open(os.path.join(ROOT, "oracle", m["file"])).read()
        assert text.count(m["find"]) == 1, "mutant %s no longer applies" % m["id"]
        if m.get("expect") == "survives":
            assert res[m["id"]]["status"] == "SURVIVED" and m.get("why"), m["id"]
        else:
            assert res[m["id"]]["status"] == "killed", "mutant %s is not noticed by the ground-truth suite" % m["id"]
            assert res[m["id"]]["by"][0].split("::")[0] in audit["kill_suite"] or res[m["id"]]["by"][0].split("::")[0] in [k.split("::")[0] for k in audit["kill_suite"]]
    assert audit["killed"] >= 28


def test_a_sample_of_the_mutants_is_killed_again():
    import run_audit
    mutants, audit = _load()
    res = {r["id"]: r for r in audit["results"]}
    for mid in ("eps_not_overlapping_1", "walk_budget_div_11", "single_longer_than_94"):
        m = next(x for x in mutants if x["id"] == mid)
        killer = res[mid]["by"][0]
        old = list(run_audit.KILL_SUITE)
        try:
            run_audit.KILL_SUITE[:] = [killer]  # (only the recorded killer: keeps this test at a few seconds per mutant)
            r = run_audit.run_one(m)
        finally:
            run_audit.KILL_SUITE[:] = old
        assert r["status"] == "killed", (mid, r)
    # ... and the unmodified oracle passes those very tests
    ok = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + [res[mid]["by"][0] for mid in ("eps_not_overlapping_1", "walk_budget_div_11")],
                        cwd=ROOT, capture_output=True, text=True)
    assert ok.returncode == 0, ok.stdout[-1000:]
