"""bench.py's own N>1 launch path on CPU: `python bench.py --gpus 2` has to start two ranks itself, build the process
group (gloo here, RCCL on a GPU node), sum a packed buffer over them, take the maximum time over the ranks and print one
line from rank 0 whose n_gpus is the size of the group -- the control flow the 8-GPU run of the driver goes through,
without device work (--dry-run)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600, env=env)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines


def test_gpus_2_starts_two_ranks_and_reports_them():
    p, lines = _run(["--gpus", "2", "--dry-run", "--backend", "gloo", "--steps", "3", "--warmup", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, "exactly one JSON line (rank 0): %r" % lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["dry_run"] is True and j["reduce_ok"] is True
    assert j["value"] is None and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"


def test_single_rank_needs_no_launcher():
    p, lines = _run(["--dry-run", "--steps", "2"])
    assert p.returncode == 0, p.stderr[-2000:]
    j = json.loads(lines[-1])
    assert j["n_gpus"] == 1 and j["reduce_ok"] is True


def test_world_size_has_to_match_gpus():
    """under a launcher that started a different number of ranks than --gpus the bench refuses instead of mislabelling"""
    p, lines = _run(["--gpus", "2", "--dry-run", "--backend", "gloo"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert p.returncode != 0 and not lines
    assert "WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_two_ranks_run_the_cfg4_shape():
    """N > 1 is BASELINE configs[3]: 1000 samples, reads sharded by read over the ranks, one packed sum of the 1000-sample
    block per step.  The dry run sums a block of exactly that size over two gloo ranks and reports the job's shape, the
    exchange's size and time and every rank's own step time; --scaling strong splits --total-reads over the ranks."""
    p, lines = _run(["--gpus", "2", "--dry-run", "--backend", "gloo", "--steps", "2", "--warmup", "0"])
    assert p.returncode == 0, p.stderr[-2000:]
    c = json.loads(lines[0])["config"]
    assert c["samples"] == 1000 and c["reads_per_rank"] == [10_000_000, 10_000_000]
    # u64 statistics of 1000 haplotypes + 2000 alleles, u32 counters of 1000 samples x (3000 + 2000 + 4000) + 1000 + 12000
    assert c["reduced_bytes_per_step"] == 8 * 5000 + 4 * (1000 * 9000 + 13000)
    assert c["reduce_ms"] > 0 and len(c["per_rank_ms_per_step"]) == 2 and all(x > 0 for x in c["per_rank_ms_per_step"])
    p, lines = _run(["--gpus", "2", "--dry-run", "--backend", "gloo", "--steps", "1", "--warmup", "0", "--scaling", "strong", "--total-reads", "80000001"])
    assert p.returncode == 0, p.stderr[-2000:]
    j = json.loads(lines[0])
    assert j["scaling"] == "strong" and j["config"]["reads_per_rank"] == [40_000_001, 40_000_000] and j["reduce_ok"] is True
    p, lines = _run(["--dry-run", "--steps", "1"])  # one GPU stays cfg2
    assert json.loads(lines[-1])["config"]["samples"] == 1 and json.loads(lines[-1])["scaling"] == "weak"


def test_the_line_of_stdout_fits_the_drivers_tail():
    """The driver keeps a bounded tail of the run's output: round 4's 17 KB line was parsed, round 5's 24 KB line was not
    (BENCH_r05.json: "parsed": null).  The line of stdout is the compact one -- the contract's keys, `roofline`,
    `cpu_baseline`, every extra leg as a few numbers -- made here from round 5's whole record."""
    sys.path.insert(0, ROOT)
    import bench
    out = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))
    line = bench.compact_line(out)
    text = json.dumps(line)
    assert len(text) < 6000, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in line and line[k] == (out[k] if k not in ("config", "roofline", "cpu_baseline") else line[k])
    assert line["config"]["workload"].startswith("cfg2: 1 sample, 10000000 synthetic 150 bp reads")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert line["roofline"][k] == out["roofline"][k]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert line["cpu_baseline"][k] == out["cpu_baseline"][k]
    assert set(line["config"]["extra"]) == set(out["config"]["extra"])
