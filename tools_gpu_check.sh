#!/bin/bash
# One GPU visit: the parity suite, then bench lines for the reported workload and the two denser graphs
# (lean / wide build of pass 1).  Usage (from the repo root): gpurun -- 'bash tools_gpu_check.sh'
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/gpu_tests.log
tail -5 gpurun_out/gpu_tests.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_cfg2.json 2> gpurun_out/bench_cfg2.err
cut -c1-300 gpurun_out/bench_cfg2.json
for mode in lean wide; do
  GTX_EXPRESS4=$mode timeout 300 python bench.py --reads 4000000 --snp-every 100 --no-cpu-baseline > gpurun_out/bench_snp100_$mode.json 2> gpurun_out/bench_snp100_$mode.err
done
GTX_EXPRESS4=wide timeout 300 python bench.py --reads 4000000 --snp-every 25 --no-cpu-baseline > gpurun_out/bench_snp25_wide.json 2> gpurun_out/bench_snp25_wide.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]["align_passes_ms"]
        print(f, "%.1f M reads/s" % (d["value"] / 1e6), "ms/step %.2f" % d["ms_per_step"], "express %.2f general %.2f hbm %.2f handed %d" % (r["express"], r["general"], r["hbm_tables"], r["tasks_handed_to_general"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
