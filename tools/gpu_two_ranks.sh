#!/bin/bash
# The N > 1 control flow of bench.py on a one-GPU box: two ranks over gloo sharing device 0 (cfg4 shape, exchange, the
# reduce self-check).  Usage: gpurun -- 'bash tools/gpu_two_ranks.sh [reads per rank]'
set -u
export GTX_BENCH_FULL_LINE=1  # bench.py prints its whole record (the default line is the compact one the driver parses)
mkdir -p gpurun_out
GTX_BENCH_SHARE_DEVICE=1 timeout 600 python bench.py --gpus 2 --backend gloo --reads ${1:-2000000} --steps 3 --warmup 1 > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err
echo "two-rank bench exit $?"; tail -3 gpurun_out/bench_2rank.err; python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/bench_2rank.json") if l.startswith("{")][-1])
print({k: j[k] for k in ("value", "n_gpus", "ms_per_step", "scaling")}, j["config"]["reduce_check"], j["config"]["reduce"])
PY
