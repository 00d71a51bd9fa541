#!/bin/bash
# bench.py without the CPU leg; prints the headline figures and the repeats leg.  Usage: gpurun -- 'bash tools/gpu_bench_repeats.sh [pytest -k expression]'
set -u
export GTX_BENCH_FULL_LINE=1  # bench.py prints its whole record (the default line is the compact one the driver parses)
mkdir -p gpurun_out
if [ $# -gt 0 ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "$1" 2>&1 | tail -5; fi
python bench.py --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/b3.json 2> gpurun_out/b3.err; echo exit $?; tail -3 gpurun_out/b3.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/b3.json') if l.startswith('{')][-1])
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel_ms'], j['config']['calls_checksum'].get('matches_pinned'))
print(json.dumps(j['config']['extra']['repeats']))
PY
