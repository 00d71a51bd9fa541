#!/bin/bash
# Phase cycle counters of one extra leg (profiling build): bash tools/prof_leg.sh {cfg3|clusters|repeats|cfg5} [bench.py arguments]
set -u
export GTX_BENCH_FULL_LINE=1  # bench.py prints its whole record (the default line is the compact one the driver parses)
leg=${1:-cfg3}; shift
mkdir -p gpurun_out
GTX_LIB=libgtx_prof.so python tools/run_extra_leg.py "$leg" --no-cpu-baseline "$@" > gpurun_out/prof_${leg}.json 2> gpurun_out/prof_${leg}.txt
grep -A13 "phase cycles" gpurun_out/prof_${leg}.txt
python - <<PY
import json
j = json.load(open("gpurun_out/prof_${leg}.json"))
print({k: j[k] for k in ("ms_per_step", "reads_per_s", "align_passes_ms", "pass_shares") if k in j})
print(j.get("align_kernels"))
PY
