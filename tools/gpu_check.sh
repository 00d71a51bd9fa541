#!/bin/bash
# One GPU visit: the parity suite, then a bench line; optional arguments go to tools/gpu_ab.sh.
# Usage (from the repo root): gpurun -- 'bash tools/gpu_check.sh [<lib>:<snp-every>:<reads>[:<GTX_EXPRESS4>] ...]'
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/gpu_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/gpu_tests.log
tail -16 gpurun_out/gpu_tests.log
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench exit $?"; tail -3 gpurun_out/bench.err; cat gpurun_out/bench.json
if [ $# -gt 0 ]; then bash tools/gpu_ab.sh "$@"; fi
