#!/bin/bash
# One GPU visit: the parity suite, then (optional arguments) bench lines through tools_gpu_ab.sh.
# Usage (from the repo root): gpurun -- 'bash tools_gpu_check.sh [<lib>:<snp-every>:<reads>[:<GTX_EXPRESS4>] ...]'
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/gpu_tests.log
tail -6 gpurun_out/gpu_tests.log
if [ $# -gt 0 ]; then bash tools_gpu_ab.sh "$@"; fi
