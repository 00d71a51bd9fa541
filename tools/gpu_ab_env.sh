#!/bin/bash
# A/B of environment switches on one box: bash tools/gpu_ab_env.sh "VAR=a" "VAR=b" ...  (each argument is an env assignment list)
set -u
mkdir -p gpurun_out
for round in 1 2; do
for cfg in "$@"; do
  out=$(env $cfg python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
k = j['roofline']['align_kernels']
print('%.3f G/s  step %.3f ms %s | ' % (j['value'] / 1e9, j['ms_per_step'], str(j['config'].get('calls_checksum'))[:12]) + ' '.join('%s %.3f' % (n.replace('gtx_align_', '').replace('_kernel', ''), v['ms']) for n, v in k.items()))")
  echo "[$cfg] $out" | tee -a gpurun_out/ab_env.log
done; done
