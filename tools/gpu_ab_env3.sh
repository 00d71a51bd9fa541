#!/bin/bash
# like tools/gpu_ab_env.sh, with the cfg3-like extra workload in the line
set -u
mkdir -p gpurun_out
for round in 1 2; do
for cfg in "$@"; do
  out=$(env $cfg python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
k = j['roofline']['align_kernels']
x = j['config']['extra']['cfg3']
print('%.3f G/s  step %.3f ms | ' % (j['value'] / 1e9, j['ms_per_step']) + ' '.join('%s %.3f' % (n.replace('gtx_align_', '').replace('_kernel', ''), v['ms']) for n, v in k.items()) + ' | cfg3 %.1f M/s step %.2f %s' % (x['reads_per_s'] / 1e6, x['ms_per_step'], {a: round(b, 2) for a, b in x['align_passes_ms'].items()}))")
  echo "[$cfg] $out" | tee -a gpurun_out/ab_env.log
done; done
