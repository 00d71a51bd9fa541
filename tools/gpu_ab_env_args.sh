#!/bin/bash
# A/B of (environment, bench.py arguments) pairs on one box: every argument is "ENV=.. ENV=.. -- bench args"; two rounds
set -u
mkdir -p gpurun_out
for round in 1 2; do
for a in "$@"; do
  e="${a%%--*}"; b="--${a#*--}"
  out=$(env $e python bench.py --warmup 2 --no-cpu-baseline --no-extra $b 2>gpurun_out/ab_env_args.err | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
k = j['roofline']['align_kernels']
print('%.3f G/s  step %.3f ms | ' % (j['value'] / 1e9, j['ms_per_step']) + ' '.join('%s %.3f' % (n.replace('gtx_align_', '').replace('_kernel', ''), v['ms']) for n, v in k.items()) + ' | frac %.3f %s' % (j['roofline']['frac'], str(j['config'].get('calls_checksum', {}).get('vcf_sha256'))[:10]))")
  echo "[$a] $out" | tee -a gpurun_out/ab_env_args.log
  tail -2 gpurun_out/ab_env_args.err | grep -v amdgpu.ids
done; done
