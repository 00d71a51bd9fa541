#!/bin/bash
# A/B of bench.py argument sets on one box (cfg2 step only): every argument is one (quoted) argument string; two rounds
set -u
mkdir -p gpurun_out
for round in 1 2; do
for a in "$@"; do
  out=$(python bench.py --warmup 2 --no-cpu-baseline --no-extra $a 2>gpurun_out/ab_args2.err | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
k = j['roofline']['align_kernels']
print('%.3f G/s  step %.3f ms | ' % (j['value'] / 1e9, j['ms_per_step']) + ' '.join('%s %.3f' % (n.replace('gtx_align_', '').replace('_kernel', ''), v['ms']) for n, v in k.items()) + ' | frac %.3f %s' % (j['roofline']['frac'], str(j['config'].get('calls_checksum', {}).get('vcf_sha256'))[:10]))")
  echo "[$a] $out" | tee -a gpurun_out/ab_args2.log
  tail -2 gpurun_out/ab_args2.err
done; done
