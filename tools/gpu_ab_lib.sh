#!/bin/bash
# A/B of library builds on one box (cfg2 step only): bash tools/gpu_ab_lib.sh libgtx.so libgtx_x.so ...  (two rounds, interleaved)
set -u
mkdir -p gpurun_out
for round in 1 2; do
for lib in "$@"; do
  out=$(GTX_LIB=$lib python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
k = j['roofline']['align_kernels']
print('%.3f G/s  step %.3f ms | align %.3f | ' % (j['value'] / 1e9, j['ms_per_step'], j['roofline']['align_wall_ms']) + ' '.join('%s %.3f' % (n.replace('gtx_align_', '').replace('_kernel', ''), v['ms']) for n, v in k.items()))")
  echo "[$lib] $out" | tee -a gpurun_out/ab_lib.log
done; done
