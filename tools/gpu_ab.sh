#!/bin/bash
# A/B on ONE box (box-to-box noise is a few percent), two interleaved rounds.  Every further argument is one variant:
#   "[ENV=value ...] [-- bench.py arguments]"        e.g.  bash tools/gpu_ab.sh "GTX_LIB=libgtx.so" "GTX_LIB=libgtx_x.so -- --lanes 1"
# Options in front of the variants:
#   --extra        keep bench.py's extra legs and show the cfg3 leg in the line (default: the cfg2 step only, --no-extra)
#   --leg <name>   run one extra leg alone instead (tools/run_extra_leg.py: cfg3 | clusters | repeats | cfg5)
#   --rounds <n>   rounds (default 2)
set -u
export GTX_BENCH_FULL_LINE=1  # bench.py prints its whole record (the default line is the compact one the driver parses)
extra=0; leg=""; rounds=2
while [ $# -gt 0 ]; do
  case "$1" in
    --extra) extra=1; shift ;;
    --leg) leg=$2; shift 2 ;;
    --rounds) rounds=$2; shift 2 ;;
    *) break ;;
  esac
done
mkdir -p gpurun_out
for round in $(seq 1 $rounds); do
for v in "$@"; do
  case "$v" in *" -- "*) e="${v%% -- *}"; b="${v#* -- }" ;; "-- "*) e=""; b="${v#-- }" ;; *) e="$v"; b="" ;; esac
  if [ -n "$leg" ]; then
    out=$(env $e python tools/run_extra_leg.py $leg --no-cpu-baseline $b 2>gpurun_out/ab.err | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('%.3f G/s  step %.3f ms | ' % (j['reads_per_s'] / 1e9, j['ms_per_step']) + ' '.join('%s %.3f' % (n.replace('gtx_align_', '').replace('_kernel', ''), x['ms']) for n, x in j['align_kernels'].items()) + ' | pass 0 %.3f general %.4f' % (j['pass_shares']['position_hinted_share'], j['pass_shares']['share_general']))")
  else
    out=$(env $e python bench.py --warmup 2 --no-cpu-baseline $([ $extra = 1 ] || echo --no-extra) $b 2>gpurun_out/ab.err | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
k = j['roofline']['align_kernels']
s = '%.3f G/s  step %.3f ms | ' % (j['value'] / 1e9, j['ms_per_step']) + ' '.join('%s %.3f' % (n.replace('gtx_align_', '').replace('_kernel', ''), x['ms']) for n, x in k.items())
s += ' | frac %.3f %s' % (j['roofline']['frac'], str(j['config'].get('calls_checksum', {}).get('vcf_sha256'))[:10])
x = j['config'].get('extra', {}).get('cfg3')
if x and 'reads_per_s' in x:
    s += ' | cfg3 %.1f M/s step %.2f %s' % (x['reads_per_s'] / 1e6, x['ms_per_step'], {a: round(b, 2) for a, b in x['align_passes_ms'].items()})
print(s)")
  fi
  echo "[$v] $out" | tee -a gpurun_out/ab.log
  tail -2 gpurun_out/ab.err | grep -v amdgpu.ids
done; done
