#!/bin/bash
# Steps in flight, A/B: bash tools/gpu_lanes.sh <leg|main> <lanes> [<lanes> ...]   (leg: cfg3 | clusters | repeats | main = the cfg2 step)
set -u
leg=$1; shift
for n in "$@"; do
  if [ "$leg" = main ]; then
    python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extra --lanes $n 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('lanes $n: %.3f G/s step %.3f ms' % (j['value'] / 1e9, j['ms_per_step']), j['config']['streams'].get('schedule'))"
  else
    python tools/run_extra_leg.py $leg --no-cpu-baseline --lanes $n 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('lanes $n: %.3f G/s step %.3f ms' % (j['reads_per_s'] / 1e9, j['ms_per_step']), j['schedule'], j['calibration'])"
  fi
done
