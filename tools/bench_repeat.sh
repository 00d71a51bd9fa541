#!/bin/bash
# The main line of bench.py in N fresh processes on one box: how often does a process' schedule misfire, and what does the line say then?
#   gpurun -- 'bash tools/bench_repeat.sh 12'
n=${1:-10}
export GTX_BENCH_FULL_LINE=1
for i in $(seq 1 $n); do
  python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
c = j['config']['streams']['calibration']
print('%.3f G/s  %.3f ms | %s | stag %.3f one %.3f whole %.3f retries %d %s' % (j['value'] / 1e9, j['ms_per_step'], j['config']['streams']['schedule'], c['staggered_ms_per_step'], c['one_at_a_time_ms_per_step'], c.get('whole_steps_on_streams_of_their_own_ms_per_step', 0), c['fresh_stream_retries'], 'MISFIRE' if 'staggered_schedule_did_not_overlap' in c else ''))"
done
