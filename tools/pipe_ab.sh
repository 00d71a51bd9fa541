#!/bin/bash
# The BAM files -> VCF text leg alone, in fresh processes: bash tools/pipe_ab.sh <rounds> "ENV=.. ENV=.." ["ENV=.." ...]
# (environment per variant, e.g. "GTX_BGZF_THREADS=4" "GTX_BGZF_THREADS=16 GTX_BGZF_LINGER_US=0" "GTX_LIB=libgtx_x.so")
export TMPDIR=/tmp
rounds=$1; shift
run() {
  out=$(env $1 timeout 300 python tools/run_extra_leg.py pipeline --no-cpu-baseline 2>/tmp/pipe.err | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('%.1f M reads/s wall %.2f | thread-s %s | decode/thread %.2f M/s | text equal %s %s' % (j['reads_per_s']/1e6, j['wall_s'], j['host_thread_seconds'], j['records_per_s_per_thread']['decode']/1e6, j['vcf_equals_resident_run'], j.get('vcf_first_differences', '')))")
  echo "[$1] $out"; tail -1 /tmp/pipe.err | grep -v amdgpu
}
for round in $(seq 1 $rounds); do for v in "$@"; do run "$v"; done; done
