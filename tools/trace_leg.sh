#!/bin/bash
# Kernel trace of one extra leg of bench.py.  Usage: gpurun -- 'bash tools/trace_leg.sh repeats [bench.py arguments]'
set -u
export GTX_BENCH_FULL_LINE=1  # bench.py prints its whole record (the default line is the compact one the driver parses)
LEG=${1:-repeats}; shift
OUT=$PWD/gpurun_out/trace_$LEG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o leg -- python $REPO/tools/run_extra_leg.py $LEG "$@" > $OUT/leg.log 2>&1
cd $REPO
grep -m1 "^{" $OUT/leg.log | cut -c1-1200
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print("%-60s calls %5s total %10.3f ms avg %10.3f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
find $OUT -type f ! -name '*stats.csv' ! -name '*.log' -delete
