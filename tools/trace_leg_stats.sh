#!/bin/bash
# Kernel trace of one extra leg, the top kernels printed: bash tools/trace_leg_stats.sh {cfg3|clusters|...} [tag]   (GTX_LIB selects the build)
leg=${1:-cfg3}; tag=${2:-$leg}
export TMPDIR=/tmp; R=$PWD; mkdir -p $R/gpurun_out/trace_$tag; cd /tmp
timeout 250 rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/trace_$tag -o leg -- python $R/tools/run_extra_leg.py $leg --no-cpu-baseline > $R/gpurun_out/trace_$tag/leg.log 2>&1
cd $R
f=$(find gpurun_out/trace_$tag -name "*kernel_stats.csv" | head -1)
python - "$f" <<PY
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print(r["Name"][:58].ljust(58), r["Calls"].rjust(6), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(10), "us")
PY
find gpurun_out/trace_$tag -type f ! -name "*stats.csv" ! -name "*.log" -delete
