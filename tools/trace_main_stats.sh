#!/bin/bash
# Kernel trace of the main workload (no extra legs), the top kernels printed: bash tools/trace_main_stats.sh [tag] [bench.py arguments]   (GTX_LIB selects the build)
export GTX_BENCH_FULL_LINE=1  # bench.py prints its whole record (the default line is the compact one the driver parses)
tag=${1:-main}; shift
export TMPDIR=/tmp; R=$PWD; mkdir -p $R/gpurun_out/trace_$tag; cd /tmp
timeout 250 rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/trace_$tag -o main -- python $R/bench.py --no-cpu-baseline --no-extra "$@" > $R/gpurun_out/trace_$tag/bench.log 2>&1
cd $R
f=$(find gpurun_out/trace_$tag -name "*kernel_stats.csv" | head -1)
python - "$f" gpurun_out/trace_$tag/bench.log <<PY
import csv, sys, json
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    print(r["Name"][:58].ljust(58), r["Calls"].rjust(6), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(10), "us")
for l in open(sys.argv[2]):
    if l.startswith("{"):
        d = json.loads(l); print("ms_per_step", round(d["ms_per_step"], 4), "under trace;", d["config"]["streams"]["schedule"], "; bench.py says the dominant kernel takes", round(d["roofline"]["kernel_ms"], 4), "ms, frac", round(d["roofline"]["frac"], 3))
PY
find gpurun_out/trace_$tag -type f ! -name "*stats.csv" ! -name "*.log" -delete
