#!/bin/bash
# PMC counters (default: the instruction cache's; GTX_PMC="..." for others) of the kernels of one extra leg, one step at a time:
#   gpurun -- 'bash tools/pmc_icache.sh cfg3'      (under `timeout`: a counter set the profiler aborts on must not hold the box)
set -u
LEG=${1:-cfg3}; shift
OUT=$PWD/gpurun_out/pmc_icache_$LEG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout 150 rocprofv3 --output-format csv --pmc ${GTX_PMC:-SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH} -d $OUT -o leg -- python $REPO/tools/run_extra_leg.py $LEG --no-cpu-baseline --lanes 1 "$@" > $OUT/leg.log 2>&1
cd $REPO
f=$(find $OUT -name '*counter_collection.csv' | head -1)
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items(), key=lambda t: -max(t[1].values()))[:6]:
    print("%-40s " % k[:40] + "  ".join("%s %.4g" % (c.replace("SQC_", "").replace("SQ_", ""), x) for c, x in sorted(v.items())))
PY
find $OUT -type f ! -name '*.log' -size +2M -delete
