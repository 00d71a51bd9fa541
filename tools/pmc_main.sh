#!/bin/bash
# PMC counters of the main workload's kernels, one step at a time (GTX_LIB selects the build; GTX_PMC the counters):
#   gpurun -- 'GTX_LIB=libgtx.so bash tools/pmc_main.sh tag [bench.py arguments]'
set -u
tag=${1:-main}; shift
OUT=$PWD/gpurun_out/pmc_$tag
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp GTX_BENCH_FULL_LINE=1
REPO=$PWD
cd /tmp
timeout 170 rocprofv3 --output-format csv --pmc ${GTX_PMC:-SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR} -d $OUT -o m -- python $REPO/bench.py --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-extra "$@" > $OUT/bench.log 2>&1
cd $REPO
f=$(find $OUT -name '*counter_collection.csv' | head -1)
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items(), key=lambda t: -max(t[1].values())):
    if "gtx" in k:
        w = v.get("SQ_WAVES", 0) or 1
        print("%-36s " % k[-36:] + "  ".join("%s %.4g (%.1f/wave)" % (c.replace("SQ_", ""), x, x / w) for c, x in sorted(v.items())))
PY
find $OUT -type f ! -name '*.log' -size +2M -delete
