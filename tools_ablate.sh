#!/bin/bash
# Instruction counts per phase of gtx_align_kernel: builds that stop after a phase (GTX_STOP_AFTER) are profiled with
# rocprofv3 PMC counters; differences between consecutive builds = instructions of that phase.  Usage (GPU box):
#   bash tools_ablate.sh "<extra bench args>"
set -u
EXTRA=${1:-}
export TMPDIR=/tmp
R=$PWD
cd /tmp
for k in s0 s1 s2 s4 full; do
  lib=libgtx_$k.so; [ $k = full ] && lib=libgtx.so
  GTX_LIB=$lib rocprofv3 --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS -d /tmp/p_$k -o b -- python $R/bench.py --reads 2000000 --steps 1 --warmup 0 --no-cpu-baseline $EXTRA > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections
for k in ["s0","s1","s2","s4","full"]:
    agg = collections.defaultdict(float)
    for f in glob.glob("/tmp/p_%s/**/*counter_collection.csv" % k, recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Kernel_Name"].startswith("gtx::gtx_align"):
                agg[row["Counter_Name"]] += float(row["Counter_Value"])
    print(k, " ".join("%s=%.0f" % (c.replace("SQ_INSTS_",""), v/2e6) for c, v in sorted(agg.items())))
PY
