#!/bin/bash
# Kernel trace (start / end of every launch, per queue) of a short bench run: bash tools/trace_schedule.sh [bench args]
set -u
export GTX_BENCH_FULL_LINE=1  # bench.py prints its whole record (the default line is the compact one the driver parses)
OUT=$PWD/gpurun_out/sched
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --output-format csv --kernel-trace -d $OUT/trace -o bench -- python $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra "$@" > $OUT/bench.log 2>&1
cd $REPO
f=$(find $OUT -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), "launches; columns:", list(rows[0].keys()))
keep = [r for r in rows if "gtx" in r["Kernel_Name"]]
t0 = min(int(r["Start_Timestamp"]) for r in keep)
out = open(sys.argv[1].rsplit("/", 1)[0] + "/../../../sched_gtx.txt", "w") if False else open("gpurun_out/sched_gtx.txt", "w")
for r in keep[-140:]:
    out.write("%-34s q%-3s %10.1f %10.1f %8.1f\n" % (r["Kernel_Name"].split("(")[0][-34:], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
out.close()
PY
find $OUT -type f -size +2M -delete
