#!/bin/bash
# kernel durations (rocprofv3 --kernel-trace --stats) of the cfg2 bench under an environment switch: bash tools/kt.sh VAR=a VAR=b
export TMPDIR=/tmp; R=$PWD; cd /tmp
for cfg in "$@"; do
  rm -rf $R/gpurun_out/kt_tmp
  env $cfg rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/kt_tmp -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
  echo "[$cfg]"
  python - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/kt_tmp/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if n.startswith("gtx::gtx_") :
            print("  %-36s calls %3s  avg %8.1f us" % (n[5:], r["Calls"], float(r["AverageNs"]) / 1000))
PY
done
rm -rf $R/gpurun_out/kt_tmp
