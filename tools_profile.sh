#!/bin/bash
# Profiling recipe used for profiles/: run on the GPU box from the repo root (gpurun -- 'bash tools_profile.sh r01').
# 1) kernel trace + stats of the bench command; 2) PMC passes (own runs, no trace domains) for instruction mix and HBM bytes.
set -u
TAG=${1:-r02}
READS=${2:-10000000}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o bench -- python $REPO/bench.py --reads $READS --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/bench_trace.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/pmc_sq -o bench -- python $REPO/bench.py --reads $READS --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $OUT/bench_pmc_sq.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq2 -o bench -- python $REPO/bench.py --reads $READS --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $OUT/bench_pmc_sq2.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- python $REPO/bench.py --reads $READS --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o bench -- python $REPO/bench.py --reads $READS --steps 1 --warmup 0 --no-cpu-baseline --no-extra > $OUT/bench_pmc_write.log 2>&1
cd $REPO
find $OUT -type f ! -name '*.csv' ! -name '*.log' -delete
find $OUT -name '*.csv' -size +4M -delete
find $OUT -name '*.csv' | head -50
for f in $OUT/*.log; do grep -m1 "^{" $f | cut -c1-400; done
python - <<PY
import csv, glob, os, collections, json, shutil
out = "$OUT"
summary = collections.defaultdict(dict)
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(f)):
        per[row.get("Kernel_Name", "?").split("(")[0]][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, d in per.items():
        if k.startswith("gtx::"):
            summary[k].update(d)
summary["_note"] = "sums over the launches of one bench.py run (--reads $READS, 1 step, no warm-up) per PMC pass; tools_profile.sh"
json.dump(summary, open(out + "/pmc_summary.json", "w"), indent=1, sort_keys=True)
ks = sorted(glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True))
if ks:
    shutil.copy(ks[0], out + "/kernel_stats.csv")
kernels = {}
for k, d in summary.items():
    if isinstance(d, dict) and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        # FETCH_SIZE / WRITE_SIZE count kilobytes; reported raw (the x2 gfx950 correction of the guide is calibrated for
        # 16 B/lane coalesced streams and is listed beside it for the reader)
        b = (d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
        kernels[k.replace("gtx::", "")] = {"fetch_size_kb": d["FETCH_SIZE"], "write_size_kb": d["WRITE_SIZE"], "hbm_bytes_per_launch": b,
                                           "hbm_bytes_per_launch_fetch_x2": (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024,
                                           "bytes_per_read_of_the_batch": b / $READS}
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools_profile.sh), one launch of every alignment kernel over $READS reads (cfg2, 1 step, no warm-up)",
           "reads_per_launch": $READS, "kernels": kernels}, open(out + "/pmc_traffic.json", "w"), indent=1)
for f in sorted(glob.glob(out + "/**/*kernel_stats.csv", recursive=True)):
    print("==", f); print(open(f).read()[:3000])
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "?")[:60]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
    print("==", f)
    for k, d in agg.items():
        for c, v in d.items():
            print("  %-60s %-24s sum=%.6g launches=%d" % (k, c, v, cnt[(k, c)]))
PY
