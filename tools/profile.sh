#!/bin/bash
# Profiling recipe used for profiles/: run on the GPU box from the repo root (gpurun -- 'bash tools/profile.sh r01').
# 1) kernel trace + stats of the bench command (default schedule: steps in flight); 2) PMC passes (own runs, no trace domains,
#    one step at a time: counters are per kernel) for instruction mix and HBM bytes.
set -u
export GTX_BENCH_FULL_LINE=1  # bench.py prints its whole record (the default line is the compact one the driver parses)
TAG=${1:-r02}
READS=${2:-10000000}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
# (the staggered schedule alone, twenty timed steps: the calibration's whole-steps schedule overlaps launches of one kernel with each other)
GTX_BENCH_CALIBRATE=0 timeout 170 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o bench -- python $REPO/bench.py --reads $READS --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench_trace.log 2>&1
timeout 170 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/pmc_sq -o bench -- python $REPO/bench.py --reads $READS --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-extra > $OUT/bench_pmc_sq.log 2>&1
timeout 170 rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq2 -o bench -- python $REPO/bench.py --reads $READS --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-extra > $OUT/bench_pmc_sq2.log 2>&1
timeout 170 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- python $REPO/bench.py --reads $READS --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-extra > $OUT/bench_pmc_fetch.log 2>&1
timeout 170 rocprofv3 --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_write -o bench -- python $REPO/bench.py --reads $READS --steps 1 --warmup 0 --lanes 1 --no-cpu-baseline --no-extra > $OUT/bench_pmc_write.log 2>&1
cd $REPO
find $OUT -name '*.csv' | head -50
for f in $OUT/*.log; do grep -m1 "^{" $f | cut -c1-400; done
python tools/profile_summary.py $OUT $READS
# only then drop what is too large to bring back
find $OUT -type f ! -name '*.csv' ! -name '*.log' ! -name '*.json' -delete
find $OUT -name '*.csv' -size +4M -delete
