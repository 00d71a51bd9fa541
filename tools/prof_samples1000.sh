set -u
OUT=$PWD/gpurun_out/prof1000; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp; REPO=$PWD; cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o bench -- python $REPO/bench.py --samples 1000 --steps 6 --warmup 1 --lanes 1 --no-cpu-baseline --no-extra > $OUT/bench.log 2>&1
cd $REPO
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
cut -d, -f1-4 $f | head -16 | cut -c1-150
find $OUT -type f -size +1M -delete
