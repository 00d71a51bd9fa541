#!/bin/bash
# A/B of library builds on one box: bash tools/gpu_ab.sh "<lib>:<snp-every>:<reads>[:<GTX_EXPRESS4>]" ...  (two rounds, interleaved)
set -u
mkdir -p gpurun_out/ab
for round in 1 2; do
  for spec in "$@"; do
    IFS=: read lib every reads mode <<< "$spec"
    tag="${lib%.so}_${every}_${mode:-auto}_$round"
    GTX_LIB=$lib GTX_EXPRESS4=${mode:-} timeout 300 python bench.py --reads $reads --snp-every $every --no-cpu-baseline > gpurun_out/ab/$tag.json 2> gpurun_out/ab/$tag.err
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d["roofline"]["align_kernels"]
        print("%-40s %7.1f M/s  step %6.2f  " % (f.split("/")[-1], d["value"] / 1e6, d["ms_per_step"]) +
              " ".join("%s %.3f" % (n.replace("gtx_align_", "").replace("_kernel", ""), v["ms"]) for n, v in k.items()))
    except Exception as e:
        print(f, "unreadable", e)
PY
if [ "${GTX_AB_TRACE:-}" != "" ]; then
  export TMPDIR=/tmp; R=$PWD; cd /tmp
  rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/ab/trace -o t -- python $R/bench.py --reads 4000000 --snp-every $GTX_AB_TRACE --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/ab/trace.log 2>&1
  cd $R
  find gpurun_out/ab/trace -type f ! -name '*kernel_stats.csv' -delete
  for f in $(find gpurun_out/ab/trace -name '*kernel_stats.csv'); do cut -c1-200 $f | head -14; done
fi
