#!/bin/bash
# Diagnostics of the repeat-rich legs on one box: the pass shares and pass times one step at a time, with the lean and the dense build of
# the position-hinted pass; then -- if graphtyper_amd/libgtx_profw.so exists (make -C graphtyper_amd/csrc OBJDIR=build_profw
# OUT=../libgtx_profw.so EXTRA="-DGTX_PROF -DGTX_PROF_WALK") -- the general pass' phase cycles with the walks in parts and why tasks leave it.
#   gpurun -- 'bash tools/diag_legs.sh'
export GTX_BENCH_FULL_LINE=1
mkdir -p gpurun_out
for leg in genome_like repeats; do
 for hb in lean dense; do
  GTX_HINT_BUILD=$hb GTX_BENCH_REPEATS_LANES=1 timeout 280 python tools/run_extra_leg.py $leg --no-cpu-baseline > gpurun_out/diag_${leg}_$hb.json 2> gpurun_out/diag_${leg}_$hb.txt
  python - <<PY
import json
j=json.load(open("gpurun_out/diag_${leg}_$hb.json"))
print("$leg $hb", {k:j.get(k) for k in ("ms_per_step","align_passes_ms","pass_shares","exact_pass")})
print("   ", j.get("align_kernels"))
PY
 done
done
[ -f graphtyper_amd/libgtx_profw.so ] || exit 0
for leg in genome_like repeats cfg3; do
 GTX_LIB=libgtx_profw.so GTX_BENCH_REPEATS_LANES=1 timeout 280 python tools/run_extra_leg.py $leg --no-cpu-baseline > gpurun_out/profw1_$leg.json 2> gpurun_out/profw1_$leg.txt
 grep -A12 "phase cycles per task" gpurun_out/profw1_$leg.txt | head -14
done
