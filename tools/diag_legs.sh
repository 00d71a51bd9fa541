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
for leg in genome_like repeats cfg3; do
 GTX_LIB=libgtx_profw.so GTX_BENCH_REPEATS_LANES=1 timeout 280 python tools/run_extra_leg.py $leg --no-cpu-baseline > gpurun_out/profw1_$leg.json 2> gpurun_out/profw1_$leg.txt
 grep -A12 "phase cycles per task" gpurun_out/profw1_$leg.txt | head -14
done
