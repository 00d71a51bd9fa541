python bench.py --no-cpu-baseline --steps 10 --warmup 2 > gpurun_out/b3.json 2> gpurun_out/b3.err; echo exit $?; tail -3 gpurun_out/b3.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/b3.json') if l.startswith('{')][-1])
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel_ms'], j['config']['calls_checksum'].get('matches_pinned'))
print(json.dumps(j['config']['extra']['repeats']))
PY
