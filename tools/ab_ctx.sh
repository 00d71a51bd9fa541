#!/bin/bash
# gtx_ctx_create time by size of the host thread team (cfg2 and the cfg3-like graph): bash tools/ab_ctx.sh on the GPU box
export GTX_BENCH_FULL_LINE=1  # bench.py prints its whole record (the default line is the compact one the driver parses)
for t in 8 16 32 64; do
  GTX_HOST_THREADS=$t python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('threads $t: cfg2 ctx %.3f s (first %.3f), cfg3 ctx %.3f s' % (j['config']['ctx_create_s'], j['config']['ctx_create_first_s'], j['config']['extra']['cfg3']['ctx_create_s']))"
done
