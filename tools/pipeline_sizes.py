#!/usr/bin/env python3
"""The BAM files -> VCF text leg of bench.py at other sizes: python tools/pipeline_sizes.py "<threads>:<reads>" ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

args = bench.parse_args(["--no-cpu-baseline"])
import torch  # noqa: E402
from graphtyper_amd import lib as gtx, synth  # noqa: E402

device = torch.device("cuda", 0)
ref, records, ref_str = bench.cfg2_graph_inputs(synth, args.region_len, args.snp_every)
ctx = gtx.Context(gtx.graph_from_records(ref_str, records, region_begin=bench.REGION_BEGIN), device=0)
for spec in sys.argv[1:]:
    threads, n = (int(x) for x in spec.split(":"))
    for rep in range(2):
        j = bench.extra_pipeline(args, torch, gtx, synth, device, ctx, ref, records, n=n, threads=threads)
        if "error" in j and "reads_per_s" not in j:
            print(threads, n, "FAILED:", j["error"], flush=True)
            continue
        print("%d threads, %d reads: %.1f M reads/s wall %.2f loop %.2f | thread-s %s | slowest %s | equal %s" % (
            j["host_threads"], n, j["reads_per_s"] / 1e6, j["wall_s"], j["read_loop_s"], j["host_thread_seconds"], j["slowest_thread_s"], j["vcf_equals_resident_run"]), "| native:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in (j.get("native_loop") or {}).items() if k != "what"}, flush=True)
