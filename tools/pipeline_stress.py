#!/usr/bin/env python3
"""The BAM files -> VCF text leg of bench.py (extra_pipeline) over and over on one context: does the text always equal the
resident run's?  Usage: python tools/pipeline_stress.py [rounds] [reads] [threads]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 16
args = bench.parse_args(["--no-cpu-baseline"])
import torch  # noqa: E402
from graphtyper_amd import lib as gtx, synth  # noqa: E402

device = torch.device("cuda", 0)
ref, records, ref_str = bench.cfg2_graph_inputs(synth, args.region_len, args.snp_every)
ctx = gtx.Context(gtx.graph_from_records(ref_str, records, region_begin=bench.REGION_BEGIN), device=0)
bad = 0
for r in range(rounds):
    out = bench.extra_pipeline(args, torch, gtx, synth, device, ctx, ref, records, n=n, threads=threads)
    ok = out.get("vcf_equals_resident_run")
    bad += not ok
    print(r, "ok" if ok else "DIFFERENT", "%.1f M reads/s" % (out.get("reads_per_s", 0) / 1e6), json.dumps({k: out[k] for k in out if k.startswith("vcf_") and k != "vcf_bytes"}) if not ok else "", flush=True)
print("rounds", rounds, "different", bad)
