#!/usr/bin/env python3
"""Does the regions leg run slower in a process that has run the BAM pipeline legs before it?  (bench.py's order)
Usage: python tools/regions_after.py [bench.py arguments]"""
import ctypes
import gc
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

args = bench.parse_args(sys.argv[1:])
import torch  # noqa: E402
from graphtyper_amd import lib as gtx, synth  # noqa: E402

device = torch.device("cuda", 0)
ref, records, ref_str = bench.cfg2_graph_inputs(synth, args.region_len, args.snp_every)


def threads():
    return [l.split()[1] for l in open("/proc/self/status") if l.startswith("Threads")][0]


def regions(tag):
    r = bench.extra_regions(args, torch, gtx, synth, device, ref)
    best = max(v["regions_per_s"] for v in r["inside_the_library_gtx_regions_run"].values())
    print(tag, "regions/s", round(r["regions_per_s"]), "best in-library", round(best), "one after the other", round(r["one_after_the_other"]["regions_per_s"]),
          r["one_after_the_other"]["stage_s"], "threads", threads(), flush=True)


regions("fresh process          ")
ctx = gtx.Context(gtx.graph_from_records(ref_str, records, region_begin=bench.REGION_BEGIN), device=0)
p = bench.extra_pipeline(args, torch, gtx, synth, device, ctx, ref, records)
print("pipeline", round(p["reads_per_s"] / 1e6, 1), "M reads/s; threads", threads(), flush=True)
regions("after the pipeline leg ")
gc.collect()
ctypes.CDLL("libc.so.6").malloc_trim(0)
gtx.lib().gtx_device_cache_release()
regions("after trim + cache drop")
