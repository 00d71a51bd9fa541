#!/usr/bin/env python3
"""The regions leg of bench.py alone in a fresh process.  Usage: python tools/regions_only.py [bench.py arguments]"""
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
for k in range(2):
    r = bench.extra_regions(args, torch, gtx, synth, device, ref)
    print(json.dumps(r), flush=True)
