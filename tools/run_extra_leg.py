#!/usr/bin/env python3
"""One `extra` leg of bench.py on its own (for kernel traces: rocprofv3 --kernel-trace --stats -- python tools/run_extra_leg.py repeats).
Usage: python tools/run_extra_leg.py {repeats|long_reads|cfg5|cfg3|clusters} [bench.py arguments]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

leg = sys.argv[1]
args = bench.parse_args(sys.argv[2:])
import torch  # noqa: E402
from graphtyper_amd import lib as gtx, synth  # noqa: E402

device = torch.device("cuda", 0)
ref, records, ref_str = bench.cfg2_graph_inputs(synth, args.region_len, args.snp_every)
if leg == "repeats":
    out = bench.extra_repeats(args, torch, gtx, synth, device, ref)
elif leg == "long_reads":
    out = bench.extra_long_reads(args, torch, gtx, synth, device, ref)
elif leg == "cfg5":
    out = bench.extra_cfg5(args, torch, gtx, synth, device)
else:
    out = bench.extra_cfg3(args, torch, gtx, synth, device, ref, leg)
print(json.dumps(out))
