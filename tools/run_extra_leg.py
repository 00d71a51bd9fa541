#!/usr/bin/env python3
"""One `extra` leg of bench.py on its own (for kernel traces: rocprofv3 --kernel-trace --stats -- python tools/run_extra_leg.py repeats).
Usage: python tools/run_extra_leg.py {repeats|regions|genome_like|long_reads|cfg5|cfg3|clusters|pipeline|pipeline64|shrink} [bench.py arguments]"""
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
elif leg in ("pipeline", "pipeline64"):  # BAM files -> VCF text on 16 / 64 host threads (GTX_BGZF_THREADS / GTX_BGZF_LINGER_US: the inflate team)
    ctx = gtx.Context(gtx.graph_from_records(ref_str, records, region_begin=bench.REGION_BEGIN), device=0)
    out = bench.extra_pipeline(args, torch, gtx, synth, device, ctx, ref, records, **(dict(n=16_000_000, threads=64) if leg == "pipeline64" else {}))
    out.pop("what", None)
elif leg == "shrink":
    out = bench.extra_shrink(args, torch, gtx, synth, device, ref, records)
elif leg == "regions":
    out = bench.extra_regions(args, torch, gtx, synth, device, ref)
elif leg == "genome_like":
    out = bench.extra_genome_like(args, torch, gtx, synth, device)
elif leg == "cfg5":
    out = bench.extra_cfg5(args, torch, gtx, synth, device)
else:
    out = bench.extra_cfg3(args, torch, gtx, synth, device, ref, leg)
print(json.dumps(out))
