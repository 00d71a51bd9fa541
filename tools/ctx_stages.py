#!/usr/bin/env python3
"""Stage times of gtx_ctx_create on the device (GTX_TIMING=1) for the cfg2 graph and the cfg3-like merged-cluster graph."""
import os
import sys
import time

os.environ["GTX_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtyper_amd import lib as gtx, synth  # noqa: E402

ref = synth.make_reference(1000000, seed=42)
for name, recs, kw in (("cfg2", synth.make_snp_records(ref, 1000, seed=7, region_begin=1000000), {}),
                       ("cfg3", synth.make_cluster_records(ref, 150, seed=13, region_begin=1000000), dict(add_all_variants=True))):
    t = time.time()
    g = gtx.graph_from_records(synth.bases_to_str(ref), recs, region_begin=1000000, **kw)
    print("%s: graph build %.3f s" % (name, time.time() - t), flush=True)
    for k in range(2):
        t = time.time()
        c = gtx.Context(g, device=0)
        print("%s: ctx_create #%d %.3f s, keys %d labels %d" % ((name, k, time.time() - t) + c.index_stats()), flush=True)
        c.close()
