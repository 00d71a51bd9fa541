#!/usr/bin/env python3
"""Stage times of gtx_ctx_create on the device (GTX_TIMING=1) for the cfg2 graph, the cfg3-like merged-cluster graph and a
50 kb region of the cfg2 graph (what a region of `graphtyper genotype` is)."""
import os
import sys
import time

os.environ["GTX_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphtyper_amd import lib as gtx, synth  # noqa: E402

ref = synth.make_reference(1000000, seed=42)
cases = (("cfg2", ref, synth.make_snp_records(ref, 1000, seed=7, region_begin=1000000), {}),
         ("cfg3", ref, synth.make_cluster_records(ref, 150, seed=13, region_begin=1000000), dict(add_all_variants=True)),
         ("50kb", ref[:50000], synth.make_snp_records(ref[:50000], 1000, seed=7, region_begin=1000000), {}))
for name, r, recs, kw in cases:
    t = time.time()
    g = gtx.graph_from_records(synth.bases_to_str(r), recs, region_begin=1000000, **kw)
    print("%s: graph build %.4f s" % (name, time.time() - t), flush=True)
    for k in range(3):
        t = time.time()
        c = gtx.Context(g, device=0)
        print("%s: ctx_create #%d %.4f s, keys %d labels %d" % ((name, k, time.time() - t) + c.index_stats()), flush=True)
        t = time.time()
        c.close()
        print("%s: ctx_destroy %.4f s" % (name, time.time() - t), flush=True)
