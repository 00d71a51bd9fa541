#!/usr/bin/env python3
"""Host time of gtx_ctx_create's flat graph + index build for the cfg2 graph, by size of the thread team
(GTX_HOST_THREADS).  No GPU needed: python tools/ctx_time.py"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1:
    from graphtyper_amd import lib as gtx, synth
    ref = synth.make_reference(1000000, seed=42)
    recs = synth.make_snp_records(ref, 1000, seed=7, region_begin=1000000)
    g = gtx.graph_from_records(synth.bases_to_str(ref), recs, region_begin=1000000)
    best = 1e9
    for _ in range(3):
        t = time.time()
        c = gtx.Context(g, device=-1)
        best = min(best, time.time() - t)
        c.close()
    print("threads %s: ctx (host) %.3f s" % (sys.argv[1], best))
else:
    print("hardware threads:", os.cpu_count())
    for t in ("1", "4", "8", "16", "32"):
        subprocess.call([sys.executable, __file__, t], env=dict(os.environ, GTX_HOST_THREADS=t))
