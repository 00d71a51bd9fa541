#!/usr/bin/env python3
"""Reads of 250 bases through the alignment passes: who finishes them and how fast (the eight-k-mer build of the position-hinted
pass, gtx_align_hinted_long_kernel).  cfg2's graph (1 Mb, SNP every 1 kb), 2 M unpaired reads, 0.5 % substitutions, 0.1 % N.
    python tools/long_reads_rate.py [read_len] [n_reads]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from graphtyper_amd import lib as gtx, synth  # noqa: E402
import harness  # noqa: E402

read_len = int(sys.argv[1]) if len(sys.argv) > 1 else 250
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
ref = synth.make_reference(1_000_000, seed=42)
recs = synth.make_snp_records(ref, 1000, seed=7, region_begin=1000000)
codes, pos = synth.make_reads(ref, recs, n, read_len=read_len, seed=5, region_begin=1000000)
order = np.argsort(pos, kind="stable")
codes, pos = codes[order], pos[order]
g = gtx.graph_from_records(synth.bases_to_str(ref), recs, region_begin=1000000)
ctx = gtx.Context(g, device=0)
ctx.pass_times()
seq = gtx.pack_nibbles(codes)
meta = harness.read_meta(np.full(n, read_len, np.uint16), pos=pos)
meta["flag"] |= 0x8000
d_seq = torch.from_numpy(seq).to("cuda:0")
d_meta = torch.from_numpy(meta.view(np.uint8).reshape(-1)).to("cuda:0")
stride = (seq.shape[1] + 15) // 16 * 16
d_planes = torch.empty((n, stride), dtype=torch.uint8, device="cuda:0")
gtx.check(gtx.lib().gtx_reads_to_planes(ctx.h, d_seq.data_ptr(), seq.shape[1], n, d_planes.data_ptr(), stride, None))
d_rec = torch.zeros(n * 2 * 64, dtype=torch.int32, device="cuda:0")
for hint in ("", "0"):
    if hint:
        os.environ["GTX_HINT"] = hint
    for _ in range(2):
        gtx.check(gtx.lib().gtx_align_batch_planes(ctx.h, d_planes.data_ptr(), stride, d_meta.data_ptr(), n, d_rec.data_ptr(), 64, None, None))
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        gtx.check(gtx.lib().gtx_align_batch_planes(ctx.h, d_planes.data_ptr(), stride, d_meta.data_ptr(), n, d_rec.data_ptr(), 64, None, None))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print("%d-base reads, %s: %.1f M reads/s aligned (%.2f ms per %d reads); kernels %s" %
          (read_len, "no position-hinted pass" if hint else "with the position-hinted pass", n / dt / 1e6, 1e3 * dt, n,
           [(k[0].replace("gtx_align_", "").replace("_kernel", ""), round(k[1], 3), k[2]) for k in ctx.kernel_times()]), flush=True)
