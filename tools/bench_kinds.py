#!/usr/bin/env python3
"""Throughput of the hot path on the indel-rich graph shapes of the test scenarios (cfg3-like), for DESIGN.md:
1 Mb region, 1 M reads generated on the host, one sample; align + score + calls per step.
Usage on a GPU box: python tools/bench_kinds.py [indel cluster]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from graphtyper_amd import lib as gtx
    from graphtyper_amd import synth
    device = torch.device("cuda", 0)
    L = gtx.lib()
    n, rb = 1_000_000, 1_000_000
    ref = synth.make_reference(1_000_000, seed=42)
    for kind in sys.argv[1:] or ["indel", "cluster"]:
        recs = (synth.make_indel_records(ref, 60, seed=4, region_begin=rb) if kind == "indel"
                else synth.make_cluster_records(ref, 150, seed=8, region_begin=rb))
        ctx = gtx.Context(gtx.graph_from_records(synth.bases_to_str(ref), recs, region_begin=rb, add_all_variants=(kind == "cluster")), device=0)
        codes, pos = synth.make_reads(ref, recs, n, read_len=150, seed=5, region_begin=rb, rev_frac=0.0)
        order = np.argsort(pos, kind="stable")
        codes, pos = codes[order], pos[order]
        d_seq = torch.from_numpy(gtx.pack_nibbles(codes)).to(device)
        stride = d_seq.shape[1]
        meta = np.zeros(n, gtx.READ_META)
        meta["l_qseq"] = 150
        d_meta = torch.from_numpy(meta.view(np.uint8).reshape(n, -1).copy()).to(device)
        items = np.zeros(n, gtx.SCORE_ITEM)
        items["first"]["align_index"] = np.arange(n, dtype=np.uint32)
        items["first"]["mapq"] = 60
        items["first"]["flag"] = gtx.FLAG_FORWARD_ONLY
        items["first"]["pos"] = pos.astype(np.int32)
        items["second"]["align_index"] = gtx.INVALID_ID
        d_items = torch.from_numpy(items.view(np.uint8).reshape(n, -1).copy()).to(device)
        rw = 64
        d_rec = torch.zeros(n * 2 * rw, dtype=torch.int32, device=device)
        nh, cap = ctx.n_hap, 1 << 24
        acc = [torch.zeros(max(x, 1), dtype=t, device=device) for x, t in (
            (ctx.total_tri, torch.int32), (ctx.total_allele, torch.int32), (nh * 4, torch.int32), (nh + 2 * ctx.total_allele, torch.int64),
            (nh + 6 * ctx.total_allele, torch.int32), (cap * 6, torch.int32), (2, torch.int32), (ctx.total_near, torch.int32))]
        buf = gtx.ScoreBuffers(1, *[a.data_ptr() for a in acc[:7]], cap, acc[7].data_ptr())
        d_phred = torch.zeros(max(ctx.total_tri, 1), dtype=torch.uint8, device=device)
        d_calls = torch.zeros(max(nh, 1) * gtx.SAMPLE_CALL.itemsize, dtype=torch.uint8, device=device)
        ctx.pass_times()

        def step():
            for k, a in enumerate(acc):
                if k != 5:
                    a.zero_()
            gtx.check(L.gtx_align_batch(ctx.h, d_seq.data_ptr(), stride, d_meta.data_ptr(), n, d_rec.data_ptr(), rw, None))
            gtx.check(L.gtx_score_batch(ctx.h, d_items.data_ptr(), n, d_rec.data_ptr(), rw, C.byref(buf), None))
            gtx.check(L.gtx_calls_batch(ctx.h, C.byref(buf), d_phred.data_ptr(), d_calls.data_ptr(), None))

        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        ms, handed = ctx.pass_times()
        prof = ctx.profile()
        if prof[15] > 0:  # GTX_LIB=libgtx_prof.so: cycles per phase of the general algorithm, per task that ran it
            names = ["load read", "keys + exact probes", "exact labels", "chain exact", "hamming lookup", "chain hamming",
                     "walk starts", "walk ends", "filters", "record"]
            tot = float(prof[:10].sum())
            for k, nm in enumerate(names):
                sys.stderr.write("  %-22s %10.0f  %5.1f%%\n" % (nm, prof[k] / float(prof[15]), 100.0 * prof[k] / tot))
            sys.stderr.write("  tasks %d\n" % prof[15])
        heads = d_rec.view(n * 2, rw)[:, 0]
        print(json.dumps({"kind": kind, "reads": n, "sites": int(nh), "max_alleles": int(ctx.hap_cnum.max()), "reads_per_s": n / dt,
                          "ms_per_step": 1e3 * dt, "express_ms": ms[0], "general_ms": ms[1], "hbm_tables_ms": ms[2],
                          "handed_to_general": handed, "overflowed": int((((heads >> 16) & gtx.ST_ERROR_MASK) != 0).sum().item()),
                          "score_items_refused": ctx.error_count(), "connections_dropped": int(acc[6][1].item())}))


if __name__ == "__main__":
    main()
