"""The same step many times over, every time on streams the process has never used: one digest.

Round 4 shipped a race for most of the round -- scratch counters and accumulator blocks were zeroed with hipMemset, which returns
before the fill has run on the null stream, and kernels on non-blocking streams are not ordered behind it: a wrong VCF text once
in ~25 fresh processes, found by luck.  This is the test that would have found it: 50 repetitions of align (all passes) -> score ->
calls on a graph with enough variant sites that every pass and the scorer have work, half of them with three steps in flight on
two streams (bench.py's staggered schedule), each repetition with new streams, events, record slots and accumulator blocks; the
records, the accumulator block and the calls of every repetition must be the same bytes."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPEATS = int(os.environ.get("GTX_DETERMINISM_REPEATS", "50"))


def test_fifty_steps_on_fresh_streams_one_digest():
    import torch
    import bench
    from graphtyper_amd import lib as gtx
    from graphtyper_amd import synth
    assert torch.cuda.is_available() and os.path.exists(gtx.LIB_PATH)
    device = torch.device("cuda", 0)
    region_len, n = 200000, 400000
    ref = synth.make_reference(region_len, seed=42)
    records = synth.make_snp_records(ref, 100, seed=7, region_begin=bench.REGION_BEGIN)  # a SNP every 100 bp: every read carries sites
    ctx = gtx.Context(gtx.graph_from_records(synth.bases_to_str(ref), records, region_begin=bench.REGION_BEGIN), device=0)
    d_seq, d_pos = bench.make_reads_on_device(torch, ref, records, n, seed=99, device=device, REGION_LEN=region_len, err_rate=0.01, n_rate=0.002)
    samples = np.random.default_rng(5).integers(0, 4, size=n).astype(np.uint32)
    digests = set()
    shares = []
    for rep in range(REPEATS):
        lanes = 3 if rep % 2 else 1
        w = bench.Workload(torch, gtx, ctx, device, d_seq, d_pos, 4, samples=samples, lanes=lanes)
        if lanes == 3:
            w.steps_staggered(3)  # three steps in flight: the last one's results are lane 2's
            ln = w.lanes[2]
        else:
            w.step(0)
            ln = w.lanes[0]
        torch.cuda.synchronize()
        h = hashlib.sha256()
        rec = ln["d_rec"].cpu().numpy().view(np.uint32)
        if ln["d_compact"] is not None:
            rec = gtx.merge_compact(rec, ln["d_compact"].cpu().numpy().view(np.uint32), ln["d_flags"].cpu().numpy(), n, bench.REC_WORDS)
        h.update(rec.tobytes())
        h.update(gtx.download(ln["buf"].d_stat_u64, np.uint8, w.reduced_bytes).tobytes())
        h.update(ln["d_calls"].cpu().numpy().tobytes())
        h.update(ln["d_phred"].cpu().numpy().tobytes())
        digests.add(h.hexdigest())
        if rep == 0:
            head = rec.reshape(2 * n, bench.REC_WORDS)[0::2, 0]
            assert ((head & 0xFFFF) > 0).mean() > 0.95 and not ((head >> 16) & gtx.ST_ERROR_MASK).any()
            k = ctx.kernel_times()
            shares = [t for _, _, t in k]
        w.close()
        del w
        assert len(digests) == 1, "repetition %d differs from the ones before it" % rep
    assert len(digests) == 1
