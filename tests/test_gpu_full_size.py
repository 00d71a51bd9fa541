"""BASELINE.json configs[1] at its full size (1 sample, 10 M reads of 150 bp, 1 Mb SNP-only graph) on the GPU, checked through
properties that do not need the oracle to process 10 M reads:

  * determinism: two runs over the same batch give the same record words;
  * independence of batch composition and order (SURVEY.md 8(b) conventions): the same reads, permuted and cut into three
    uneven batches, give the same record for every read;
  * a random sample of the 10 M records equals the oracle's GenotypePaths (bit-exact) -- with position hints, i.e. records
    of the position-hinted pass (97 % of them), and the whole batch again without hints gives the same words;
  * the score accumulators are sums: scoring the whole batch == scoring its two halves one after the other into the same
    accumulators == scoring the items in reverse order, and the genotype calls derived from them agree;
  * conservation: every aligned read that overlaps a variant site adds exactly one unit of depth to that site.

The second test needs no property: it pushes ALL reads of the benchmark's own read set through the oracle (sharded over the
host's cores, summed) and compares every site's accumulators, SampleCalls and the VCF bytes.

GTX_FULL_READS overrides the read count (debugging on a small box)."""
import ctypes as C
import os

import numpy as np
import pytest

from graphtyper_amd import lib as gtx
from graphtyper_amd import synth
from oracle_lib import Oracle, sharded_genotyper

pytestmark = pytest.mark.gpu

N_READS = int(os.environ.get("GTX_FULL_READS", "10000000"))
REC_WORDS = 64
REGION_BEGIN = 1000000


def _buffers(torch, ctx, device, conn_cap=1 << 24):
    nh = ctx.n_hap
    acc = dict(log_score=torch.zeros(ctx.total_tri, dtype=torch.int32, device=device),
               gt_cov=torch.zeros(ctx.total_allele, dtype=torch.int32, device=device),
               hap_u32=torch.zeros(nh * 4, dtype=torch.int32, device=device),
               stat_u64=torch.zeros(nh + 2 * ctx.total_allele, dtype=torch.int64, device=device),
               stat_u32=torch.zeros(nh + 6 * ctx.total_allele, dtype=torch.int32, device=device),
               conn_log=torch.zeros(conn_cap * 6, dtype=torch.int32, device=device),
               conn_count=torch.zeros(2, dtype=torch.int32, device=device),
               conn_near=torch.zeros(max(ctx.total_near, 1), dtype=torch.int32, device=device))
    buf = gtx.ScoreBuffers(1, acc["log_score"].data_ptr(), acc["gt_cov"].data_ptr(), acc["hap_u32"].data_ptr(),
                           acc["stat_u64"].data_ptr(), acc["stat_u32"].data_ptr(), acc["conn_log"].data_ptr(),
                           acc["conn_count"].data_ptr(), conn_cap, acc["conn_near"].data_ptr())
    return acc, buf


def test_cfg2_full_size_properties():
    import torch
    import bench
    assert torch.cuda.is_available() and os.path.exists(gtx.LIB_PATH)
    device = torch.device("cuda", 0)
    L = gtx.lib()
    n = N_READS
    ref = synth.make_reference(bench.REGION_LEN, seed=42)
    records = synth.make_snp_records(ref, 1000, seed=7, region_begin=REGION_BEGIN)
    ref_str = synth.bases_to_str(ref)
    ctx = gtx.Context(gtx.graph_from_records(ref_str, records, region_begin=REGION_BEGIN), device=0)
    d_seq, d_pos = bench.make_reads_on_device(torch, ref, records, n, seed=99, device=device)
    # every read carries its position, as a record of a sorted BAM does (gtx_read_meta::pos): the position-hinted pass --
    # the kernel the benchmark spends its reads in -- is what the properties below and the oracle sample are about
    meta = np.zeros(n, gtx.READ_META)
    meta["l_qseq"] = bench.READ_LEN
    meta["pos"] = d_pos.cpu().numpy().astype(np.int32)
    d_meta = torch.from_numpy(meta.view(np.uint8).reshape(n, gtx.READ_META.itemsize).copy()).to(device)
    meta["pos"] = -1
    d_meta_nohint = torch.from_numpy(meta.view(np.uint8).reshape(n, gtx.READ_META.itemsize).copy()).to(device)
    del meta
    ctx.pass_times()  # arms the per-launch timing and task counts

    def align(seq, met, count, out):
        gtx.check(L.gtx_align_batch(ctx.h, seq.data_ptr(), 80, met.data_ptr(), count, out.data_ptr(), REC_WORDS, None))

    # ---- determinism
    rec = torch.zeros(n * 2 * REC_WORDS, dtype=torch.int32, device=device)
    align(d_seq, d_meta, n, rec)
    torch.cuda.synchronize()
    hinted_done = ctx.kernel_times()[0][2]
    assert hinted_done > 0.95 * n, "the position-hinted pass finished only %d of %d reads" % (hinted_done, n)
    again = torch.zeros_like(rec)
    align(d_seq, d_meta, n, again)
    torch.cuda.synchronize()
    assert torch.equal(rec, again), "two runs over the same batch differ"
    # ---- the hint decides who does the work, never the result: the same batch without positions (express / general passes)
    again.zero_()
    align(d_seq, d_meta_nohint, n, again)
    torch.cuda.synchronize()
    assert ctx.kernel_times()[0][2] == 0
    assert torch.equal(rec, again), "records depend on the position hint"
    del d_meta_nohint
    heads = rec.view(n * 2, REC_WORDS)[:, 0]
    assert int((((heads >> 16) & gtx.ST_ERROR_MASK) != 0).sum().item()) == 0, "a table overflowed at cfg2"
    n_aligned = int(((heads[0::2] & 0xFFFF) > 0).sum().item())
    assert n_aligned > 0.97 * n  # (0.5 % errors: a few reads lose every k-mer or exceed the mismatch filters)

    # ---- batch composition and order
    g = torch.Generator(device=device)
    g.manual_seed(5)
    perm = torch.randperm(n, generator=g, device=device)
    p_seq = d_seq[perm].contiguous()
    p_meta = d_meta[perm].contiguous()
    again.zero_()
    cuts = [0, 1, min(n, n // 3 + 7), n]
    rows = again.view(n, 2 * REC_WORDS)
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b > a:
            align(p_seq[a:b], p_meta[a:b], b - a, rows[a:b])
    torch.cuda.synchronize()
    assert torch.equal(rec.view(n, 2 * REC_WORDS)[perm], rows), "a read's record depends on the batch it is in"
    del again, rows, p_seq, p_meta

    # ---- a random sample against the oracle (bit-exact)
    rng = np.random.default_rng(11)
    pick = np.sort(rng.choice(n, size=min(n, 20000), replace=False))
    d_pick = torch.from_numpy(pick).to(device)
    sample_rec = rec.view(n, 2 * REC_WORDS)[d_pick].cpu().numpy().view(np.uint32).reshape(-1)
    sample_reads = bench.unpack_nibbles(d_seq[d_pick].cpu().numpy(), bench.READ_LEN)
    big, _ = ctx.big_records()
    got = gtx.parse_records(sample_rec, len(pick), REC_WORDS, ctx.hap_order, big)
    oracle = Oracle(ref_str, records, region_begin=REGION_BEGIN)
    want = oracle.align(list(sample_reads))
    for i, (a, b) in enumerate(zip(got, want)):
        for o in range(2):
            assert a[o]["status"] == 0
            assert dict(longest=a[o]["longest"], paths=a[o]["paths"]) == b[o], "read %d orientation %d" % (pick[i], o)

    # ---- scoring: accumulators are sums
    items = np.zeros(n, gtx.SCORE_ITEM)
    items["first"]["align_index"] = np.arange(n, dtype=np.uint32)
    items["first"]["mapq"] = 60
    items["first"]["flag"] = gtx.FLAG_FORWARD_ONLY
    items["first"]["pos"] = d_pos.cpu().numpy().astype(np.int32)
    items["second"]["align_index"] = gtx.INVALID_ID
    isz = gtx.SCORE_ITEM.itemsize
    d_items = torch.from_numpy(items.view(np.uint8).reshape(n, isz).copy()).to(device)

    def score(parts):
        acc, buf = _buffers(torch, ctx, device)
        for part in parts:
            gtx.check(L.gtx_score_batch(ctx.h, part.data_ptr(), part.shape[0], rec.data_ptr(), REC_WORDS, C.byref(buf), None))
        d_phred = torch.zeros(max(ctx.total_tri, 1), dtype=torch.uint8, device=device)
        d_calls = torch.zeros(max(ctx.n_hap, 1) * gtx.SAMPLE_CALL.itemsize, dtype=torch.uint8, device=device)
        gtx.check(L.gtx_calls_batch(ctx.h, C.byref(buf), d_phred.data_ptr(), d_calls.data_ptr(), None))
        torch.cuda.synchronize()
        assert ctx.error_count() == 0
        return acc, d_phred, d_calls

    whole, ph_w, ca_w = score([d_items])
    halves, ph_h, ca_h = score([d_items[: n // 2], d_items[n // 2:]])
    back, ph_b, ca_b = score([torch.flip(d_items, dims=[0]).contiguous()])
    for name in ("log_score", "gt_cov", "hap_u32", "stat_u64", "stat_u32", "conn_near"):
        assert torch.equal(whole[name], halves[name]), name + ": whole batch != its two halves"
        assert torch.equal(whole[name], back[name]), name + ": depends on the item order"
    assert torch.equal(ph_w, ph_h) and torch.equal(ca_w, ca_h) and torch.equal(ph_w, ph_b) and torch.equal(ca_w, ca_b)
    assert int(whole["conn_count"][1].item()) == 0

    # ---- conservation of depth: a read whose path covers a site supports one of its alleles (both only when the base
    #      there is N), and Haplotype::add_coverage (haplotype.cpp:180) counts a read for a site when exactly one allele is
    #      supported -- so a site's depth is the number of aligned reads covering it, minus the few that
    #      are_genotype_paths_good (vcf_writer.cpp:28) turns away.  (Reads with several paths, 0.2 % here, are counted at
    #      their first path: hence the slack.)
    gt_cov = whole["gt_cov"].cpu().numpy().astype(np.int64)
    allele_off = np.asarray(ctx.allele_off, np.int64)
    site_depth = gt_cov[allele_off] + gt_cov[allele_off + 1]
    r = rec.view(n * 2, REC_WORDS)[0::2]
    one_path = (r[:, 0] & 0xFFFF) >= 1
    start = r[:, 2].to(torch.int64) & 0xFFFFFFFF
    end = r[:, 3].to(torch.int64) & 0xFFFFFFFF
    order = torch.from_numpy(np.asarray(ctx.hap_order, np.int64)).to(device)
    s_sorted, _ = torch.sort(start[one_path])
    e_sorted, _ = torch.sort(end[one_path])
    # reads covering position p = #(start <= p) - #(end < p)
    covering = (torch.searchsorted(s_sorted, order, right=True) - torch.searchsorted(e_sorted, order, right=False)).cpu().numpy()
    inner = (np.asarray(ctx.hap_order) > REGION_BEGIN + 200) & (np.asarray(ctx.hap_order) < REGION_BEGIN + bench.REGION_LEN - 200)
    assert site_depth[inner].sum() > 0
    assert np.all(site_depth[inner] <= covering[inner] + covering[inner] // 100 + 3)
    assert site_depth[inner].sum() >= 0.97 * covering[inner].sum()


def test_cfg2_every_read_against_the_oracle():
    """The benchmark's own result at the benchmark's own size: read set 0 of bench.py (same seed, same generator) through
    bench.py's own step (gtx_align_batch_flags with position hints -> gtx_score_batch_flags -> gtx_calls_batch) and
    gtx_vcf_records; the same 10 M reads, every one of them, through the oracle on the host (contiguous shards on the host's
    threads, Genotyper::merge_from).  Equal: the canonical score stream (every accumulator of every site), the SampleCalls,
    the VCF text byte for byte -- and therefore the digest bench.py prints as config.calls_checksum."""
    import hashlib
    import time
    import torch
    import bench
    import harness
    device = torch.device("cuda", 0)
    n = N_READS
    ref, records, ref_str = bench.cfg2_graph_inputs(synth)
    ctx = gtx.Context(gtx.graph_from_records(ref_str, records, region_begin=REGION_BEGIN), device=0)
    d_seq, d_pos = bench.make_reads_on_device(torch, ref, records, n, seed=bench.CFG2_READ_SEED, device=device)
    w = bench.Workload(torch, gtx, ctx, device, d_seq, d_pos, 1, hint=True)
    ctx.pass_times()
    digest = w.calls_checksum(0)
    assert ctx.kernel_times()[0][2] > 0.95 * n, "the position-hinted pass did not carry the benchmark's reads"
    assert ctx.error_count() == 0
    text, calls = w.vcf_text()
    assert hashlib.sha256(text).hexdigest() == digest["vcf_sha256"]
    final = w.vcf_final_text()
    # the product's accumulators in the oracle's canonical form
    acc = harness.Accumulators(ctx, 1, conn_cap=1)
    nh, ta = ctx.n_hap, ctx.total_allele
    acc.log_score = gtx.download(w.buf.d_log_score, np.uint32, ctx.total_tri)
    acc.gt_cov = gtx.download(w.buf.d_gt_cov, np.uint32, ta)
    acc.hap_u32 = gtx.download(w.buf.d_hap_u32, np.uint32, nh * 4)
    acc.stat_u64 = gtx.download(w.buf.d_stat_u64, np.uint64, nh + 2 * ta)
    acc.stat_u32 = gtx.download(w.buf.d_stat_u32, np.uint32, nh + 6 * ta)
    acc.conn_count = gtx.download(w.buf.d_conn_count, np.uint32, 2)
    assert int(acc.conn_count[0]) == 0 and int(acc.conn_count[1]) == 0  # (sites 1 kb apart: no read sees two)
    acc.conn_log = np.zeros(6, np.uint32)
    acc.conn_near = gtx.download(w.buf.d_conn_near, np.uint32, ctx.total_near) if ctx.total_near else np.zeros(0, np.uint32)
    phred = w.d_phred.cpu().numpy()[:ctx.total_tri]
    got_scores = harness.canonical_scores(ctx, acc)
    got_calls = harness.canonical_calls(ctx, phred, calls, 1)
    # the oracle over every read
    codes = bench.unpack_nibbles(d_seq.cpu().numpy(), bench.READ_LEN)
    pos = d_pos.cpu().numpy()
    w.close()
    del d_seq, w
    oracle = Oracle(ref_str, records, region_begin=REGION_BEGIN)
    t0 = time.perf_counter()
    og, threads = sharded_genotyper(oracle, codes, pos)
    print("oracle: %d reads on %d host threads in %.1f s" % (n, threads, time.perf_counter() - t0))
    assert og.counts()["records"] == n
    want_scores = og.scores()
    assert len(got_scores) == len(want_scores)
    bad = np.nonzero(got_scores != want_scores)[0]
    assert len(bad) == 0, "score streams differ at words %s" % bad[:10]
    want_calls = og.calls()
    assert len(got_calls) == len(want_calls) and np.array_equal(got_calls, want_calls), "sample calls differ"
    want_text = og.vcf_records("chr20", w_names(1))
    if text != want_text:
        gl, wl = text.split(b"\n"), want_text.split(b"\n")
        first = [i for i in range(min(len(gl), len(wl))) if gl[i] != wl[i]][:1]
        raise AssertionError("VCF text differs (%d vs %d lines), first at line %s" % (len(gl), len(wl), first))
    assert hashlib.sha256(want_text).hexdigest() == digest["vcf_sha256"]
    # ... and the file genotype() ends with: vcf_merge_and_break with the variants broken down (a SNP graph: no site needs paw::Skyr)
    want_final = og.vcf_records_final("chr20", w_names(1), ref_str, REGION_BEGIN + 1)
    assert final == want_final and hashlib.sha256(final).hexdigest() == digest["final_vcf_sha256"]
    assert 0 < final.count(b"\n") - 1 <= len(records)
    # ... and the digest committed under tests/golden (bench.py reports `matches_pinned` against it) is the ORACLE's
    import json
    pin = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg2_vcf_digest.json")))
    if n == pin["reads"]:
        assert hashlib.sha256(want_text).hexdigest() == pin["vcf_sha256"] and len(want_text) == pin["vcf_bytes"]
    assert text.count(b"\n") - 1 == len(records) and (calls["gt_second"] > 0).sum() > len(records) // 4  # not vacuous


def w_names(n_samples):
    return ["SAMP%04d" % i for i in range(n_samples)]
