"""BASELINE.json configurations beyond cfg2 on the device: cfg4 (1 000 samples jointly; the exchange step of the N-GPU
run through gtx_scores_reduce on a one-rank communicator), each against the oracle at a size it finishes in seconds and at
full size through size-independent properties."""
import ctypes as C
import os

import numpy as np
import pytest

import harness
import scenarios
from graphtyper_amd import lib as gtx
from graphtyper_amd import synth
from oracle_lib import Oracle
from test_emu_parity import run_stream

pytestmark = pytest.mark.gpu
REC_WORDS = harness.REC_WORDS


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    assert os.path.exists(gtx.LIB_PATH), "libgtx.so must be built (HIP path, no fallback)"


def test_cfg4_thousand_samples_vs_oracle():
    """cfg4 at reduced size: 1 000 samples in one position-sorted stream of pairs, duplicates, low-MAPQ and filtered
    records -> accumulators, SampleCalls and phasing flags of every (site, sample) == oracle.  A workgroup of the scoring
    kernel sees up to 256 different samples here: its LDS combiner overflows into direct atomics (same sums)."""
    ref, recs, codes, rec = scenarios.paired_case("snp100", n_ref=30000, n_pairs=12000, region_begin=310000, n_samples=1000)
    assert len(np.unique(rec["sample"])) > 990
    o = Oracle(ref, recs, region_begin=310000)
    b = harness.GpuBackend(gtx.graph_from_records(ref, recs, region_begin=310000))
    want = run_stream(b, o, codes, rec, n_samples=1000)
    assert want.sum() > 0


def _packed(ctx, n_samples, conn_cap=1 << 20):
    buf = gtx.ScoreBuffers()
    reduced = C.c_uint64()
    gtx.check(gtx.lib().gtx_scores_alloc(ctx.h, n_samples, conn_cap, C.byref(buf), C.byref(reduced)))
    return buf, int(reduced.value)


def _download(ctx, buf, n_samples):
    acc = harness.Accumulators(ctx, n_samples, conn_cap=buf.conn_cap)
    acc.log_score[...] = gtx.download(buf.d_log_score, np.uint32, len(acc.log_score))
    acc.gt_cov[...] = gtx.download(buf.d_gt_cov, np.uint32, len(acc.gt_cov))
    acc.hap_u32[...] = gtx.download(buf.d_hap_u32, np.uint32, len(acc.hap_u32))
    acc.stat_u64[...] = gtx.download(buf.d_stat_u64, np.uint64, len(acc.stat_u64))
    acc.stat_u32[...] = gtx.download(buf.d_stat_u32, np.uint32, len(acc.stat_u32))
    acc.conn_near[...] = gtx.download(buf.d_conn_near, np.uint32, len(acc.conn_near))
    acc.conn_count[...] = gtx.download(buf.d_conn_count, np.uint32, 2)
    acc.conn_log[...] = gtx.download(buf.d_conn_log, np.uint32, len(acc.conn_log))
    return acc


def test_packed_accumulators_and_reduce_on_one_rank():
    """gtx_scores_alloc / gtx_scores_zero / gtx_scores_reduce: the packed block gives the same accumulators as separately
    allocated arrays, an all-reduce over a one-rank RCCL communicator (made through gtx_comm_*) leaves them unchanged --
    the code path of the 8-GPU run with the world cut down to what this box has -- and zero clears them"""
    import torch
    L = gtx.lib()
    ref, recs, codes, rec = scenarios.paired_case("snp25", n_ref=20000, n_pairs=3000, region_begin=0, n_samples=5)
    b = harness.GpuBackend(gtx.graph_from_records(ref, recs))
    st = gtx.Stream(b.ctx.params, 1)
    a_seq, a_meta, items = st.push(rec, gtx.pack_nibbles(codes))
    records = b.align(a_seq, a_meta)
    want = harness.canonical_scores(b.ctx, b.score(items, records, 5))
    buf, reduced = _packed(b.ctx, 5)
    assert reduced == 8 * (b.ctx.n_hap + 2 * b.ctx.total_allele) + 4 * (5 * (b.ctx.total_tri + b.ctx.total_allele + 4 * b.ctx.n_hap + b.ctx.total_near) + b.ctx.n_hap + 6 * b.ctx.total_allele)
    d_items, d_rec = b._dev(items), b._dev(records)
    gtx.check(L.gtx_score_batch(b.ctx.h, d_items.data_ptr(), len(items), d_rec.data_ptr(), REC_WORDS, C.byref(buf), None))
    torch.cuda.synchronize()
    assert np.array_equal(harness.canonical_scores(b.ctx, _download(b.ctx, buf, 5)), want)
    # a view torch can reduce (bench.py's fallback) sees the same memory
    import bench
    t = torch.as_tensor(bench.DevView(buf.d_log_score, 5 * b.ctx.total_tri, "<i4"), device="cuda:0")
    assert np.array_equal(t.cpu().numpy().view(np.uint32), gtx.download(buf.d_log_score, np.uint32, 5 * b.ctx.total_tri))
    ident = (C.c_uint8 * 128)()
    gtx.check(L.gtx_comm_unique_id(ident))
    comm = C.c_void_p()
    gtx.check(L.gtx_comm_init_rank(ident, 1, 0, 0, C.byref(comm)))
    gtx.check(L.gtx_scores_reduce(b.ctx.h, C.byref(buf), comm, None))
    torch.cuda.synchronize()
    assert np.array_equal(harness.canonical_scores(b.ctx, _download(b.ctx, buf, 5)), want)
    # separately allocated arrays take the per-array route of the same group
    acc = harness.Accumulators(b.ctx, 5)
    devs = [b._dev(a) for a in acc.arrays()]
    loose = acc.buffers([d.data_ptr() for d in devs])
    gtx.check(L.gtx_score_batch(b.ctx.h, d_items.data_ptr(), len(items), d_rec.data_ptr(), REC_WORDS, C.byref(loose), None))
    gtx.check(L.gtx_scores_reduce(b.ctx.h, C.byref(loose), comm, None))
    torch.cuda.synchronize()
    for host, dev in zip(acc.arrays(), devs):
        host[...] = dev.cpu().numpy().view(host.dtype)
    assert np.array_equal(harness.canonical_scores(b.ctx, acc), want)
    gtx.check(L.gtx_comm_destroy(comm))
    gtx.check(L.gtx_scores_zero(b.ctx.h, C.byref(buf), None))
    torch.cuda.synchronize()
    z = _download(b.ctx, buf, 5)
    assert not any(a.any() for a in (z.log_score, z.gt_cov, z.hap_u32, z.stat_u64, z.stat_u32, z.conn_near, z.conn_count))
    gtx.check(L.gtx_scores_free(b.ctx.h, C.byref(buf)))
    assert not buf.d_stat_u64


def test_cfg4_full_size_properties():
    """cfg4 at its full size -- 1 000 samples x 80 000 reads over the 1 Mb SNP graph, in 8 batches of 10 M -- through a
    size-independent property: every effect of a read is an addition to counters of its own sample, so summing the
    1 000-sample accumulators over the samples must give exactly what the same reads give when they all belong to one
    sample (unsaturated u32 sums); and no record may carry an overflow status."""
    import torch
    import bench
    device = torch.device("cuda", 0)
    L = gtx.lib()
    n_samples, per_batch, n_batches = 1000, 10_000_000, 8
    ref = synth.make_reference(bench.REGION_LEN, seed=42)
    records = synth.make_snp_records(ref, 1000, seed=7, region_begin=bench.REGION_BEGIN)
    ctx = gtx.Context(gtx.graph_from_records(synth.bases_to_str(ref), records, region_begin=bench.REGION_BEGIN), device=0)
    joint, _ = _packed(ctx, n_samples, 1 << 22)
    single, _ = _packed(ctx, 1, 1 << 22)
    d_rec = torch.empty(per_batch * 2 * REC_WORDS, dtype=torch.int32, device=device)
    meta = np.zeros(per_batch, gtx.READ_META)
    meta["l_qseq"] = bench.READ_LEN
    items = np.zeros(per_batch, gtx.SCORE_ITEM)
    items["first"]["align_index"] = np.arange(per_batch, dtype=np.uint32)
    items["first"]["mapq"] = 60
    items["first"]["flag"] = gtx.FLAG_FORWARD_ONLY
    items["second"]["align_index"] = gtx.INVALID_ID
    n_aligned = 0
    for k in range(n_batches):
        d_seq, d_pos = bench.make_reads_on_device(torch, ref, records, per_batch, seed=500 + k, device=device)
        pos = d_pos.cpu().numpy().astype(np.int32)
        meta["pos"] = pos
        d_meta = torch.from_numpy(meta.view(np.uint8).reshape(per_batch, -1).copy()).to(device)
        gtx.check(L.gtx_align_batch(ctx.h, d_seq.data_ptr(), 80, d_meta.data_ptr(), per_batch, d_rec.data_ptr(), REC_WORDS, None))
        heads = d_rec.view(per_batch * 2, REC_WORDS)[:, 0]
        assert int((((heads >> 16) & gtx.ST_ERROR_MASK) != 0).sum().item()) == 0
        n_aligned += int(((heads[0::2] & 0xFFFF) > 0).sum().item())
        items["first"]["pos"] = pos
        for buf, samples in ((joint, np.random.default_rng(k).integers(0, n_samples, size=per_batch)), (single, 0)):
            items["sample"] = samples
            d_items = torch.from_numpy(items.view(np.uint8).reshape(per_batch, -1).copy()).to(device)
            gtx.check(L.gtx_score_batch(ctx.h, d_items.data_ptr(), per_batch, d_rec.data_ptr(), REC_WORDS, C.byref(buf), None))
            torch.cuda.synchronize()
        del d_seq, d_pos, d_meta
    assert n_aligned > 0.97 * per_batch * n_batches and ctx.error_count() == 0
    a, s = _download(ctx, joint, n_samples), _download(ctx, single, 1)
    assert s.log_score.sum() > 0 and s.gt_cov.sum() > 1000
    assert np.array_equal(a.log_score.reshape(n_samples, -1).sum(0, dtype=np.uint64), s.log_score.astype(np.uint64))
    assert np.array_equal(a.gt_cov.reshape(n_samples, -1).sum(0, dtype=np.uint64), s.gt_cov.astype(np.uint64))
    assert np.array_equal(a.hap_u32.reshape(n_samples, -1).sum(0, dtype=np.uint64), s.hap_u32.astype(np.uint64))
    assert np.array_equal(a.conn_near.reshape(n_samples, -1).sum(0, dtype=np.uint64), s.conn_near.astype(np.uint64))
    assert np.array_equal(a.stat_u64, s.stat_u64) and np.array_equal(a.stat_u32, s.stat_u32)  # per site, not per sample
    # every sample got its share (80 000 reads each, ~12x): calls exist for all of them
    phred = torch.zeros(n_samples * ctx.total_tri, dtype=torch.uint8, device=device)
    calls = torch.zeros(n_samples * ctx.n_hap * gtx.SAMPLE_CALL.itemsize, dtype=torch.uint8, device=device)
    gtx.check(L.gtx_calls_batch(ctx.h, C.byref(joint), phred.data_ptr(), calls.data_ptr(), None))
    torch.cuda.synchronize()
    c = calls.cpu().numpy().view(gtx.SAMPLE_CALL).reshape(n_samples, ctx.n_hap)
    depth = c["ref_total_depth"].astype(np.int64) + c["alt_total_depth"]
    assert (depth.sum(1) > 0).all() and (c["gt_second"] > 0).any(1).all()
    for buf in (joint, single):
        gtx.check(L.gtx_scores_free(ctx.h, C.byref(buf)))


def test_two_host_threads_share_one_context():
    """re-entrancy of the boundary (the reference calls align_read / update_haplotype_scores_geno from `jobs` worker
    threads against one index + graph, src/typer/caller.cpp:399-436): two host threads, each with its own stream and
    buffers, call gtx_align_batch + gtx_score_batch on ONE context at the same time, several rounds; every result must
    equal what one thread alone gets"""
    import threading
    import torch
    L = gtx.lib()
    b = harness.GpuBackend(gtx.graph_from_records(*scenarios.synthetic_case("snp100", n_ref=200000, n_reads=10, region_begin=0)[:2]))
    jobs = []
    for t in range(2):
        ref, recs, codes, pos = scenarios.synthetic_case("snp100", n_ref=200000, n_reads=150000, region_begin=0, seed=0)
        codes, pos = codes[t::2], pos[t::2]  # (same graph, different reads per thread)
        order = np.argsort(pos, kind="stable")
        rec = scenarios.stream_records(len(codes), pos[order], sample=np.arange(len(codes)) % 3)
        st = gtx.Stream(b.ctx.params, 1)
        a_seq, a_meta, items = st.push(rec, gtx.pack_nibbles(codes[order]))
        records = b.align(a_seq, a_meta).copy()
        want = harness.canonical_scores(b.ctx, b.score(items, records, 3))
        jobs.append((a_seq, a_meta, items, records, want))
    results, errors = [[] for _ in jobs], []

    def work(t):
        try:
            a_seq, a_meta, items, _, _ = jobs[t]
            stream = torch.cuda.Stream(device="cuda:0")
            sp = C.c_void_p(stream.cuda_stream)
            d_seq, d_meta, d_items = b._dev(a_seq), b._dev(a_meta), b._dev(items)
            d_rec = torch.zeros(len(a_meta) * 2 * REC_WORDS, dtype=torch.int32, device="cuda:0")
            torch.cuda.synchronize()
            for _ in range(6):
                buf, _ = _packed(b.ctx, 3)
                d_rec.zero_()
                torch.cuda.current_stream().synchronize()
                gtx.check(L.gtx_align_batch(b.ctx.h, d_seq.data_ptr(), a_seq.shape[1], d_meta.data_ptr(), len(a_meta), d_rec.data_ptr(), REC_WORDS, sp))
                gtx.check(L.gtx_score_batch(b.ctx.h, d_items.data_ptr(), len(items), d_rec.data_ptr(), REC_WORDS, C.byref(buf), sp))
                stream.synchronize()
                results[t].append((d_rec.cpu().numpy().view(np.uint32).copy(), harness.canonical_scores(b.ctx, _download(b.ctx, buf, 3))))
                gtx.check(L.gtx_scores_free(b.ctx.h, C.byref(buf)))
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append(repr(e))

    team = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in team:
        th.start()
    for th in team:
        th.join()
    assert not errors, errors
    for t, (_, _, _, records, want) in enumerate(jobs):
        assert len(results[t]) == 6
        for rec, scores in results[t]:
            assert np.array_equal(rec, records), "thread %d: alignment records differ from the single-thread run" % t
            assert np.array_equal(scores, want), "thread %d: accumulators differ from the single-thread run" % t
    assert b.ctx.error_count() == 0


def test_cfg5_sv_graph(tmp_path):
    """BASELINE cfg5 on the device: `genotype_sv`, 100 samples, a 1 Mb SV-augmented graph (100 <DEL> of 50..5 000 bp, 50 <INS>
    with 152-bp breakpoint alleles, built from FASTA + VCF by gtx_graph_from_files), FR pairs over every breakpoint plus
    background -> SV stream logic -> align -> score -> calls == oracle"""
    from test_emu_parity import cfg5_case
    on_sv = cfg5_case(harness.GpuBackend, tmp_path, n_ref=1_000_000, n_del=100, n_ins=50, n_samples=100, pairs_per_sv=160,
                      background_pairs=12000)
    assert on_sv > 2000


@pytest.mark.parametrize("case", ["chr1", "chr2", "chr3", "chr4", "chr9", "chr10", "chr11", "snp100", "indel", "dense", "cluster", "sv"])
def test_device_built_index_equals_the_oracles(case):
    """the index is built on the device (gtx_index_dev.hip: sort, grouping, hash tables, hint tables; the host only
    enumerates the k-mers): keys, label counts and the labels in bucket order, fetched back through gtx_index_dump, must equal
    the oracle's index_graph on the scenarios of tests/test_host_parity.py -- whose oracle side is pinned on the reference's
    test/index/test_index.cpp -- and gtx_index_get must find what PHIndex::get finds"""
    from fixtures import contig, sv_contig
    kw, okw = {}, {}
    rb = 0
    if case.startswith("chr"):
        ref, recs = contig(case)
    elif case == "snp100":
        rb = 1000000
        r = synth.make_reference(120000, seed=5)
        ref, recs = synth.bases_to_str(r), synth.make_snp_records(r, every=100, seed=9, region_begin=rb)
    elif case == "indel":
        rb = 5000
        r = synth.make_reference(60000, seed=6)
        ref, recs = synth.bases_to_str(r), synth.make_indel_records(r, every=37, seed=3, region_begin=rb)
    elif case == "dense":
        r = synth.make_reference(900, seed=8)
        ref, recs = synth.bases_to_str(r), synth.make_snp_records(r, every=3, seed=2, first=40)
    elif case == "cluster":
        ref, recs, _, _ = scenarios.synthetic_case("cluster", n_ref=60000, n_reads=1, region_begin=0)
        kw, okw = dict(add_all_variants=True), dict(add_all_variants=True)
    else:
        ref, recs = sv_contig("chr5")
        kw, okw = dict(is_sv_graph=True), dict(is_sv_graph=True)
    o = Oracle(ref, recs, region_begin=rb, **okw)
    g = gtx.graph_from_records(ref, recs, region_begin=rb, **kw)
    c = gtx.Context(g, device=0, is_sv_graph=(case == "sv"))
    k1, c1, l1 = o.index_dump()
    assert c.index_stats() == (len(k1), len(l1))
    k2, c2, l2 = c.index_dump()
    assert np.array_equal(k1, k2) and np.array_equal(c1, c2) and np.array_equal(l1, l2)
    rng = np.random.default_rng(1)
    for k in rng.choice(len(k1), size=min(len(k1), 200), replace=False):
        assert c.index_get(int(k1[k])) == o.index_get(int(k1[k]))
    assert c.index_get(int(k1[0]) ^ 0x5555) == o.index_get(int(k1[0]) ^ 0x5555)
    # and the host build (contexts without a device, the test emulation) gives the same
    h = gtx.Context(g, device=-1, is_sv_graph=(case == "sv"))
    k3, c3, l3 = h.index_dump()
    assert np.array_equal(k1, k3) and np.array_equal(c1, c3) and np.array_equal(l1, l3)
    # the tables of the position-hinted pass: the device's build == the host's (gtx_ctx_hint_table)
    for which in range(5):
        assert np.array_equal(c.hint_table(which), h.hint_table(which)), "hint table %d" % which


def test_sites_with_more_than_64_alleles_on_the_device():
    """a 100-allele site and a merged cluster of > 1000 alleles: gtx_align_wide_kernel / gtx_score_wide_kernel == oracle
    (alignment records with every kind of hint, scores, calls over the half-million-entry genotype triangle, VCF text)"""
    from test_wide_sites import wide_sites_case
    b = wide_sites_case(harness.GpuBackend)
    assert b.ctx.n_hap > 10


def test_saturation_guard_is_replayed_across_two_ranks_on_the_device():
    """gtx_scores_replay_log / gtx_scores_replay_apply: the reads sharded over two ranks' worth of blocks, the guard reached only in
    their sum, the logs of both halves replayed in stream order == the oracle over all reads"""
    from test_saturation import two_rank_replay_case
    two_rank_replay_case(harness.GpuBackend)


def test_saturation_guard_is_replayed_on_the_device():
    """gtx_scores_replay: 10 500 reads of one sample over one SNP drive max_log_score past 0xFFFF; the replayed cell equals
    the oracle's sequential explain_to_score (haplotype.cpp:560)"""
    from test_saturation import saturation_case
    saturation_case(harness.GpuBackend)


def test_bam_files_to_vcf_text_on_the_device(tmp_path):
    """BAM files -> gtx_reads -> gtx_stream -> gtx_align_batch -> gtx_score_batch -> gtx_calls_batch -> gtx_vcf_records == oracle"""
    from test_bam_ingest import bam_to_calls_case
    bam_to_calls_case(harness.GpuBackend, tmp_path)


@pytest.mark.parametrize("threads,chunk", [(2, 256), (1, 100), (2, 65536)])
def test_the_librarys_own_host_loop(tmp_path, threads, chunk):
    """gtx_pipeline_run (host threads, staging, copies and launches inside the library) over the two BAM files of the case above ==
    the oracle's VCF text; with one thread both files are one merged stream, with small chunks mates arrive batches apart; twice
    on one context (the streams of the first run are the second's)"""
    import ctypes as C
    import torch
    from test_bam_ingest import bam_to_calls_case
    import scenarios
    want = bam_to_calls_case(harness.GpuBackend, tmp_path)
    rb = 310000
    ref, recs, _, rec = scenarios.paired_case("snp100", n_ref=12000, n_pairs=600, region_begin=rb, n_samples=2)
    ctx = gtx.Context(gtx.graph_from_records(ref, recs, region_begin=rb), device=0)
    L = gtx.lib()
    paths = [str(tmp_path / "SAMP0.bam"), str(tmp_path / "SAMP1.bam")]
    for again in range(2):
        buf = gtx.ScoreBuffers()
        gtx.check(L.gtx_scores_alloc(ctx.h, 2, 1 << 16, C.byref(buf), None))
        st = gtx.pipeline_run(ctx, paths, threads, buf, harness.REC_WORDS, len(rec), chunk=chunk, region="chr7")
        assert st["records"] == len(rec) and st["n_samples"] == 2 and st["n_threads"] == threads
        d_phred = torch.zeros(max(2 * ctx.total_tri, 1), dtype=torch.uint8, device="cuda:0")
        d_calls = torch.zeros(max(2 * ctx.n_hap, 1) * gtx.SAMPLE_CALL.itemsize, dtype=torch.uint8, device="cuda:0")
        gtx.check(L.gtx_calls_batch(ctx.h, C.byref(buf), d_phred.data_ptr(), d_calls.data_ptr(), None))
        torch.cuda.synchronize()
        nh, ta = ctx.n_hap, ctx.total_allele
        text = ctx.vcf_records("chr7", ["person0", "person1"], gtx.download(buf.d_gt_cov, np.uint32, 2 * ta), gtx.download(buf.d_stat_u64, np.uint64, nh + 2 * ta),
                               gtx.download(buf.d_stat_u32, np.uint32, nh + 6 * ta), d_phred.cpu().numpy()[:2 * ctx.total_tri],
                               d_calls.cpu().numpy().view(gtx.SAMPLE_CALL)[:2 * nh])
        L.gtx_scores_free(ctx.h, C.byref(buf))
        assert text == want
    with pytest.raises(gtx.GtxError):  # fewer record slots than reads
        buf = gtx.ScoreBuffers()
        gtx.check(L.gtx_scores_alloc(ctx.h, 2, 1 << 16, C.byref(buf), None))
        try:
            gtx.pipeline_run(ctx, paths, threads, buf, harness.REC_WORDS, 10, chunk=chunk, region="chr7")
        finally:
            L.gtx_scores_free(ctx.h, C.byref(buf))
    ctx.close()
