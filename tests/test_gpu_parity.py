"""Parity tests proper: the HIP kernels, called through libgtx's C ABI on a real MI355X, against the CPU oracle.
Bit-exact (integer / index work): alignment records == oracle GenotypePaths, score accumulators == oracle
haplotype state, for every read."""
import os

import numpy as np
import pytest

import harness
import scenarios
import ctypes as C
from graphtyper_amd import lib as gtx, synth
from oracle_lib import Oracle, encode
from test_emu_parity import check_align, five_kmer_case, cfg3_case, neardup_case, direct_probes_case, n_runs_case, three_n_case, sv_deletion_case, edge_case, express_variants_case, forced_second_pass_case, iupac_case, run_stream, second_pass_case, sv_stream_case, satellite_case, homopolymer_case, dinucleotide_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    assert os.path.exists(gtx.LIB_PATH), "libgtx.so must be built (HIP path, no fallback)"


@pytest.mark.parametrize("chrom", ["chr1", "chr2", "chr3", "chr9", "chr10", "chr11"])
def test_align_index_test_contigs(chrom):
    ref, recs, reads = scenarios.contig_reads(chrom)
    o = Oracle(ref, recs, force_both=True)
    b = harness.GpuBackend(gtx.graph_from_records(ref, recs), force_both=True)
    check_align(b, o, [encode(r) for r in reads])


@pytest.mark.parametrize("kind", ["snp1k", "snp100", "snp25", "indel"])
def test_align_synthetic(kind):
    ref, recs, codes, pos = scenarios.synthetic_case(kind, n_ref=200000, n_reads=20000, region_begin=1000000)
    o = Oracle(ref, recs, region_begin=1000000)
    b = harness.GpuBackend(gtx.graph_from_records(ref, recs, region_begin=1000000))
    check_align(b, o, list(codes), pos=pos)  # with, without and with wrong position hints: same words
    share = check_align.hinted_done / float(len(codes))
    assert share >= {"snp1k": 0.70, "snp100": 0.10, "snp25": 0.0, "indel": 0.05}[kind], share


def test_align_ragged_and_short_reads():
    ref, recs, codes, pos = scenarios.synthetic_case("snp100", n_ref=30000, n_reads=400, region_begin=5000)
    rng = np.random.default_rng(3)
    reads = [c[:int(L)] for c, L in zip(codes, rng.integers(40, 151, size=len(codes)))]  # 40..150 bp; < 63 stay unaligned
    reads.append(np.full(100, 15, np.uint8))  # all N
    reads.append(codes[0][:63])
    o = Oracle(ref, recs, region_begin=5000)
    b = harness.GpuBackend(gtx.graph_from_records(ref, recs, region_begin=5000))
    check_align(b, o, reads)


def test_align_is_independent_of_batch_composition():
    ref, recs, codes, pos = scenarios.synthetic_case("snp100", n_ref=50000, n_reads=3000, region_begin=0)
    b = harness.GpuBackend(gtx.graph_from_records(ref, recs))
    seq = gtx.pack_nibbles(codes)
    meta = harness.read_meta(np.full(len(codes), 150))
    whole = b.align(seq, meta).reshape(len(codes), -1).copy()
    perm = np.random.default_rng(1).permutation(len(codes))
    shuffled = b.align(seq[perm], meta[perm]).reshape(len(codes), -1)
    assert np.array_equal(whole[perm], shuffled)


@pytest.mark.parametrize("kind", ["snp100", "snp25", "indel"])
def test_stream_scores(kind):
    ref, recs, codes, rec = scenarios.paired_case(kind, n_ref=100000, n_pairs=8000, region_begin=310000, n_samples=3)
    o = Oracle(ref, recs, region_begin=310000)
    b = harness.GpuBackend(gtx.graph_from_records(ref, recs, region_begin=310000))
    want = run_stream(b, o, codes, rec, n_samples=3)
    assert want.sum() > 0


def test_scores_do_not_depend_on_item_order():
    ref, recs, codes, rec = scenarios.paired_case("snp100", n_ref=60000, n_pairs=4000, region_begin=0, n_samples=2)
    b = harness.GpuBackend(gtx.graph_from_records(ref, recs))
    st = gtx.Stream(b.ctx.params, 1)
    a_seq, a_meta, items = st.push(rec, gtx.pack_nibbles(codes))
    records = b.align(a_seq, a_meta)
    s1 = harness.canonical_scores(b.ctx, b.score(items, records, 2))
    s2 = harness.canonical_scores(b.ctx, b.score(items[::-1].copy(), records, 2))
    assert np.array_equal(s1, s2)


def test_merged_multiallelic_graph():
    """cfg3-like: add_all_variants merges clusters of SNP, SNP, indel into multi-allelic sites (graph built by
    gtx_graph_build); 30 samples"""
    ref, recs, codes, pos = scenarios.synthetic_case("cluster", n_ref=300000, n_reads=30000, region_begin=1000000)
    o = Oracle(ref, recs, region_begin=1000000, add_all_variants=True)
    g = gtx.graph_from_records(ref, recs, region_begin=1000000, add_all_variants=True)
    assert int(g["ref_nvar"].max()) >= 6
    b = harness.GpuBackend(g)
    check_align(b, o, list(codes), pos=pos)  # (every record with correct, missing, shifted and foreign position hints)
    assert check_align.hinted_done > 0.65 * len(codes)  # (the dense build of pass 0: 70 %; 18 % with the lean one)
    order = np.argsort(pos, kind="stable")
    rec = scenarios.stream_records(len(codes), pos, sample=np.arange(len(codes)) % 30)
    run_stream(b, o, codes[order], rec[order], n_samples=30)


def test_cfg3_graph():
    done = cfg3_case(harness.GpuBackend, 30000, n_ref=300000)
    assert done > 0.94 * 30000, done


@pytest.mark.parametrize("kind", ["repeat", "snp7"])
def test_second_pass(kind):
    second_pass_case(harness.GpuBackend, kind, 5000 if kind == "repeat" else 3000)  # (snp7: ~200 connection entries per read)


def test_satellite_repeats_reach_the_exact_pass():
    """reads in long low-complexity repeats chain more paths than any fixed table holds: the exact pass (tables sized at run
    time in a slab of HBM) must finish every one of them with the oracle's records, and the scores / calls / VCF text follow"""
    satellite_case(harness.GpuBackend, 4000)
    satellite_case(harness.GpuBackend, 1500, read_len=250, seed=4)


def test_exact_pass_with_the_whole_slab(monkeypatch):
    satellite_case(harness.GpuBackend, 1000, read_len=250, seed=1, exact_pass_mb=48, monkeypatch=monkeypatch)


def test_reads_in_a_long_homopolymer():
    homopolymer_case(harness.GpuBackend)


def test_exact_tasks_behind_batches_that_had_none():
    """The exact pass' launches are sized by what the batch before sent there, and behind a batch that sent NOTHING they use the
    build of the pass that can be placed beside a full chip (gtx_align_exact_light_kernel: 128 registers, 2 KB of LDS, four
    workgroups).  A batch whose reads do reach the pass then is done by that build: same records, nothing refused -- and the batch
    behind it gets the full build again."""
    ref = synth.make_reference(20000, seed=3)
    ref[10000:10280] = 1
    recs = synth.make_snp_records(ref, 50, seed=4, region_begin=1000)
    refs = synth.bases_to_str(ref)
    o = Oracle(refs, recs, region_begin=1000)
    b = harness.GpuBackend(gtx.graph_from_records(refs, recs, region_begin=1000))
    easy, epos = synth.make_reads(ref[2000:6000], [r for r in recs if 2000 < r[0] - 1000 < 5998], 800, read_len=150, seed=11, region_begin=1000 + 2000)
    eseq, elens = harness.pack_ragged(list(easy))
    for _ in range(2):  # (the second of these calls is already sized by the first)
        b.align(eseq, harness.read_meta(elens, pos=epos))
        assert b.exact_pass_tasks()[0] == 0
    hard, hpos = synth.make_reads(ref[9700:10580], [r for r in recs if 9700 < r[0] - 1000 < 10578], 1500, read_len=150, seed=9, err=0, n_rate=0,
                                  region_begin=1000 + 9700, rev_frac=0.0)
    check_align(b, o, list(hard), pos=hpos, allow_overflow=False)  # (its first call meets the light build, the others the full one)
    assert b.exact_pass_tasks()[0] >= 15 and b.exact_pass_tasks()[3] == 0
    check_align(b, o, list(easy), pos=epos, allow_overflow=False)


def test_reads_in_copies_of_a_dinucleotide_repeat():
    dinucleotide_case(harness.GpuBackend)


def test_align_five_kmer_reads():
    assert five_kmer_case(harness.GpuBackend, 20000) > 20000


def test_align_near_duplicate_reference():
    done = neardup_case(harness.GpuBackend, 20000, n_ref=200000)
    assert 0 < done < 20000


def test_forced_second_pass(monkeypatch):
    forced_second_pass_case(harness.GpuBackend, monkeypatch, 6000)


def test_sv_calling_host_logic():
    sv_stream_case(harness.GpuBackend, 6000)


def test_align_iupac_codes():
    iupac_case(harness.GpuBackend, 10000)


def test_edge_cases():
    edge_case(harness.GpuBackend)


def test_express_variants_agree(monkeypatch):
    express_variants_case(harness.GpuBackend, monkeypatch, 40000)


def test_pass_times_are_reported():
    """gtx_ctx_pass_times: HIP-event durations of the three alignment passes (what bench.py prices the roofline on)"""
    ref, recs, codes, pos = scenarios.synthetic_case("snp100", n_ref=50000, n_reads=20000, region_begin=0)
    ctx = gtx.Context(gtx.graph_from_records(ref, recs), device=0)
    assert ctx.pass_times() == ([0.0, 0.0, 0.0], 0)  # the first call arms the timing
    b = harness.GpuBackend(gtx.graph_from_records(ref, recs))
    b.align(gtx.pack_nibbles(codes), harness.read_meta(np.full(len(codes), 150), pos=pos))
    ms, handed_on = b.ctx.pass_times()
    assert all(x > 0 for x in ms) and 0 < handed_on < len(codes)
    kt = b.ctx.kernel_times()
    assert [k[0] for k in kt] == ["gtx_align_hinted_kernel", "gtx_align_express4_kernel", "gtx_align_kernel", "gtx_align_big_kernel"]
    assert all(k[1] > 0 for k in kt) and sum(k[2] for k in kt) == len(codes) and kt[0][2] > 0.2 * len(codes)
    assert abs(kt[0][1] + kt[1][1] - ms[0]) < 1e-3 and kt[2][2] + kt[3][2] == handed_on


def test_align_over_an_sv_deletion():
    sv_deletion_case(harness.GpuBackend)


def test_direct_probes_and_half_key_buckets_agree(monkeypatch):
    direct_probes_case(harness.GpuBackend, monkeypatch, 20000)


def test_three_ambiguous_bases_stay_in_the_lds_pass():
    three_n_case(harness.GpuBackend)


def test_align_reference_with_iupac_letters():
    from test_emu_parity import iupac_reference_case
    iupac_reference_case(harness.GpuBackend, 20000)


def test_align_reference_with_n_runs():
    n_runs_case(harness.GpuBackend, 20000)


@pytest.mark.parametrize("rec_words", [16, 18, 24, 100])
def test_record_slot_sizes(rec_words):
    """rec_words is the caller's choice (gtx_align_batch): slots of 16, 24 and 100 words take the staged 16-byte stores of the
    position-hinted pass, 18 words (slots that are not 16-byte aligned) the word-wise ones; a record that does not fit its
    slot goes to the arena.  The parsed paths must be those of 64-word slots."""
    ref, recs, codes, pos = scenarios.synthetic_case("snp100", n_ref=30000, n_reads=3000, region_begin=1000)
    b = harness.GpuBackend(gtx.graph_from_records(ref, recs, region_begin=1000))
    seq, lens = harness.pack_ragged(list(codes))
    meta = harness.read_meta(lens, pos=pos)
    want = gtx.parse_records(b.align(seq, meta), len(codes), harness.REC_WORDS, b.ctx.hap_order, b.big_records()[0])
    done64 = b.hinted_done()
    b.rewind_big_records()
    rec = b.align(seq, meta, rec_words=rec_words)
    assert not ((rec.reshape(-1, rec_words)[:, 0] >> 16) & gtx.ST_ERROR_MASK).any()
    got = gtx.parse_records(rec, len(codes), rec_words, b.ctx.hap_order, b.big_records()[0])
    assert got == want
    assert b.hinted_done() > len(codes) // 2 and (rec_words < 24 or b.hinted_done() == done64)


@pytest.mark.parametrize("hint", [True, False])
def test_task_flags_side_array(hint):
    """gtx_align_batch_flags / gtx_score_batch_flags: the dense byte per (read, orientation) equals bit 31 of the record's
    second word for every task -- whichever pass finished it -- and scoring with it gives the same accumulators"""
    import ctypes as C
    torch = pytest.importorskip("torch")
    ref, recs, codes, rec = scenarios.paired_case("snp100", n_ref=40000, n_pairs=6000, region_begin=310000, n_samples=3)
    rec = rec.copy()
    rec["flag"][::3] &= np.uint16(0xFFFF & ~(1 | 2 | 8 | 32 | 64 | 128))  # every third record unpaired: forward-only items of one read
    b = harness.GpuBackend(gtx.graph_from_records(ref, recs, region_begin=310000))
    st = gtx.Stream(b.ctx.params, 1)
    a_seq, a_meta, items = st.push(rec, gtx.pack_nibbles(codes))
    if not hint:
        a_meta = a_meta.copy()
        a_meta["pos"] = -1
    n = len(a_meta)
    d_seq, d_meta = b._dev(a_seq), b._dev(a_meta)
    d_rec = torch.zeros(n * 2 * harness.REC_WORDS, dtype=torch.int32, device="cuda:0")
    d_flags = torch.full((2 * n,), 0x55, dtype=torch.uint8, device="cuda:0")
    L = gtx.lib()
    gtx.check(L.gtx_align_batch_flags(b.ctx.h, d_seq.data_ptr(), a_seq.shape[1], d_meta.data_ptr(), n, d_rec.data_ptr(), harness.REC_WORDS,
                                      d_flags.data_ptr(), None))
    torch.cuda.synchronize()
    records = d_rec.cpu().numpy().view(np.uint32)
    flags = d_flags.cpu().numpy()
    want = (records.reshape(2 * n, harness.REC_WORDS)[:, 1] >> 31).astype(np.uint8)
    assert np.array_equal(flags, want) and 0 < int(want.sum()) < n
    assert (want[1::2] != 0).any(), "no reverse-orientation task with a variant: the scenario does not cover the general pass' tasks"
    acc = b.score(items, records, 3)
    acc2 = harness.Accumulators(b.ctx, 3)
    devs = [b._dev(a) for a in acc2.arrays()]
    buf = acc2.buffers([d.data_ptr() for d in devs])
    gtx.check(L.gtx_score_batch_flags(b.ctx.h, b._dev(np.ascontiguousarray(items, gtx.SCORE_ITEM)).data_ptr(), len(items), d_rec.data_ptr(),
                                      harness.REC_WORDS, d_flags.data_ptr(), C.byref(buf), None))
    torch.cuda.synchronize()
    for host, dev in zip(acc2.arrays(), devs):
        host[...] = dev.cpu().numpy().view(host.dtype)
    assert np.array_equal(harness.canonical_scores(b.ctx, acc), harness.canonical_scores(b.ctx, acc2))
    # gtx_score_batch_words: the items' compact form for the first stage (one word per item; pairs and reads aligned in both
    # orientations keep GTX_ITEM_WORD_FULL) -- the same accumulators
    words = gtx.item_words(items)
    one = words != 0xFFFFFFFF
    assert 0 < int(one.sum()) < len(words)
    assert np.array_equal(words[one], np.asarray(items)["first"]["align_index"][one])
    acc3 = harness.Accumulators(b.ctx, 3)
    devs = [b._dev(a) for a in acc3.arrays()]
    buf = acc3.buffers([d.data_ptr() for d in devs])
    gtx.check(L.gtx_score_batch_words(b.ctx.h, b._dev(np.ascontiguousarray(items, gtx.SCORE_ITEM)).data_ptr(), b._dev(words).data_ptr(), len(items),
                                      d_rec.data_ptr(), harness.REC_WORDS, d_flags.data_ptr(), C.byref(buf), None))
    torch.cuda.synchronize()
    for host, dev in zip(acc3.arrays(), devs):
        host[...] = dev.cpu().numpy().view(host.dtype)
    assert np.array_equal(harness.canonical_scores(b.ctx, acc), harness.canonical_scores(b.ctx, acc3))
    assert L.gtx_score_batch_words(b.ctx.h, None, b._dev(words).data_ptr(), 0, None, harness.REC_WORDS, None, C.byref(buf), None) != 0  # words need flags


def test_batches_in_flight_equal_one_at_a_time():
    """gtx_align_batch_planes_staged: batches in flight -- position-hinted passes on one stream, the express / general
    queues behind each on a second one, the front event between them -- leave the records and side bytes the plain call
    leaves, batch by batch; and gtx_ctx_kernel_times then holds the mean over exactly those calls"""
    import ctypes as C
    import torch
    ref, recs, codes, pos = scenarios.synthetic_case("snp100", n_ref=120000, n_reads=9000, region_begin=1000000, seed=3)
    ctx = gtx.Context(gtx.graph_from_records(ref, recs, region_begin=1000000), device=0)
    L = gtx.lib()
    seq, lens = harness.pack_ragged(list(codes))
    meta = harness.read_meta(lens, pos=pos)
    stride = (seq.shape[1] + 15) // 16 * 16
    planes = gtx.pack_planes(seq, stride)
    # (eight batches: more than the library keeps scratches in flight for one stream -- the later calls wait for the oldest)
    parts = [(0, 1500), (1500, 3000), (3000, 4200), (4200, 5400), (5400, 6600), (6600, 7400), (7400, 8200), (8200, 9000)]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to("cuda:0")
    want = []
    for a, b in parts:  # one at a time, default stream
        d_p, d_m = dev(planes[a:b]), dev(meta[a:b])
        d_rec = torch.zeros((b - a) * 2 * harness.REC_WORDS, dtype=torch.int32, device="cuda:0")
        d_fl = torch.zeros((b - a) * 2, dtype=torch.uint8, device="cuda:0")
        gtx.check(L.gtx_align_batch_planes(ctx.h, d_p.data_ptr(), stride, d_m.data_ptr(), b - a, d_rec.data_ptr(), harness.REC_WORDS,
                                           d_fl.data_ptr(), None))
        torch.cuda.synchronize()
        want.append((d_rec.cpu().numpy().copy(), d_fl.cpu().numpy().copy()))
    ctx.pass_times()  # arms the timing
    ctx.pass_times()
    H, T = torch.cuda.Stream(), torch.cuda.Stream()
    held, fronts = [], []
    for a, b in parts:
        d_p, d_m = dev(planes[a:b]), dev(meta[a:b])
        d_rec = torch.zeros((b - a) * 2 * harness.REC_WORDS, dtype=torch.int32, device="cuda:0")
        d_fl = torch.zeros((b - a) * 2, dtype=torch.uint8, device="cuda:0")
        held.append((d_p, d_m, d_rec, d_fl))
    torch.cuda.synchronize()
    for (a, b), (d_p, d_m, d_rec, d_fl) in zip(parts, held):
        ev, done = torch.cuda.Event(), torch.cuda.Event()
        ev.record(H)  # (creates the HIP events)
        done.record(H)
        fronts.append((ev, done))
        gtx.check(L.gtx_align_batch_planes_staged(ctx.h, d_p.data_ptr(), stride, d_m.data_ptr(), b - a, d_rec.data_ptr(), harness.REC_WORDS,
                                                  d_fl.data_ptr(), C.c_void_p(H.cuda_stream), C.c_void_p(ev.cuda_event), C.c_void_p(T.cuda_stream),
                                                  C.c_void_p(done.cuda_event) if (a // 1000) % 2 else None))
    torch.cuda.synchronize()
    for (w_rec, w_fl), (_, _, d_rec, d_fl) in zip(want, held):
        assert np.array_equal(d_rec.cpu().numpy(), w_rec) and np.array_equal(d_fl.cpu().numpy(), w_fl)
    kt = ctx.kernel_times()
    assert kt[0][1] > 0 and kt[2][1] > 0  # mean durations over the three staged calls
    assert kt == ctx.kernel_times()       # a second query without a call in between: the same series
    assert 0 < kt[0][2] <= parts[-1][1] - parts[-1][0]  # task counts: the last call's
    # a tail stream without the front event is refused
    d_p, d_m, d_rec, d_fl = held[0]
    assert L.gtx_align_batch_planes_staged(ctx.h, d_p.data_ptr(), stride, d_m.data_ptr(), 10, d_rec.data_ptr(), harness.REC_WORDS, None,
                                           C.c_void_p(H.cuda_stream), None, C.c_void_p(T.cuda_stream), None) != 0


@pytest.mark.parametrize("kind", ["repeat", "satellite"])
def test_whole_calls_in_flight_on_streams_of_their_own(kind):
    """bench.py's schedule for repeat-rich input: three gtx_align_batch_planes calls on one context, each on a stream of its own with
    its own record slots, in flight together -- the HBM-table and exact passes of one beside the front passes of another, all of
    them taking places in ONE arena.  Every call leaves what it leaves alone (records in the arena compared parsed: their offsets
    are handed out in whatever order the workgroups finish)."""
    import ctypes as C
    import torch
    ref, recs, codes, pos = scenarios.synthetic_case(kind, n_ref=60000, n_reads=1800, region_begin=1000000, seed=11)
    ctx = gtx.Context(gtx.graph_from_records(ref, recs, region_begin=1000000), device=0)
    L = gtx.lib()
    seq, lens = harness.pack_ragged(list(codes))
    meta = harness.read_meta(lens, pos=pos)
    stride = (seq.shape[1] + 15) // 16 * 16
    planes = gtx.pack_planes(seq, stride)
    parts = [(0, 600), (600, 1200), (1200, 1800)]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to("cuda:0")

    def parsed(d_rec, n):
        words = d_rec.cpu().numpy().view(np.uint32).copy().reshape(-1, harness.REC_WORDS)
        arena, _ = ctx.big_records()
        assert not ((words[:, 0] >> 16) & gtx.ST_ERROR_MASK).any()
        ext = np.nonzero(((words[:, 0] >> 16) & gtx.ST_EXTERNAL) != 0)[0]
        long_reads = np.unique(ext // 2)
        body = gtx.parse_records(words.reshape(-1, 2 * harness.REC_WORDS)[long_reads].reshape(-1), len(long_reads), harness.REC_WORDS, ctx.hap_order,
                                 np.asarray(arena))
        words[ext, 2] = 0
        return words, list(long_reads), body

    want = []
    for a, b in parts:  # one at a time
        d_p, d_m = dev(planes[a:b]), dev(meta[a:b])
        d_rec = torch.zeros((b - a) * 2 * harness.REC_WORDS, dtype=torch.int32, device="cuda:0")
        d_fl = torch.zeros((b - a) * 2, dtype=torch.uint8, device="cuda:0")
        ctx.rewind_big_records()
        gtx.check(L.gtx_align_batch_planes(ctx.h, d_p.data_ptr(), stride, d_m.data_ptr(), b - a, d_rec.data_ptr(), harness.REC_WORDS, d_fl.data_ptr(), None))
        torch.cuda.synchronize()
        want.append(parsed(d_rec, b - a) + (d_fl.cpu().numpy().copy(),))
    assert sum(len(w[1]) for w in want) > 20  # (records in the arena: what the passes behind the general one write)
    for _ in range(3):
        ctx.rewind_big_records()
        streams = [torch.cuda.Stream() for _ in parts]
        held = []
        for a, b in parts:
            held.append((dev(planes[a:b]), dev(meta[a:b]), torch.zeros((b - a) * 2 * harness.REC_WORDS, dtype=torch.int32, device="cuda:0"),
                         torch.zeros((b - a) * 2, dtype=torch.uint8, device="cuda:0")))
        torch.cuda.synchronize()
        for (a, b), (d_p, d_m, d_rec, d_fl), st in zip(parts, held, streams):
            gtx.check(L.gtx_align_batch_planes(ctx.h, d_p.data_ptr(), stride, d_m.data_ptr(), b - a, d_rec.data_ptr(), harness.REC_WORDS, d_fl.data_ptr(),
                                               C.c_void_p(st.cuda_stream)))
        torch.cuda.synchronize()
        for (a, b), (_, _, d_rec, d_fl), (w_words, w_long, w_body, w_fl) in zip(parts, held, want):
            words, long_reads, body = parsed(d_rec, b - a)
            assert np.array_equal(words, w_words) and long_reads == w_long and body == w_body and np.array_equal(d_fl.cpu().numpy(), w_fl)


@pytest.mark.parametrize("kind", ["cfg3", "cluster", "snp25"])
def test_both_builds_of_pass_0_write_the_same_records(kind, monkeypatch):
    """gtx_align_hinted_kernel and gtx_align_hinted_dense_kernel (GTX_HINT_BUILD) on one context: the same record words,
    the dense build finishes more of a dense graph's reads itself; both equal the oracle (tests above run whichever is the default)"""
    ref, recs, codes, pos = scenarios.synthetic_case(kind, n_ref=200000, n_reads=20000, region_begin=1000000, seed=2)
    aav = kind in ("cluster", "cfg3")
    g = gtx.graph_from_records(ref, recs, region_begin=1000000, add_all_variants=aav)
    b = harness.GpuBackend(g)
    seq, lens = harness.pack_ragged(list(codes))
    meta = harness.read_meta(lens, pos=pos)
    out, done = {}, {}
    for build in ("lean", "dense"):
        monkeypatch.setenv("GTX_HINT_BUILD", build)
        out[build] = b.align(seq, meta).reshape(2 * len(lens), -1).copy()
        done[build] = b.hinted_done()
    external = ((out["lean"][:, 0] >> 16) & gtx.ST_EXTERNAL) != 0
    differ = out["lean"] != out["dense"]
    differ[external, 2:] = False
    assert not differ.any(), np.nonzero(differ.any(1))[0][:5]
    assert done["dense"] >= done["lean"] and (kind == "snp25" or done["dense"] > 1.05 * done["lean"])


@pytest.mark.parametrize("kind,read_len", [("snp1k", 250), ("snp100", 256), ("cfg3", 200)])
def test_long_reads_through_pass_0(kind, read_len):
    """reads of 161..256 bases: gtx_align_hinted_long_kernel (eight k-mers, rows of 128 bytes), records == oracle with four kinds of hints"""
    from test_long_reads import long_read_case
    long_read_case(harness.GpuBackend, kind, read_len, 6000, 0.9 if kind == "snp1k" else 0.2)


@pytest.mark.parametrize("kind", ["snp1k", "snp100", "cfg3", "long"])
def test_dense_records_of_pass_0(kind, monkeypatch):
    """gtx_align_batch_planes_compact: what the position-hinted pass finishes without a variant site leaves as 32-byte records side by
    side (d_compact, GTX_TASK_COMPACT in the side array), its slot untouched; read back as gtx_align_batch's records they are the
    oracle's paths with four kinds of hints (harness.GpuBackend.align under HARNESS_COMPACT), all three builds of the pass"""
    monkeypatch.setenv("HARNESS_COMPACT", "1")
    if kind == "long":
        from test_long_reads import long_read_case
        long_read_case(harness.GpuBackend, "snp1k", 250, 6000, 0.9)
        return
    if kind == "cfg3":
        cfg3_case(harness.GpuBackend, 6000)
        return
    ref, recs, codes, pos = scenarios.synthetic_case(kind, n_ref=200000, n_reads=20000, region_begin=1000000)
    o = Oracle(ref, recs, region_begin=1000000)
    b = harness.GpuBackend(gtx.graph_from_records(ref, recs, region_begin=1000000))
    check_align(b, o, list(codes), pos=pos)
    assert b.compact is not None
    # (the last align of check_align carries foreign hints; the first one, with the right ones, is what the share is about)
    seq, meta = gtx.pack_nibbles(codes), harness.read_meta(np.full(len(codes), 150), pos=pos)
    b.align(seq, meta)
    # (a SNP every 100 bases: every read carries a site, its record is not one of the dense ones)
    assert (b.compact["share"] > 0.6) if kind == "snp1k" else (b.compact["share"] < 0.05), b.compact["share"]


@pytest.mark.parametrize("triaged", ["", "words", "items"])
@pytest.mark.parametrize("kind", ["snp100", "snp25"])
def test_stream_scores_over_dense_records(kind, triaged, monkeypatch):
    """the scorer reads a mate's record where the position-hinted pass left it (gtx_score_batch_compact, gtx_scores_replay_compact):
    pairs, duplicates, parked mates, three samples -- accumulators, calls, flags and VCF text == the oracle's; `triaged`: the
    scorer's first stage behind the alignment on a second stream, the second stage alone over its queue
    (gtx_align_batch_planes_triaged, gtx_score_batch_queued), with and without the items' words"""
    monkeypatch.setenv("HARNESS_COMPACT", "1")
    if triaged:
        monkeypatch.setenv("HARNESS_TRIAGED", "1")
        monkeypatch.setenv("HARNESS_TRIAGED_WORDS", "1" if triaged == "words" else "0")
    ref, recs, codes, rec = scenarios.paired_case(kind, n_ref=100000, n_pairs=8000, region_begin=310000, n_samples=3)
    o = Oracle(ref, recs, region_begin=310000)
    b = harness.GpuBackend(gtx.graph_from_records(ref, recs, region_begin=310000))
    want = run_stream(b, o, codes, rec, n_samples=3)
    assert want.sum() > 0 and b.compact is not None
    assert not triaged or b.triaged["queued"].size > 0


@pytest.mark.parametrize("kind", ["snp1k", "snp100", "cfg3"])
def test_first_scoring_stage_behind_the_alignment(kind, monkeypatch):
    """gtx_align_batch_planes_triaged + gtx_score_batch_queued over unpaired reads (item i = read i), a third of them with a missing or
    shifted hint (their records come from the passes behind the position-hinted one): the accumulators of gtx_score_batch_compact,
    whichever way the first stage learns of the reads with a variant site -- the items' words, the items, the reads' bits
    (GTX_TRIAGE_ITEMS_ARE_READS) -- and the same queue all three times"""
    monkeypatch.setenv("HARNESS_COMPACT", "1")
    ref, recs, codes, pos = scenarios.synthetic_case(kind, n_ref=200000, n_reads=20000, region_begin=1000000, seed=5)
    b = harness.GpuBackend(gtx.graph_from_records(ref, recs, region_begin=1000000, add_all_variants=kind == "cfg3"))
    n = len(codes)
    hint = np.array(pos, np.int64)
    rng = np.random.default_rng(7)
    hint[rng.random(n) < 0.15] = -1
    shifted = rng.random(n) < 0.15
    hint[shifted & (hint >= 0)] += 1
    meta = harness.read_meta(np.full(n, 150), flags=np.full(n, gtx.FLAG_FORWARD_ONLY), pos=hint)
    records = b.align(gtx.pack_nibbles(codes), meta)
    assert b.compact is not None
    items = np.zeros(n, gtx.SCORE_ITEM)
    items["first"]["align_index"] = np.arange(n, dtype=np.uint32)
    items["first"]["mapq"] = 60
    items["first"]["flag"] = gtx.FLAG_FORWARD_ONLY
    items["first"]["pos"] = pos
    items["second"]["align_index"] = gtx.INVALID_ID
    want = harness.canonical_scores(b.ctx, b.score(items, records))  # (the connection log is a list in the order the items were scored)
    assert want.sum() > 0
    queues = []
    for mode in ("words", "items", "reads"):
        monkeypatch.setenv("HARNESS_TRIAGED", "1")
        monkeypatch.setenv("HARNESS_TRIAGED_WORDS", "0" if mode == "items" else "1")
        monkeypatch.setenv("HARNESS_TRIAGED_READS", "1" if mode == "reads" else "0")
        got = harness.canonical_scores(b.ctx, b.score(items, records))
        assert np.array_equal(want, got), mode
        queues.append(b.triaged["queued"])
    assert len(queues[0]) > 0 and np.array_equal(queues[0], queues[1]) and np.array_equal(queues[0], queues[2])
    # (the queue is the reads whose forward record carries a variant site: the side array says which)
    assert np.array_equal(queues[0], np.nonzero(b.compact["fl"][0::2] & gtx.TASK_HAS_VARIANTS)[0])


def test_ambiguous_bases_beside_substitutions():
    """(the emulation's case on the device, more reads)"""
    from test_emu_parity import ambiguous_beside_substitutions_case
    done = ambiguous_beside_substitutions_case(harness.GpuBackend, 6000)
    assert done["snp1k"] > 1500, done


def test_arguments_of_the_triaged_entry_points():
    """gtx_align_batch_planes_triaged / gtx_score_batch_queued refuse what they cannot do: no side array, no items without
    GTX_TRIAGE_ITEMS_ARE_READS, a queue that is not 16-byte aligned, n_items != n_reads with that flag, an unknown flag bit; an
    empty batch leaves an empty queue"""
    import torch
    ref, recs, codes, pos = scenarios.synthetic_case("snp1k", n_ref=30000, n_reads=256, region_begin=1000000, seed=3)
    b = harness.GpuBackend(gtx.graph_from_records(ref, recs, region_begin=1000000))
    L, n = gtx.lib(), len(codes)
    seq = gtx.pack_nibbles(codes)
    stride = (seq.shape[1] + 15) // 16 * 16
    d_seq = b._dev(seq)
    d_planes = torch.zeros(n * stride, dtype=torch.uint8, device="cuda:0")
    gtx.check(L.gtx_reads_to_planes(b.ctx.h, d_seq.data_ptr(), seq.shape[1], n, d_planes.data_ptr(), stride, None))
    d_meta = b._dev(harness.read_meta(np.full(n, 150), flags=np.full(n, gtx.FLAG_FORWARD_ONLY), pos=pos))
    d_rec = torch.zeros(n * 2 * harness.REC_WORDS, dtype=torch.int32, device="cuda:0")
    d_comp = torch.zeros(n * gtx.COMPACT_WORDS, dtype=torch.int32, device="cuda:0")
    d_fl = torch.zeros(2 * n, dtype=torch.uint8, device="cuda:0")
    d_work = torch.full((n + gtx.WORK_HEADER_WORDS + 4,), -1, dtype=torch.int32, device="cuda:0")
    items = np.zeros(n, gtx.SCORE_ITEM)
    items["first"]["align_index"] = np.arange(n, dtype=np.uint32)
    items["first"]["flag"] = gtx.FLAG_FORWARD_ONLY
    items["second"]["align_index"] = gtx.INVALID_ID
    d_items = b._dev(items)

    def call(fl=d_fl.data_ptr(), it=d_items.data_ptr(), n_items=n, flags=0, work=d_work.data_ptr(), n_reads=n):
        return L.gtx_align_batch_planes_triaged(b.ctx.h, d_planes.data_ptr(), stride, d_meta.data_ptr(), n_reads, d_rec.data_ptr(), harness.REC_WORDS,
                                                d_comp.data_ptr(), fl, it, None, n_items, flags, work, None, None, None, None)

    assert call(fl=None) == 1
    assert call(it=None) == 1
    assert call(work=d_work.data_ptr() + 4) == 1
    assert call(flags=gtx.TRIAGE_ITEMS_ARE_READS, n_items=n - 1) == 1
    assert call(flags=2) == 1
    assert call(it=None, flags=gtx.TRIAGE_ITEMS_ARE_READS) == 0  # (the items are not looked at)
    torch.cuda.synchronize()
    by_bits = d_work.cpu().numpy().view(np.uint32).copy()
    assert call() == 0
    torch.cuda.synchronize()
    by_items = d_work.cpu().numpy().view(np.uint32)
    k = int(by_bits[0])
    assert 0 < k == int(by_items[0]) < n
    assert np.array_equal(np.sort(by_bits[gtx.WORK_HEADER_WORDS:gtx.WORK_HEADER_WORDS + k]), np.sort(by_items[gtx.WORK_HEADER_WORDS:gtx.WORK_HEADER_WORDS + k]))
    assert call(n_reads=0, n_items=0) == 0
    torch.cuda.synchronize()
    assert int(d_work.cpu().numpy().view(np.uint32)[0]) == 0
    acc = harness.Accumulators(b.ctx, 1)
    devs = [b._dev(a) for a in acc.arrays()]
    buf = acc.buffers([d.data_ptr() for d in devs])
    assert L.gtx_score_batch_queued(b.ctx.h, d_items.data_ptr(), n, d_rec.data_ptr(), harness.REC_WORDS, d_comp.data_ptr(), d_fl.data_ptr(), None, C.byref(buf), None) == 1
    assert L.gtx_score_batch_queued(b.ctx.h, d_items.data_ptr(), n, d_rec.data_ptr(), harness.REC_WORDS, d_comp.data_ptr(), None, d_work.data_ptr(), C.byref(buf), None) == 1
