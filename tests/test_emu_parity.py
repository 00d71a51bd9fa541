"""Kernel sources run through the host emulation (tests/emu) against the oracle.  A debugging aid for containers
without a GPU -- the parity claims of record are the `-m gpu` tests, which run the same cases through libgtx."""
import ctypes as C

import os

import numpy as np
import pytest

import harness
import scenarios
from graphtyper_amd import lib as gtx
from graphtyper_amd import synth
from oracle_lib import Oracle, encode


@pytest.fixture(scope="module", autouse=True)
def _built():
    gtx.build()


def check_align(backend, oracle, reads, flags=None, tid=None, mtid=None, isize=None, allow_overflow=False, pos=None):
    """kernel records == oracle GenotypePaths for every read; reads whose status word reports a table overflow are
    only tolerated where the test says so (low-complexity contigs) and are never compared as if they were results.
    pos: position hints (gtx_read_meta::pos).  The records must not depend on them: the batch is also aligned without
    hints, with every hint off by one base and with the hints of other reads, and all four results must be the same words."""
    seq, lens = harness.pack_ragged(reads)
    meta = harness.read_meta(lens, flags, tid, mtid, isize, pos)
    if pos is not None:
        pos = np.asarray(pos, np.int64)
        variants = [None, pos + 1, pos - 31, np.roll(pos, 1)]
        words = []
        for v in variants:
            words.append(backend.align(seq, harness.read_meta(lens, flags, tid, mtid, isize, v)).copy())
            if v is None and not os.environ.get("HARNESS_SHORT_READS"):
                # (the device's count includes the reads the pass settles without aligning them -- shorter than two k-mers: the sweep
                #  of stress_emu.py --gpu has such reads and says so)
                assert backend.hinted_done() == 0, "a read without a hint was finished by the position-hinted pass"
    rec = backend.align(seq, meta)
    check_align.hinted_done = backend.hinted_done()
    if pos is not None:
        # (a record that went to the big-record arena keeps an arena offset in its slot, which differs from call to call:
        # only its two header words are compared here -- the position-hinted pass never writes such records)
        a = rec.reshape(2 * len(reads), -1)
        external = ((a[:, 0] >> 16) & gtx.ST_EXTERNAL) != 0
        for v, w in zip(variants, words):
            w = w.reshape(2 * len(reads), -1)
            differ = (w != a)
            differ[external, 2:] = False
            diff = np.nonzero(differ.any(1))[0] // 2
            assert len(diff) == 0, "records depend on the position hint (reads %s, hint variant %r)" % (diff[:5], None if v is None else "shifted")
    big, _ = backend.big_records()
    got = gtx.parse_records(rec, len(reads), harness.REC_WORDS, backend.ctx.hap_order, big)
    want = oracle.align(reads, flags=flags, tid=tid, mtid=mtid, isize=isize)
    n_over = 0
    for i, (a, b) in enumerate(zip(got, want)):
        for o in range(2):
            if a[o]["status"] != 0:
                assert allow_overflow, "read %d orientation %d: kernel table overflow %d" % (i, o, a[o]["status"])
                assert a[o]["paths"] == [] and a[o]["longest"] == 0
                n_over += 1
                continue
            ga = dict(longest=a[o]["longest"], paths=a[o]["paths"])
            assert ga == b[o], "read %d orientation %d: kernel %r != oracle %r" % (i, o, ga, b[o])
    return rec, n_over


@pytest.mark.parametrize("chrom", ["chr1", "chr2", "chr3", "chr9", "chr10", "chr11"])
def test_align_index_test_contigs(chrom):
    ref, recs, reads = scenarios.contig_reads(chrom)
    o = Oracle(ref, recs, force_both=True)
    b = harness.EmuBackend(gtx.graph_from_records(ref, recs), force_both=True)
    check_align(b, o, [encode(r) for r in reads])  # (chr9 is 80 bp of poly-G: its reads need the second pass)


@pytest.mark.parametrize("kind", ["snp1k", "snp100", "snp25", "indel"])
def test_align_synthetic(kind):
    ref, recs, codes, pos = scenarios.synthetic_case(kind, n_ref=100000, n_reads=6000, region_begin=777000)
    o = Oracle(ref, recs, region_begin=777000)
    b = harness.EmuBackend(gtx.graph_from_records(ref, recs, region_begin=777000))
    check_align(b, o, list(codes), pos=pos)
    # the position-hinted pass has to carry its share of the reads (else the hints were not understood)
    share = check_align.hinted_done / float(len(codes))
    assert share >= {"snp1k": 0.70, "snp100": 0.10, "snp25": 0.0, "indel": 0.05}[kind], share
    print("position-hinted share", kind, share)
    test_align_synthetic.share = getattr(test_align_synthetic, "share", {})
    test_align_synthetic.share[kind] = share


def neardup_case(Backend, n_reads, n_ref=60000):
    """a reference with planted near-duplicate segments (copies one substitution apart) and homopolymer runs, 1 % errors:
    the flags of the position-hinted pass must know which places are provably simple; every record with correct, missing,
    shifted and foreign hints against the oracle"""
    ref, recs, codes, pos = scenarios.synthetic_case("neardup", n_ref=n_ref, n_reads=n_reads, region_begin=44000, err=0.01, n_rate=0.002)
    o = Oracle(ref, recs, region_begin=44000)
    b = Backend(gtx.graph_from_records(ref, recs, region_begin=44000))
    check_align(b, o, list(codes), pos=pos, allow_overflow=False)
    return check_align.hinted_done


def five_kmer_case(Backend, n_reads):
    """reads of 156-160 bases: five k-mers, the fifth ending on base 155 with four bases of tail behind it -- where the
    position-hinted pass' rules for holes, parallel chains (an ambiguous base opening the run behind a hole) and runs of one
    length meet its last counters; 2 % substitutions, 0.5 % N, SNPs every 100 and every 1000 bases"""
    done = 0
    for kind, seed in (("snp100", 5), ("snp1k", 6)):
        ref, recs, codes, pos = scenarios.synthetic_case(kind, n_ref=60000, n_reads=n_reads, region_begin=20000, err=0.02, n_rate=0.005,
                                                       seed=seed, read_len=160)
        rng = np.random.default_rng(seed)
        reads = [c[:int(n)].copy() for c, n in zip(codes, rng.integers(156, 161, size=len(codes)))]
        # planted on every eighth read (on top of its own errors): k-mers 1 and 2 broken by two substitutions each, an ambiguous
        # base in k-mer 3 -- the run of k-mers 3, 4 opens with a parallel chain behind a hole --, and one substitution in k-mer 4,
        # on its last base (155, the tail walk's first) or next to it
        sub = lambda c: np.uint8({1: 2, 2: 4, 4: 8, 8: 1}.get(int(c), 1))
        for i in range(0, len(reads), 8):
            r = reads[i]
            for j in (33, 40, 66, 71):
                r[j] = sub(r[j])
            r[int(rng.integers(96, 120))] = 15
            j = int(rng.choice([155, 155, 154, 130]))
            r[j] = sub(r[j])
        o = Oracle(ref, recs, region_begin=20000)
        b = Backend(gtx.graph_from_records(ref, recs, region_begin=20000))
        check_align(b, o, reads, pos=pos)
        done += check_align.hinted_done
    return done


def test_align_five_kmer_reads():
    assert five_kmer_case(harness.EmuBackend, 6000) > 6000


def test_align_near_duplicate_reference():
    done = neardup_case(harness.EmuBackend, 3000)
    assert 0 < done < 3000


def iupac_case(Backend, n_reads):
    """reads with IUPAC ambiguity codes of every kind (2-, 3- and 4-base sets; 1 % of the bases): key lists in
    to_uint64_vec order for one ambiguous base per k-mer (vector path) and for several (sequential expansion, up to the
    97-partial-keys bail-out)"""
    ref, recs, codes, pos = scenarios.synthetic_case("snp100", n_ref=40000, n_reads=n_reads, region_begin=1000, n_rate=0.0)
    rng = np.random.default_rng(11)
    codes = codes.copy()
    amb = rng.random(codes.shape) < 0.01
    # widen the true base to a set that contains it (so most reads still align) or, sometimes, to an unrelated set
    extra = rng.integers(1, 16, size=codes.shape).astype(np.uint8)
    codes[amb] = np.where(rng.random(int(amb.sum())) < 0.8, codes[amb] | extra[amb], extra[amb])
    dense = rng.random(len(codes)) < 0.02  # a few reads with many ambiguous bases
    codes[dense, 40:60] = 15
    o = Oracle(ref, recs, region_begin=1000)
    b = Backend(gtx.graph_from_records(ref, recs, region_begin=1000))
    check_align(b, o, list(codes), pos=pos)


def test_align_iupac_codes():
    iupac_case(harness.EmuBackend, 3000)


def two_ambiguous_case(Backend, n_reads):
    """two ambiguity codes in one k-mer: in the same half (the position-hinted pass proves the result from the other
    half) or one in each (left to the global lookups); sets with and without the true base"""
    ref, recs, codes, pos = scenarios.synthetic_case("snp1k", n_ref=40000, n_reads=n_reads, region_begin=1000, n_rate=0.0)
    rng = np.random.default_rng(12)
    codes = codes.copy()
    same_half = 0
    for c in codes:
        k, half = int(rng.integers(0, 4)), int(rng.integers(0, 2))
        a, b2 = rng.choice(16, size=2, replace=False)
        split = rng.random() < 0.3
        at = [31 * k + 16 * half + int(a), 31 * k + 16 * (half ^ 1 if split else half) + int(b2)]
        same_half += not split
        for x in at:
            r = rng.random()
            c[x] = 15 if r < 0.6 else (c[x] | int(rng.integers(1, 16))) if r < 0.85 else int(rng.integers(1, 16))
    o = Oracle(ref, recs, region_begin=1000)
    b = Backend(gtx.graph_from_records(ref, recs, region_begin=1000))
    check_align(b, o, list(codes), pos=pos)
    return check_align.hinted_done, same_half


def test_two_ambiguous_bases_in_one_kmer():
    done, same_half = two_ambiguous_case(harness.EmuBackend, 1500)
    assert done > same_half // 2, "the position-hinted pass takes none of the k-mers with two ambiguous bases in one half"


def ambiguous_beside_substitutions_case(Backend, n_reads):
    """reads with 2 % N and 3 % substitutions over SNP and indel graphs: k-mers with ambiguous bases AND substitutions -- the
    position-hinted pass calls such a k-mer label-less when the half without an ambiguous base has a substitution and occurs in
    no indexed key, whatever the other half holds (hinted.hpp: hint_kmer_judge); records == oracle with four kinds of hints.
    Returns the reads the pass finished, per kind."""
    done = {}
    for kind, rb in (("snp1k", 1000000), ("snp100", 0), ("cfg3", 1000)):
        ref, recs, codes, pos = scenarios.synthetic_case(kind, n_ref=60000, n_reads=n_reads, region_begin=rb, err=0.03, n_rate=0.02, seed=61)
        aav = kind == "cfg3"
        b = Backend(gtx.graph_from_records(ref, recs, region_begin=rb, add_all_variants=aav))
        check_align(b, Oracle(ref, recs, region_begin=rb, add_all_variants=aav), list(codes), pos=pos)
        done[kind] = check_align.hinted_done
    return done


def test_ambiguous_bases_beside_substitutions():
    done = ambiguous_beside_substitutions_case(harness.EmuBackend, 1200)
    assert done["snp1k"] > 300, done  # (at these rates nine reads in ten hold such a k-mer: the pass still finishes a third of them)


def run_stream(backend, oracle, codes, rec, n_samples, n_rg=1):
    """product: gtx_stream -> align -> score; oracle: Genotyper::push; compares the canonical score streams"""
    og = oracle.genotyper(n_samples, n_rg)
    og.push(list(codes), flags=rec["flag"], tid=rec["tid"], mtid=rec["mtid"], pos=rec["pos"], isize=rec["isize"],
            mapq=rec["mapq"], score_diff=rec["score_diff"], name=rec["name_id"], sample=rec["sample"], rg=rec["rg"])
    want = og.scores()
    st = gtx.Stream(backend.ctx.params, n_rg)
    seq = gtx.pack_nibbles(codes)
    # feed in two pushes to exercise state carried across calls (duplicate detection, parked mates)
    h = len(rec) // 2
    parts = [st.push(rec[:h], seq[:h]), st.push(rec[h:], seq[h:])]
    a_seq = np.concatenate([p[0] for p in parts])
    a_meta = np.concatenate([p[1] for p in parts])
    items = np.concatenate([p[2] for p in parts])
    assert st.counts() == og.counts()
    records = backend.align(a_seq, a_meta)
    status = records.reshape(-1, harness.REC_WORDS)[:, 0] >> 16
    assert not (status & gtx.ST_ERROR_MASK).any(), "kernel table overflow"
    acc = backend.score(items, records, n_samples)
    got = harness.canonical_scores(backend.ctx, acc)
    assert len(got) == len(want)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, "score streams differ at words %s" % bad[:10]
    # connections between near sites went to the dense counters; without them every pair is logged: same content
    acc_log = backend.score(items, records, n_samples, near=False)
    assert np.array_equal(harness.canonical_scores(backend.ctx, acc_log), got)
    assert int(acc.conn_count[0]) <= int(acc_log.conn_count[0])
    run_stream.logged = (int(acc.conn_count[0]), int(acc_log.conn_count[0]))
    # genotype calls: PL, GT, GQ, depths per haplotype and sample (vcf.cpp:47-82, sample_call.cpp:34-131)
    phred, calls = backend.calls(acc, n_samples)
    got_calls, want_calls = harness.canonical_calls(backend.ctx, phred, calls, n_samples), og.calls()
    assert len(got_calls) == len(want_calls) and np.array_equal(got_calls, want_calls), "sample calls differ"
    assert len(np.unique(phred)) > 5 and (calls["gt_second"] > 0).any()  # not vacuous
    # phasing flags from the finalised depths and the connection log (hts_parallel_reader.cpp:782-904)
    gt_cov = np.minimum(acc.gt_cov, 0xFFFF).astype(np.uint32)
    ph = backend.ctx.phase_flags(n_samples, gt_cov, acc.conn_log, int(acc.conn_count[0]), acc.conn_near)
    want_ph = og.phase_flags()
    assert ph.shape == want_ph.shape and np.array_equal(ph, want_ph), "phase flags differ"
    run_stream.last_phase_rows = len(ph)
    # VCF text of the region's sites: Vcf::add_haplotype -> scan_calls / generate_infos -> write_record (oracle/gto_vcf.hpp)
    if not backend.ctx.params.is_sv_graph:
        names = ["SAMP%02d" % i for i in range(n_samples)]
        lo, hi = int(backend.ctx.hap_order[len(backend.ctx.hap_order) // 4]), int(backend.ctx.hap_order[-2])
        for kw in (dict(), dict(region_begin=lo, region_end=hi, filter_zero_qual=True)):
            got_vcf = backend.ctx.vcf_records("chrT", names, acc.gt_cov, acc.stat_u64, acc.stat_u32, phred, calls,
                                              variant_suffix_id="x1" if kw else None, **kw)
            want_vcf = og.vcf_records("chrT", names, suffix_id="x1" if kw else None, **kw)
            if got_vcf != want_vcf:
                gl, wl = got_vcf.split(b"\n"), want_vcf.split(b"\n")
                bad = [i for i in range(min(len(gl), len(wl))) if gl[i] != wl[i]]
                raise AssertionError("VCF text differs (%d vs %d lines), first at line %s:\n%r\n%r" % (len(gl), len(wl), bad[:1], gl[bad[0]] if bad else b"", wl[bad[0]] if bad else b""))
            if not kw:
                run_stream.vcf_full = got_vcf
        run_stream.vcf = got_vcf
        # the file genotype() ends with: vcf_merge_and_break with the variants broken down (vcf_operations.cpp:480-732).  Sites with
        # alleles of different lengths need paw::Skyr in the reference: both sides write them whole with no_variant_overlapping and
        # refuse without it
        names = ["SAMP%02d" % i for i in range(n_samples)]
        o_ = og.o
        run_stream.final_by_mode = {}
        for nvo in (True, False):
            try:
                want_final = og.vcf_records_final("chrT", names, o_.reference, o_.region_begin + 1, no_variant_overlapping=nvo)
            except RuntimeError as e:
                assert not nvo and "Skyr" in str(e)
                with pytest.raises(gtx.GtxError):
                    backend.ctx.vcf_records_final("chrT", names, acc.gt_cov, acc.stat_u64, acc.stat_u32, phred, calls, no_variant_overlapping=nvo)
                continue
            got_final = backend.ctx.vcf_records_final("chrT", names, acc.gt_cov, acc.stat_u64, acc.stat_u32, phred, calls, no_variant_overlapping=nvo)
            if got_final != want_final:
                gl, wl = got_final.split(b"\n"), want_final.split(b"\n")
                bad = [i for i in range(min(len(gl), len(wl))) if gl[i] != wl[i]]
                raise AssertionError("final VCF (no_variant_overlapping=%s) differs (%d vs %d lines), first at line %s:\n%r\n%r" %
                                     (nvo, len(gl), len(wl), bad[:1], gl[bad[0]][:300] if bad else b"", wl[bad[0]][:300] if bad else b""))
            run_stream.final = got_final
            run_stream.final_by_mode[nvo] = got_final
        # the sites the next iteration's graph is built from: vcf_merge_and_filter (vcf_operations.cpp:278-478) with the flags above
        got_sites = backend.ctx.vcf_sites("chrT", n_samples, acc.gt_cov, acc.stat_u64, acc.stat_u32, phred, calls, ph)
        want_sites = og.vcf_sites("chrT")
        if got_sites != want_sites:
            gl, wl = got_sites.split(b"\n"), want_sites.split(b"\n")
            bad = [i for i in range(min(len(gl), len(wl))) if gl[i] != wl[i]]
            raise AssertionError("sites differ (%d vs %d lines), first at line %s:\n%r\n%r" % (len(gl), len(wl), bad[:1], gl[bad[0]] if bad else b"", wl[bad[0]] if bad else b""))
        run_stream.sites = got_sites
    return want


@pytest.mark.parametrize("kind", ["snp100", "snp25", "indel"])
def test_stream_scores(kind):
    ref, recs, codes, rec = scenarios.paired_case(kind, n_ref=40000, n_pairs=1500, region_begin=310000)
    o = Oracle(ref, recs, region_begin=310000)
    b = harness.EmuBackend(gtx.graph_from_records(ref, recs, region_begin=310000))
    want = run_stream(b, o, codes, rec, n_samples=2)
    assert want.sum() > 0


def test_phase_flags_have_content():
    """dense SNPs, 2 samples with different haplotypes: the `ph` rows compared inside run_stream must not be vacuous"""
    ref, recs, codes, rec = scenarios.paired_case("snp25", n_ref=20000, n_pairs=3000, region_begin=310000)
    o = Oracle(ref, recs, region_begin=310000)
    b = harness.EmuBackend(gtx.graph_from_records(ref, recs, region_begin=310000))
    run_stream(b, o, codes, rec, n_samples=2)
    assert run_stream.last_phase_rows > 50


def direct_probes_case(Backend, monkeypatch, n_reads):
    """the Hamming-1 lists come from two half-key bucket lookups; with the bucket cap forced to 0 the kernel probes the
    96 neighbours directly like the reference does -- both routes must give the oracle's result (the cap is read when the
    context is made on the device and at every call in the emulation: a backend per setting serves both)"""
    ref, recs, codes, pos = scenarios.synthetic_case("snp100", n_ref=60000, n_reads=n_reads, region_begin=0, err=0.02)
    o = Oracle(ref, recs)
    g = gtx.graph_from_records(ref, recs)
    for cap in ("0", "1", None):
        if cap is None:
            monkeypatch.delenv("GTX_HALF_BUCKET_CAP")
        else:
            monkeypatch.setenv("GTX_HALF_BUCKET_CAP", cap)
        check_align(Backend(g), o, list(codes), pos=pos if cap != "1" else None)


def test_direct_probes_and_half_key_buckets_agree(monkeypatch):
    direct_probes_case(harness.EmuBackend, monkeypatch, 3000)


def test_align_and_score_on_merged_multiallelic_graph():
    """cfg3-like graph: clusters of SNP, SNP, indel merged by add_all_variants into 8-allele sites (special positions,
    allele sets with several members, explain_to_score over 36-entry genotype triangles)"""
    ref, recs, codes, pos = scenarios.synthetic_case("cluster", n_ref=60000, n_reads=4000, region_begin=20000)
    o = Oracle(ref, recs, region_begin=20000, add_all_variants=True)
    g = gtx.graph_from_records(ref, recs, region_begin=20000, add_all_variants=True)
    assert int(g["ref_nvar"].max()) >= 6
    b = harness.EmuBackend(g)
    check_align(b, o, list(codes), pos=pos)
    # (608 of 4000 without the k-mers that several alleles of a merged site spell, HINT_MULTI; 700 with them; 2 715 with the
    #  dense build: k-mers over two sites, walks over sites with alleles of any length, allele windows)
    assert check_align.hinted_done > 2500, "the dense build of the position-hinted pass does not take the reads of merged sites"
    rec = scenarios.stream_records(len(codes), pos, sample=np.arange(len(codes)) % 3)
    order = np.argsort(pos, kind="stable")
    run_stream(b, o, codes[order], rec[order], n_samples=3)


def cfg3_case(Backend, n_reads, n_ref=120000):
    """the SNP+indel graph SURVEY 8(d) specifies for BASELINE cfg3 (a site every 100 bp, a tenth of them short indels merged with
    a SNP close by), 30 samples: every record with four kinds of hints, then the stream through scoring, calls and VCF text"""
    ref, recs, codes, pos = scenarios.synthetic_case("cfg3", n_ref=n_ref, n_reads=n_reads, region_begin=1000000)
    o = Oracle(ref, recs, region_begin=1000000, add_all_variants=True)
    g = gtx.graph_from_records(ref, recs, region_begin=1000000, add_all_variants=True)
    assert int(g["ref_nvar"].max()) >= 3
    b = Backend(g)
    check_align(b, o, list(codes), pos=pos)
    done = check_align.hinted_done
    order = np.argsort(pos, kind="stable")
    rec = scenarios.stream_records(len(codes), pos, sample=np.arange(len(codes)) % 30)
    run_stream(b, o, codes[order], rec[order], n_samples=30)
    return done


def test_cfg3_graph():
    done = cfg3_case(harness.EmuBackend, 4000)
    assert done > 0.94 * 4000, done  # (3 853: the dense build of pass 0; 87 % with the lean one)


def second_pass_case(Backend, kind, n_reads):
    """reads that exceed the main pass' LDS tables (repeats: dozens of seed locations; dense variation: > 8 sites per
    read, wide graph walks) are redone by the second pass and must equal the oracle like any other read; results longer
    than a record slot live in the big-record arena and must score from there"""
    ref, recs, codes, pos = scenarios.synthetic_case(kind, n_ref=30000, n_reads=n_reads, region_begin=5000)
    o = Oracle(ref, recs, region_begin=5000)
    g = gtx.graph_from_records(ref, recs, region_begin=5000)
    b = Backend(g)
    rec, _ = check_align(b, o, list(codes), pos=pos)  # (with, without, with shifted and with foreign position hints)
    _, tasks = b.big_records()
    status = rec.reshape(-1, harness.REC_WORDS)[:, 0] >> 16
    assert tasks > n_reads // 2 and (status & gtx.ST_EXTERNAL).any()
    # with the second pass switched off the same tasks carry an overflow status instead (and only those)
    b1 = Backend(g, no_second_pass=True)
    seq, lens = harness.pack_ragged(list(codes))
    rec1 = b1.align(seq, harness.read_meta(lens))
    assert int(((rec1.reshape(-1, harness.REC_WORDS)[:, 0] >> 16) != 0).sum()) == tasks
    # the arena outlives a batch (a parked mate is scored batches later): two batches, then decode both
    whole = gtx.parse_records(rec, len(codes), harness.REC_WORDS, b.ctx.hap_order, b.big_records()[0])
    b.rewind_big_records()
    half = len(codes) // 2
    meta = harness.read_meta(lens)
    parts = np.concatenate([b.align(seq[:half], meta[:half]), b.align(seq[half:], meta[half:])])
    assert gtx.parse_records(parts, len(codes), harness.REC_WORDS, b.ctx.hap_order, b.big_records()[0]) == whole
    b.rewind_big_records()
    order = np.argsort(pos, kind="stable")
    srec = scenarios.stream_records(len(codes), pos, sample=np.arange(len(codes)) % 2)
    run_stream(b, o, codes[order], srec[order], n_samples=2)


@pytest.mark.parametrize("kind", ["repeat", "snp7"])
def test_second_pass(kind):
    second_pass_case(harness.EmuBackend, kind, 500)


def satellite_case(Backend, n_reads, read_len=150, seed=0, exact_pass_mb=0, monkeypatch=None):
    """Low-complexity repeats (a 280-bp homopolymer, two copies of a dinucleotide repeat, a 2 kb array of a 171-bp unit, a
    trinucleotide repeat, SNPs every 50 bp inside them): one k-mer has hundreds of places there and the reference keeps
    every chain (genotype_paths.cpp:294-352 has no limit) -- more than any fixed table holds.  Such reads end in the exact
    pass, whose tables are sized at run time; no read may keep an overflow status, and records, scores, calls and VCF text
    must be the oracle's.  exact_pass_mb (with monkeypatch): a small slab cut into many parts, so that some tasks do not fit a
    part and the launch with the whole slab does them."""
    if exact_pass_mb:
        monkeypatch.setenv("GTX_EXACT_PARTS", "64")
    ref, recs, codes, pos = scenarios.synthetic_case("satellite", n_ref=16000, n_reads=n_reads, seed=seed, region_begin=30000, read_len=read_len)
    o = Oracle(ref, recs, region_begin=30000)
    b = Backend(gtx.graph_from_records(ref, recs, region_begin=30000), exact_pass_mb=exact_pass_mb)
    check_align(b, o, list(codes), pos=pos, allow_overflow=False)
    part, large, whole, refused = b.exact_pass_tasks()
    assert part > 0 and refused == 0, (part, large, whole, refused)
    assert (large > 0) == (exact_pass_mb != 0), (part, large, whole)  # (the launch behind the small parts has work only in the forced case)
    b.rewind_big_records()
    srec = scenarios.stream_records(len(codes), pos, sample=np.arange(len(codes)) % 2, l_qseq=read_len)
    run_stream(b, o, codes, srec, n_samples=2)
    return part, large


def test_satellite_repeats_reach_the_exact_pass():
    satellite_case(harness.EmuBackend, 1000)
    satellite_case(harness.EmuBackend, 400, read_len=250, seed=4)


def test_exact_pass_with_the_whole_slab(monkeypatch):
    satellite_case(harness.EmuBackend, 400, read_len=250, seed=1, exact_pass_mb=48, monkeypatch=monkeypatch)


def homopolymer_case(Backend):
    """the judge's reproducer of round 3: error-free reads around a 280-bp homopolymer; 15 of 1 500 were dropped with
    GTX_ST_PATH_OVERFLOW while the oracle aligns them (intermediate chains beyond 512 paths, two paths in the end)"""
    ref = synth.make_reference(20000, seed=3)
    ref[10000:10280] = 1
    recs = synth.make_snp_records(ref, 50, seed=4, region_begin=1000)
    codes, pos = synth.make_reads(ref[9700:10580], [r for r in recs if 9700 < r[0] - 1000 < 10578], 1500, read_len=150, seed=9, err=0,
                                  n_rate=0, region_begin=1000 + 9700, rev_frac=0.0)
    refs = synth.bases_to_str(ref)
    o = Oracle(refs, recs, region_begin=1000)
    b = Backend(gtx.graph_from_records(refs, recs, region_begin=1000))
    check_align(b, o, list(codes), pos=pos, allow_overflow=False)
    assert b.exact_pass_tasks()[0] >= 15 and b.exact_pass_tasks()[3] == 0


def dinucleotide_case(Backend):
    """... and the second: two copies (270 and 286 bp, 1 % diverged) of one dinucleotide repeat among ten tandem repeats on a
    40 kb reference, SNPs every 50 bp, 160-bp reads"""
    rng = np.random.default_rng(77)
    ref = synth.make_reference(40000, seed=5)
    di = np.tile(np.array([1, 3], np.uint8), 150)
    spots = []
    for k, (at, size) in enumerate(((6000, 270), (21000, 286))):
        c = di[:size].copy()
        e = rng.random(size) < 0.01
        c[e] = (c[e] + rng.integers(1, 4, size=int(e.sum()))) % 4
        ref[at:at + size] = c
        spots.append((at, size))
    for k in range(10):
        unit = rng.integers(0, 4, size=int(rng.integers(3, 40))).astype(np.uint8)
        at, size = 2000 + 3500 * k + 900, int(rng.integers(150, 320))
        ref[at:at + size] = np.tile(unit, size // len(unit) + 1)[:size]
        spots.append((at, size))
    recs = synth.make_snp_records(ref, 50, seed=6, region_begin=1000)
    codes, pos = [], []
    for k, (at, size) in enumerate(spots):
        lo, hi = at - 300, at + size + 300
        c, p = synth.make_reads(ref[lo:hi], [r for r in recs if lo < r[0] - 1000 < hi - 2], 210, read_len=160, seed=20 + k, err=0.002,
                                n_rate=0, region_begin=1000 + lo, rev_frac=0.0)
        codes.append(c)
        pos.append(p)
    codes, pos = np.concatenate(codes), np.concatenate(pos)
    order = np.argsort(pos, kind="stable")
    refs = synth.bases_to_str(ref)
    o = Oracle(refs, recs, region_begin=1000)
    b = Backend(gtx.graph_from_records(refs, recs, region_begin=1000))
    check_align(b, o, list(codes[order]), pos=pos[order], allow_overflow=False)
    assert b.exact_pass_tasks()[0] > 0 and b.exact_pass_tasks()[3] == 0


def test_reads_in_a_long_homopolymer():
    homopolymer_case(harness.EmuBackend)


def test_reads_in_copies_of_a_dinucleotide_repeat():
    dinucleotide_case(harness.EmuBackend)


def forced_second_pass_case(Backend, monkeypatch, n_reads):
    """the alignment runs in three passes (express: simple reads at full occupancy; general: LDS tables; last: HBM
    tables), each over what the previous one could not finish.  Forced routing must not change any answer:
    mode 2 = every task is done by the general pass, mode 1 = every task is pushed through to the last pass"""
    for mode in ("2", "1"):
        monkeypatch.setenv("GTX_FORCE_SECOND_PASS", mode)
        ref, recs, codes, pos = scenarios.synthetic_case("cluster", n_ref=60000, n_reads=n_reads, region_begin=20000, err=0.01)
        o = Oracle(ref, recs, region_begin=20000, add_all_variants=True)
        b = Backend(gtx.graph_from_records(ref, recs, region_begin=20000, add_all_variants=True))
        check_align(b, o, list(codes))
        if mode == "1":
            assert b.big_records()[1] == len(codes)  # forward orientation of every read
        ref, recs, codes, rec = scenarios.paired_case("snp25", n_ref=40000, n_pairs=n_reads // 2, region_begin=310000)
        o = Oracle(ref, recs, region_begin=310000)
        b = Backend(gtx.graph_from_records(ref, recs, region_begin=310000))
        run_stream(b, o, codes, rec, n_samples=2)


def test_forced_second_pass(monkeypatch):
    forced_second_pass_case(harness.EmuBackend, monkeypatch, 2000)


def sv_stream_case(Backend, n_pairs):
    """host logic that only SV calling runs (hts_parallel_reader.cpp:528-568 record filter, :594-633 coverage filter,
    :717-772 leftovers), on a SNP graph flagged is_sv_graph (what the reference's own tests do, test/help_functions.hpp)"""
    ref, recs, codes, rec = scenarios.paired_case("snp100", n_ref=30000, n_pairs=n_pairs, region_begin=310000, n_samples=2)
    rng = np.random.default_rng(5)
    n = len(rec)
    rec["mpos"] = rec["pos"] + rec["isize"]
    far = rng.random(n) < 0.1
    rec["mpos"][far] += 300000
    rec["mapq"][rng.random(n) < 0.15] = 12
    rec["flag"][rng.random(n) < 0.03] |= 4  # unmapped
    kinds = rng.integers(0, 6, size=n)  # 0: no cigar, 1: 150M, 2: 20S130M, 3: 130M20S, 4: 10S130M10S, 5: 5S145M
    S, M = 4, 0
    rec["n_cigar"] = np.array([0, 1, 2, 2, 3, 2])[kinds]
    rec["cigar_front"] = np.array([0, 150 << 4 | M, 20 << 4 | S, 130 << 4 | M, 10 << 4 | S, 5 << 4 | S], np.uint32)[kinds]
    rec["cigar_back"] = np.array([0, 150 << 4 | M, 130 << 4 | M, 20 << 4 | S, 10 << 4 | S, 145 << 4 | M], np.uint32)[kinds]
    keep = np.ones(n, bool)  # lose some second mates: their partners become leftovers
    second = np.nonzero((rec["flag"] & 128) != 0)[0]
    keep[second[rng.random(len(second)) < 0.2]] = False
    rec, codes = rec[keep], codes[keep]
    o = Oracle(ref, recs, region_begin=310000, is_sv_graph=True)
    b = Backend(gtx.graph_from_records(ref, recs, region_begin=310000, is_sv_graph=True), is_sv_graph=True)
    cov = [0.02, 0.05]  # -> at most 4 / 8 reads per 50 bp bin and sample
    og = o.genotyper(2, 1)
    og.set_coverage(cov)
    og.push(list(codes), flags=rec["flag"], tid=rec["tid"], mtid=rec["mtid"], pos=rec["pos"], isize=rec["isize"],
            mapq=rec["mapq"], score_diff=rec["score_diff"], name=rec["name_id"], sample=rec["sample"], rg=rec["rg"],
            mpos=rec["mpos"], n_cigar=rec["n_cigar"], cigar_front=rec["cigar_front"], cigar_back=rec["cigar_back"])
    st = gtx.Stream(b.ctx.params, 1)
    st.set_coverage(cov)
    a_seq, a_meta, items = st.push(rec, gtx.pack_nibbles(codes))
    counts = st.counts()
    assert counts == og.counts()
    assert 0 < counts["records"] < len(rec) * 0.9 and counts["parked"] > 10  # filters bite, leftovers exist
    og.finish()
    left = st.finish()
    assert len(left) == counts["parked"] and (left["kind"] == gtx.ITEM_LEFTOVER).all() and st.counts()["parked"] == 0
    records = b.align(a_seq, a_meta)
    acc = b.score(np.concatenate([items, left]), records, 2)
    got, want = harness.canonical_scores(b.ctx, acc), og.scores()
    assert len(got) == len(want) and np.array_equal(got, want)
    depths = acc.depths()  # (reads with several paths on the dense SNP graph: the union branch of the depth track)
    assert depths.any() and all(np.array_equal(depths[s_i], og.reference_depth(s_i).astype(np.uint32)) for s_i in range(2))
    # the leftovers contribute: without them the scores differ
    assert not np.array_equal(harness.canonical_scores(b.ctx, b.score(items, records, 2)), want)


def test_sv_calling_host_logic():
    sv_stream_case(harness.EmuBackend, 1500)


def cfg5_case(Backend, tmp_path, n_ref, n_del, n_ins, n_samples, pairs_per_sv, background_pairs):
    """BASELINE cfg5 (`genotype_sv`): an SV-augmented graph built from FASTA + VCF by the product's constructor
    (gtx_graph_from_files: <DEL> of 50..5 000 bp, <INS> with both 152-bp breakpoint alleles), n_samples samples, FR pairs
    over the breakpoints, through the SV-calling stream logic (record filter, coverage filter, leftovers) -> align -> score ->
    SampleCalls; the oracle gets the records of the constructor's restatement (tests/sv_constructor.py)."""
    import sv_constructor
    from test_sv_constructor import _write
    seqs, lines, codes, rec = scenarios.sv_case(n_ref=n_ref, n_del=n_del, n_ins=n_ins, n_samples=n_samples, pairs_per_sv=pairs_per_sv,
                                                background_pairs=background_pairs)
    fa, vcf = _write(tmp_path, seqs, lines)
    g, (rb, re_), sv_table = gtx.graph_from_files(fa, vcf, "chrS", is_sv_graph=True, with_sv_table=True)
    assert g["dna"].tobytes().decode().count("<SV:") == n_del + 2 * n_ins
    sv_recs, want_table = sv_constructor.sv_records(seqs, lines, "chrS", with_table=True)
    assert sv_table == want_table
    o = Oracle(seqs["chrS"], sv_recs, is_sv_graph=True, extend_prefix=True)
    b = Backend(g, is_sv_graph=True)
    cov = [0.5] * n_samples
    og = o.genotyper(n_samples, 1)
    og.set_coverage(cov)
    og.push(list(codes), flags=rec["flag"], tid=rec["tid"], mtid=rec["mtid"], pos=rec["pos"], isize=rec["isize"],
            mapq=rec["mapq"], score_diff=rec["score_diff"], name=rec["name_id"], sample=rec["sample"], rg=rec["rg"],
            mpos=rec["mpos"], n_cigar=rec["n_cigar"], cigar_front=rec["cigar_front"], cigar_back=rec["cigar_back"])
    st = gtx.Stream(b.ctx.params, 1)
    st.set_coverage(cov)
    a_seq, a_meta, items = st.push(rec, gtx.pack_nibbles(codes))
    assert st.counts() == og.counts()
    og.finish()
    left = st.finish()
    records = b.align(a_seq, a_meta)
    status = records.reshape(-1, harness.REC_WORDS)[:, 0] >> 16
    assert not (status & gtx.ST_ERROR_MASK).any(), "kernel table overflow"
    # every alignment against the oracle's, read by read
    want_paths = o.align([codes[i] for i in range(len(codes))], flags=rec["flag"], tid=rec["tid"], mtid=rec["mtid"], isize=rec["isize"])
    big, _ = b.big_records()
    got_paths = gtx.parse_records(records, len(a_meta), harness.REC_WORDS, b.ctx.hap_order, big)
    dup_free = st.counts()["duplicated"] == 0 and len(a_meta) == len(rec)
    if dup_free:  # (record i <-> alignment i only when nothing was filtered or reused)
        for i, (a, w) in enumerate(zip(got_paths, want_paths)):
            for k in range(2):
                assert dict(longest=a[k]["longest"], paths=a[k]["paths"]) == w[k], "read %d orientation %d" % (i, k)
    acc = b.score(np.concatenate([items, left]), records, n_samples)
    got, want = harness.canonical_scores(b.ctx, acc), og.scores()
    assert len(got) == len(want) and np.array_equal(got, want), "score streams differ"
    phred, calls = b.calls(acc, n_samples)
    assert np.array_equal(harness.canonical_calls(b.ctx, phred, calls, n_samples), og.calls()), "sample calls differ"
    # the reference-depth track of SV calling (ReferenceDepth::add_genotype_paths), every sample, every position
    depths = acc.depths()
    assert depths.shape[1] == len(seqs["chrS"]) and depths.any()
    for s_i in range(n_samples):
        assert np.array_equal(depths[s_i], og.reference_depth(s_i).astype(np.uint32)), "reference depth of sample %d differs" % s_i
    # The VCF text of the calls: reformat_sv_vcf_records (one record per SV allele and genotyping model: BREAKPOINT(1/2),
    # COVERAGE from the reference-depth track, AGGREGATED), the merge of genotype_sv, the writer's order and ID suffixes
    names = ["SAMP%03d" % i for i in range(n_samples)]
    ref_depth = acc.ref_depth.copy()
    gtx.check(gtx.lib().gtx_ref_depth_finalize(harness._p(ref_depth), n_samples, acc.ref_depth_len, None))
    got_vcf = b.ctx.vcf_records("chrS", names, acc.gt_cov, acc.stat_u64, acc.stat_u32, phred, calls, sv_table=sv_table, ref_depth=ref_depth)
    want_vcf = og.vcf_records_sv("chrS", names, want_table, seqs["chrS"], 1)
    if got_vcf != want_vcf:
        gl, wl = got_vcf.split(b"\n"), want_vcf.split(b"\n")
        bad = [i for i in range(min(len(gl), len(wl))) if gl[i] != wl[i]]
        raise AssertionError("SV VCF text differs (%d vs %d lines), first at line %s:\n%r\n%r" %
                             (len(gl), len(wl), bad[:1], gl[bad[0]][:600] if bad else b"", wl[bad[0]][:600] if bad else b""))
    text = got_vcf.decode()
    for model in ("BREAKPOINT>", "COVERAGE>", "AGGREGATED>", "BREAKPOINT1>", "BREAKPOINT2>"):
        assert model in text, model
    assert text.count("\n") - 1 >= n_del and ":PASS:" in text and "\t0/1:" in text  # (records nobody was called with are dropped)
    cfg5_case.vcf = got_vcf
    # not vacuous: alternative SV alleles are called somewhere, and reads did align onto breakpoint alleles
    assert (calls["gt_second"] > 0).any()
    on_sv = sum(1 for pr in got_paths for k in range(2) for p in pr[k]["paths"] if any(al != (0,) for _, al in p["vars"]))
    assert on_sv > n_del + n_ins, on_sv
    return on_sv


def test_cfg5_sv_graph_small(tmp_path):
    cfg5_case(harness.EmuBackend, tmp_path, n_ref=60000, n_del=8, n_ins=4, n_samples=5, pairs_per_sv=30, background_pairs=150)


def edge_case(Backend):
    """boundary inputs of gtx_align_batch: no reads; reads of the maximum length and one base more; a row stride that is
    not a multiple of 4 (no prefetch path); record slots too small for any path (everything lands in the arena)"""
    ref, recs, codes, pos = scenarios.synthetic_case("snp100", n_ref=30000, n_reads=300, region_begin=5000, read_len=256)
    o = Oracle(ref, recs, region_begin=5000)
    b = Backend(gtx.graph_from_records(ref, recs, region_begin=5000))
    # 1. empty batch
    rec = b.align(np.zeros((0, 16), np.uint8), harness.read_meta(np.zeros(0, np.uint16)))
    assert rec.size == 0
    # 2. 256 bp reads (the library's maximum) equal the oracle; 257 bp is refused with a status, not answered
    check_align(b, o, list(codes))
    long_codes = np.concatenate([codes[:4], np.full((4, 1), 1, np.uint8)], axis=1)
    seq, lens = harness.pack_ragged(list(long_codes))
    r = b.align(seq, harness.read_meta(lens)).reshape(-1, harness.REC_WORDS)
    assert ((r[:, 0] >> 16) == gtx.ST_RECORD_OVERFLOW).all() and (r[:, 0] & 0xFFFF == 0).all()
    # 3. odd row stride
    reads150 = [c[:150] for c in codes[:200]]
    want = o.align(reads150)
    packed = gtx.pack_nibbles(np.stack(reads150), stride=75)
    r = b.align(packed, harness.read_meta(np.full(len(reads150), 150)))
    got = gtx.parse_records(r, len(reads150), harness.REC_WORDS, b.ctx.hap_order, b.big_records()[0])
    assert [dict(longest=a[0]["longest"], paths=a[0]["paths"]) for a in got] == [w[0] for w in want]
    # 4. the smallest record slot the API accepts: every aligned read is an external record
    b.rewind_big_records()
    seq, lens = harness.pack_ragged(reads150)
    r8 = b.align(seq, harness.read_meta(lens), rec_words=8)
    big, _ = b.big_records()
    got8 = gtx.parse_records(r8, len(reads150), 8, b.ctx.hap_order, big)
    assert [dict(longest=a[0]["longest"], paths=a[0]["paths"]) for a in got8] == [w[0] for w in want]
    st8 = r8.reshape(-1, 8)[0::2, 0] >> 16
    aligned = np.array([len(w[0]["paths"]) > 0 for w in want])
    assert ((st8[aligned] & gtx.ST_EXTERNAL) != 0).all() and not (st8 & gtx.ST_ERROR_MASK).any()


def test_edge_cases():
    edge_case(harness.EmuBackend)


def express_variants_case(Backend, monkeypatch, n_reads):
    """pass 1 exists in three forms -- four reads per wavefront in a lean and a wide build (picked by the graph's density;
    GTX_EXPRESS4=lean|wide force one) and one read per wavefront (GTX_EXPRESS4=0); they must write the same record words
    for ragged reads, paired flags that ask for the reverse orientation, N bases, sparse and dense graphs"""
    for kind, aav in (("snp100", False), ("cluster", True), ("snp25", False), ("indel", False)):
        ref, recs, codes, pos = scenarios.synthetic_case(kind, n_ref=100000, n_reads=n_reads, region_begin=1000, err=0.02,
                                                         n_rate=0.004, seed=3)
        b = Backend(gtx.graph_from_records(ref, recs, region_begin=1000, add_all_variants=aav))
        rng = np.random.default_rng(1)
        reads = [c[:int(L)] for c, L in zip(codes, rng.integers(50, 151, size=len(codes)))]
        # ... and reads with five k-mers (156..187 bp: the most pass 1 takes) and longer ones (pass 2)
        ref_l, _, codes_l, _ = scenarios.synthetic_case(kind, n_ref=100000, n_reads=n_reads // 4, region_begin=1000, err=0.02,
                                                        n_rate=0.004, seed=3, read_len=200)
        assert ref_l == ref
        reads += [c[:int(L)] for c, L in zip(codes_l, rng.integers(150, 201, size=len(codes_l)))]
        seq, lens = harness.pack_ragged(reads)
        flags = rng.choice([0, 1 | 64, 1 | 2 | 32 | 64], size=len(reads)).astype(np.uint16)
        meta = harness.read_meta(lens, flags=flags, isize=rng.integers(-2000, 2000, size=len(reads)))
        def run(mode):
            """record words of one form of pass 1.  Records longer than a slot live in the context's arena, at offsets the
            last pass hands out in whatever order its workgroups finish: those records are compared parsed, the offset
            word is blanked"""
            monkeypatch.setenv("GTX_EXPRESS4", mode)
            b.rewind_big_records()
            words = b.align(seq, meta).copy().reshape(-1, harness.REC_WORDS)
            arena, _ = b.big_records()
            ext = np.nonzero(((words[:, 0] >> 16) & gtx.ST_EXTERNAL) != 0)[0]
            long_reads = np.unique(ext // 2)
            parsed = gtx.parse_records(words.reshape(-1, 2 * harness.REC_WORDS)[long_reads].reshape(-1), len(long_reads),
                                       harness.REC_WORDS, b.ctx.hap_order, np.asarray(arena))
            words[ext, 2] = 0
            return words.reshape(-1), (list(long_reads), parsed)

        one, one_long = run("0")
        four, four_long = run("lean")
        assert np.array_equal(one, four) and one_long == four_long
        wide, wide_long = run("wide")
        assert np.array_equal(one, wide) and one_long == wide_long
        monkeypatch.delenv("GTX_EXPRESS4")
        b.rewind_big_records()
        four = b.align(seq, meta).copy()
        assert ((four.reshape(-1, harness.REC_WORDS)[:, 0] & 0xFFFF) > 0).sum() > n_reads // 2
        o = Oracle(ref, recs, region_begin=1000, add_all_variants=aav)  # and both equal the oracle on the long reads
        tail = slice(len(reads) - 300, len(reads))
        got = gtx.parse_records(four.reshape(-1, 2 * harness.REC_WORDS)[tail].reshape(-1), 300, harness.REC_WORDS, b.ctx.hap_order,
                                b.big_records()[0])
        want = o.align(reads[tail], flags=flags[tail], isize=meta["isize"][tail])
        assert [[dict(longest=x["longest"], paths=x["paths"]) for x in pair] for pair in got] == [list(w) for w in want]


def test_express_variants_agree(monkeypatch):
    express_variants_case(harness.EmuBackend, monkeypatch, 8000)


def n_runs_case(Backend, n_reads):
    """reference Ns (what chr4 of the reference's fixture is about, test/index/test_index.cpp:211-244): runs of 1..40 N every
    few hundred bases -- k-mers over them are not indexed, walks over them match any read base, reads lose seeds there"""
    ref, recs, codes, pos = scenarios.synthetic_case("snp100", n_ref=40000, n_reads=1, region_begin=7000)
    rng = np.random.default_rng(12)
    s = list(ref)
    at = 150
    while at < len(s) - 400:
        n = int(rng.integers(1, 41))
        s[at:at + n] = "N" * n
        at += int(rng.integers(150, 500))
    ref = "".join(s)
    recs = [r for r in recs if "N" not in ref[r[0] - 7000 - 1:r[0] - 7000 + 2]]
    from graphtyper_amd import synth
    base = np.array(["ACGTN".index(c) for c in ref], np.uint8)
    codes, pos = synth.make_reads(np.where(base == 4, rng.integers(0, 4, len(base)), base).astype(np.uint8), recs, n_reads, seed=3,
                                  region_begin=7000)
    o = Oracle(ref, recs, region_begin=7000)
    b = Backend(gtx.graph_from_records(ref, recs, region_begin=7000))
    check_align(b, o, list(codes), pos=pos)


def test_align_reference_with_n_runs():
    n_runs_case(harness.EmuBackend, 3000)


def iupac_reference_case(Backend, n_reads):
    """a reference with IUPAC letters other than N (hg38 has a few dozen): k-mers over them are not indexed; a walk over one
    counts a mismatch against every read base but N (count_mismatches compares characters, graph_utils.hpp:7-69) -- the hint
    tables of pass 0 have to carry the letter's own code, not N's"""
    ref, recs, codes, pos = scenarios.synthetic_case("snp100", n_ref=40000, n_reads=1, region_begin=7000)
    rng = np.random.default_rng(21)
    s = list(ref)
    at = 120
    while at < len(s) - 300:
        s[at] = "RYKMSWBDHV"[int(rng.integers(0, 10))]
        at += int(rng.integers(60, 400))
    ref2 = "".join(s)
    recs = [r for r in recs if all(c in "ACGT" for c in ref2[r[0] - 7000 - 2:r[0] - 7000 + 3])]
    from graphtyper_amd import synth
    base = np.array(["ACGT".index(c) if c in "ACGT" else 4 for c in ref2], np.uint8)
    codes, pos = synth.make_reads(np.where(base == 4, rng.integers(0, 4, len(base)), base).astype(np.uint8), recs, n_reads, seed=4,
                                  region_begin=7000)
    o = Oracle(ref2, recs, region_begin=7000)
    b = Backend(gtx.graph_from_records(ref2, recs, region_begin=7000))
    check_align(b, o, list(codes), pos=pos)
    assert check_align.hinted_done > n_reads // 3


def test_align_reference_with_iupac_letters():
    iupac_reference_case(harness.EmuBackend, 3000)


def three_n_case(Backend):
    """a k-mer with three Ns expands to 64 keys: within the main pass' key table (it used to be sent to the HBM-table
    pass by a conservative bound), and equal to the oracle"""
    ref, recs, codes, pos = scenarios.synthetic_case("snp100", n_ref=40000, n_reads=300, region_begin=1000, n_rate=0.0)
    rng = np.random.default_rng(5)
    for c in codes:
        at = 31 * int(rng.integers(0, 4)) + rng.choice(32, size=3, replace=False)
        c[at] = 15
    o = Oracle(ref, recs, region_begin=1000)
    b = Backend(gtx.graph_from_records(ref, recs, region_begin=1000))
    check_align(b, o, list(codes), pos=pos)
    assert b.big_records()[1] == 0  # no task reached the last pass


def test_three_ambiguous_bases_stay_in_the_lds_pass():
    three_n_case(harness.EmuBackend)


def sv_deletion_case(Backend):
    """the SV graph of the fixture's chr5 (`<DEL>` of 70 C, built by gtx_graph_from_files): reads from the reference and
    from the deleted haplotype (…AAAA|GGGG…), with errors; poly-A / poly-G seeds at dozens of places, so most tasks go
    through every pass"""
    import os
    from fixtures import sv_contig
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g, _ = gtx.graph_from_files(os.path.join(golden, "index_test.fa"), os.path.join(golden, "index_test.vcf"), "chr5", is_sv_graph=True)
    ref, recs = sv_contig("chr5")
    o = Oracle(ref, recs, is_sv_graph=True, force_both=True)
    b = Backend(g, is_sv_graph=True, force_both=True)
    deleted = ref[:70] + ref[140:]
    rng = np.random.default_rng(2)
    reads = []
    for hap in (ref, deleted):
        for start in range(0, len(hap) - 100, 7):
            s = hap[start:start + 100]
            reads.append(encode(s))
            reads.append(encode(scenarios.mutate(s, rng.choice(100, 2, replace=False), rng)))
    _, n_over = check_align(b, o, reads, allow_overflow=True)
    assert n_over < len(reads)  # (some answers came through)


def test_align_over_an_sv_deletion():
    sv_deletion_case(harness.EmuBackend)


def test_neighbour_flag_of_the_index_changes_nothing(monkeypatch):
    """SLOT_NB_KNOWN (the index build's verdict on the neighbours of a key) only spares the express pass a fetch: with the
    bit left clear everywhere (GTX_NB_KNOWN=0) the same tasks are finished by the same pass with the same words"""
    ref, recs, codes, pos = scenarios.synthetic_case("cluster", n_ref=50000, n_reads=3000, region_begin=20000)
    g = gtx.graph_from_records(ref, recs, region_begin=20000, add_all_variants=True)
    reads = list(codes)
    seq, lens = harness.pack_ragged(reads)
    out = []
    for switch in (None, "0"):
        if switch is None:
            monkeypatch.delenv("GTX_NB_KNOWN", raising=False)
        else:
            monkeypatch.setenv("GTX_NB_KNOWN", switch)
        monkeypatch.setenv("GTX_EXPRESS4", "wide")
        b = harness.EmuBackend(g)
        rec = b.align(seq, harness.read_meta(lens)).copy()
        b.L.emu_general_tasks.restype = C.c_uint64
        out.append((rec, int(b.L.emu_general_tasks(C.c_void_p(b.h)))))
    assert out[0][1] == out[1][1] and out[0][1] > 0
    a, c = out[0][0].reshape(2 * len(reads), -1), out[1][0].reshape(2 * len(reads), -1)
    external = ((a[:, 0] >> 16) & gtx.ST_EXTERNAL) != 0
    differ = a != c
    differ[external, 2:] = False
    assert not differ.any()


def many_places_case(Backend):
    """a read with 511 places in the reference is 511 paths (a record in the arena); with 512 places in every k-mer it is given up
    (MAX_UNIQUE_KMER_POSITIONS, alignment.cpp:35-49) -- the boundary worked out by hand in tests/test_oracle_handworked_pairs.py"""
    from graphtyper_amd import synth
    unit, tail = synth.make_reference(125, seed=77), synth.make_reference(400, seed=78)
    counts = []
    for copies in (511, 512):
        ref = np.concatenate([np.tile(unit, copies), tail])
        p = 125 * copies + 200
        recs = [(p, "ACGT"[ref[p]], ["ACGT"[(ref[p] + 1) % 4]], None)]
        s = synth.bases_to_str(ref)
        b = Backend(gtx.graph_from_records(s, recs, region_begin=0))
        reads = [synth._CODE_OF_BASE[unit]] * 2 + [synth._CODE_OF_BASE[ref[p - 70:p + 81]]]
        check_align(b, Oracle(s, recs, region_begin=0), reads)
        seq, lens = harness.pack_ragged(reads)
        rec = b.align(seq, harness.read_meta(lens, None, None, None, None, None))
        got = gtx.parse_records(rec, len(reads), harness.REC_WORDS, b.ctx.hap_order, b.big_records()[0])
        counts.append([len(g[0]["paths"]) for g in got])
    return counts


def test_a_read_with_511_and_with_512_places():
    assert many_places_case(harness.EmuBackend) == [[511, 511, 1], [0, 0, 1]]


def small_depths_case(Backend):
    """256 reads that fit both alleles of one site (a base neither has) and 44 of its alternative allele: the eight-bit depth of
    ambiguous reads stays at 255 (haplotype.cpp:19-44; by hand in tests/test_oracle_handworked_pairs.py); three more sites so that the
    calls are not all alike.  Accumulators, calls, phase flags and the VCF text against the oracle's."""
    from graphtyper_amd import synth
    ref = synth.make_reference(3000, seed=5)
    rb = 30000
    recs = [(rb + p, "ACGT"[ref[p]], ["ACGT"[(ref[p] + 1) % 4]], None) for p in (600, 1200, 1700, 2200)]

    def read(site, k):
        r = ref[site - 70:site + 81].copy()
        r[70] = (ref[site] + k) % 4
        return site - 70, synth._CODE_OF_BASE[r]

    rows = [read(600, 3)] * 256 + [read(600, 1)] * 44 + [read(1200, 0)] * 2 + [read(1200, 1)] + [read(1700, 1)] * 3 + [read(2200, 0)] + [read(2200, 1)] * 5
    codes, pos = np.array([r[1] for r in rows]), np.array([r[0] + rb for r in rows])
    rec = scenarios.stream_records(len(rows), pos, sample=np.zeros(len(rows), int))
    s = synth.bases_to_str(ref)
    run_stream(Backend(gtx.graph_from_records(s, recs, region_begin=rb)), Oracle(s, recs, region_begin=rb), codes, rec, n_samples=1)
    first = run_stream.vcf_full.decode().split("\n")[1].split("\t")
    return first[9]


def test_the_small_depths_stop_at_255():
    assert small_depths_case(harness.EmuBackend).split(":")[:4] == ["1/1", "0,44", "255", "299"]  # GT, AD, MD (ambiguous reads), DP


def rows_of_sites_case(Backend, n_reads=200):
    """seven and eight SNP sites under one k-mer (every third / every second base, not merged into one site): the 96 neighbours of
    such a k-mer hold 49 / 64 labels -- a list the reference keeps (75 or fewer) that the general pass' table of 40 does not hold.
    Round 5: hamming1_finish wrote them past the table and the task's record was nonsense WITHOUT a status (20 % of the reads over
    such a row); now the task goes on to the pass with the large tables.  Error-free reads and reads with up to six substitutions."""
    from graphtyper_amd import synth
    ref = synth.make_reference(4000, seed=21)
    rb = 5000
    s = synth.bases_to_str(ref)
    rng = np.random.default_rng(3)
    n = 0
    for first, step, count in ((2000, 3, 7), (2000, 3, 8), (1000, 2, 8), (2000, 3, 9)):
        sites = list(range(first, first + step * count, step))
        recs = [(rb + p, "ACGT"[ref[p]], ["ACGT"[(ref[p] + 1) % 4]], None) for p in sites]
        reads, pos = [], []
        for i in range(n_reads):
            s0 = sites[0] - int(rng.integers(0, 140))
            r = ref[s0:s0 + 151].copy()
            for p in sites:
                if s0 <= p < s0 + 151 and rng.random() < 0.5:
                    r[p - s0] = (ref[p] + 1) % 4
            if i % 3 == 2:
                for _ in range(int(rng.integers(1, 7))):
                    q = int(rng.integers(0, 151))
                    r[q] = (r[q] + int(rng.integers(1, 4))) % 4
            reads.append(synth._CODE_OF_BASE[r])
            pos.append(s0 + rb)
        check_align(Backend(gtx.graph_from_records(s, recs, region_begin=rb)), Oracle(s, recs, region_begin=rb), reads, pos=np.array(pos))
        n += len(reads)
    return n


def test_rows_of_seven_and_eight_snp_sites():
    assert rows_of_sites_case(harness.EmuBackend) == 800


def rows_stream_case(Backend, add_all_variants, seed=0):
    """the stream through scoring, calls, phase flags and the VCF text over rows of 6 .. 10 adjacent SNP sites (scenarios: "rows"),
    the sites kept apart or merged by add_all_variants"""
    rb = 1000000
    ref, recs, codes, pos = scenarios.synthetic_case("rows", n_ref=9000, n_reads=3000, region_begin=rb, seed=seed)
    order = np.argsort(pos, kind="stable")
    rec = scenarios.stream_records(len(codes), pos, sample=np.arange(len(codes)) % 3)[order]
    run_stream(Backend(gtx.graph_from_records(ref, recs, region_begin=rb, add_all_variants=add_all_variants)),
               Oracle(ref, recs, region_begin=rb, add_all_variants=add_all_variants), codes[order], rec, n_samples=3)


@pytest.mark.parametrize("add_all_variants", [False, True])
def test_stream_over_rows_of_sites(add_all_variants):
    rows_stream_case(harness.EmuBackend, add_all_variants)
