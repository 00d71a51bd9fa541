"""The oracle against ground truth that does not come from the reference's code: reads cut from a known haplotype of a
SNP-only graph must come back as ONE path over exactly the bases they were cut from, without mismatches, carrying at
every SNP site the allele of that haplotype.  (The reference has no active tests for seed chaining and graph walks --
test/typer/test_genotype_path.cpp is commented out -- so this is the independent check the oracle gets for them; the
coordinates are the reference's: 1-based contig positions, test/index/test_index.cpp:183-190.)"""
import numpy as np
import pytest

from graphtyper_amd import synth
from oracle_lib import Oracle

CODE = np.array([1, 2, 4, 8], np.uint8)


@pytest.mark.parametrize("every,read_len", [(1000, 150), (100, 150), (37, 150), (100, 101), (37, 63)])
def test_error_free_reads_come_back_as_their_haplotype(every, read_len):
    n_ref, region_begin, n_reads = 30000, 700000, 300
    rng = np.random.default_rng(every + read_len)
    ref = synth.make_reference(n_ref, seed=every)
    recs = synth.make_snp_records(ref, every, seed=3, region_begin=region_begin)
    pos = np.array([p - region_begin for p, _, _, _ in recs])
    alt = np.array(["ACGT".index(a[0]) for _, _, a, _ in recs], np.uint8)
    take = rng.random(len(recs)) < 0.5
    hap1 = ref.copy()
    hap1[pos[take]] = alt[take]
    haps = [ref, hap1]
    which = rng.integers(0, 2, size=n_reads)
    start = rng.integers(1, n_ref - read_len, size=n_reads)  # (position 0 would be contig position region_begin: 0-based edge)
    reads = [CODE[haps[h][s:s + read_len]] for h, s in zip(which, start)]
    o = Oracle(synth.bases_to_str(ref), recs, region_begin=region_begin)
    got = o.align(reads)
    for i, (fwd, _rev) in enumerate(got):
        assert len(fwd["paths"]) == 1, "read %d: %r" % (i, fwd)
        p = fwd["paths"][0]
        first = region_begin + int(start[i]) + 1  # 1-based contig position of the read's first base
        assert (p["start"], p["end"], p["rs"], p["re"], p["mm"]) == (first, first + read_len - 1, 0, read_len - 1, 0), (i, p)
        assert fwd["longest"] == read_len
        inside = np.nonzero((pos >= start[i]) & (pos < start[i] + read_len))[0]
        want = {int(region_begin + pos[k] + 1): (1 if (which[i] == 1 and take[k]) else 0) for k in inside}
        seen = {order: nums for order, nums in p["vars"]}
        assert set(seen) == set(want), (i, seen, want)
        for order, allele in want.items():
            assert seen[order] == (allele,), (i, order, seen[order], allele)


def test_genotype_calls_recover_the_sample():
    """scoring + SampleCall against ground truth: a diploid sample (haplotype 0 = reference, haplotype 1 carries a known
    half of the SNPs), error-free reads at ~40x: every site must be called 0/1 where haplotype 1 carries the alt allele
    and 0/0 elsewhere, with the depth split between the alleles"""
    n_ref, region_begin, read_len = 20000, 300000, 150
    n_reads = 40 * n_ref // read_len
    rng = np.random.default_rng(9)
    ref = synth.make_reference(n_ref, seed=77)
    recs = synth.make_snp_records(ref, 100, seed=5, region_begin=region_begin)
    pos = np.array([p - region_begin for p, _, _, _ in recs])
    alt = np.array(["ACGT".index(a[0]) for _, _, a, _ in recs], np.uint8)
    take = rng.random(len(recs)) < 0.5
    hap1 = ref.copy()
    hap1[pos[take]] = alt[take]
    haps = [ref, hap1]
    which = rng.integers(0, 2, size=n_reads)
    start = np.sort(rng.integers(1, n_ref - read_len, size=n_reads))
    reads = [CODE[haps[h][s:s + read_len]] for h, s in zip(which, start)]
    o = Oracle(synth.bases_to_str(ref), recs, region_begin=region_begin)
    g = o.genotyper(1, 1)
    g.push(reads, pos=start + region_begin)
    words = g.calls()
    at = 0
    n_het = 0
    for k in range(len(recs)):
        gt1, gt2, gq, ref_depth, alt_depth, _amb, _pp, n_tri = (int(x) for x in words[at:at + 8])
        assert n_tri == 3
        phred = words[at + 8:at + 8 + n_tri]
        at += 8 + n_tri
        if pos[k] < 200 or pos[k] > n_ref - 200:
            continue  # (thin coverage at the region's ends)
        if take[k]:
            assert (gt1, gt2) == (0, 1) and phred[1] == 0 and ref_depth > 3 and alt_depth > 3, (k, gt1, gt2, list(phred), ref_depth, alt_depth)
            n_het += 1
        else:
            assert (gt1, gt2) == (0, 0) and phred[0] == 0 and alt_depth == 0, (k, gt1, gt2, list(phred), ref_depth, alt_depth)
        assert gq > 0
    assert at == len(words) and n_het > 30


def test_reads_of_an_indel_haplotype_come_back_with_its_alleles():
    """SNPs, insertions and deletions every 80 bp: error-free reads cut from the alternative haplotype must align as one
    path over the whole read without mismatches and carry, at every site well inside the read, the allele the haplotype
    has there (graph walks over alleles of unequal length, special positions of inserted bases)"""
    n_ref, rb, L, n_reads = 30000, 500000, 150, 400
    rng = np.random.default_rng(4)
    ref = synth.make_reference(n_ref, seed=11)
    recs = synth.make_indel_records(ref, 80, seed=6, region_begin=rb)
    take = rng.random(len(recs)) < 0.5
    hap, coord, cur = [], [], 0  # the haplotype and, per base, the reference coordinate it came from (-1: inserted)
    for (p, r, alts, _), t in zip(recs, take):
        p -= rb
        hap.append(ref[cur:p])
        coord.append(np.arange(cur, p))
        if t:
            a = np.array(["ACGT".index(c) for c in alts[0]], np.uint8)
            c = np.full(len(a), -1)
            c[0] = p
            hap.append(a)
            coord.append(c)
        else:
            hap.append(ref[p:p + len(r)])
            coord.append(np.arange(p, p + len(r)))
        cur = p + len(r)
    hap.append(ref[cur:])
    coord.append(np.arange(cur, n_ref))
    hap1, coord1 = np.concatenate(hap), np.concatenate(coord)
    o = Oracle(synth.bases_to_str(ref), recs, region_begin=rb)
    start = rng.integers(1, len(hap1) - L, size=n_reads)
    got = o.align([CODE[hap1[s:s + L]] for s in start])
    site_pos = np.array([p - rb for p, _, _, _ in recs])
    site_end = site_pos + np.array([len(r[1]) for r in recs])
    n_sites = 0
    for i, (fwd, _rev) in enumerate(got):
        assert len(fwd["paths"]) == 1, (i, fwd)
        p = fwd["paths"][0]
        assert (p["mm"], p["rs"], p["re"]) == (0, 0, L - 1), (i, p)
        seen = {order: nums for order, nums in p["vars"]}
        cs = coord1[start[i]:start[i] + L]
        cs = cs[cs >= 0]
        for k in np.nonzero((site_pos > cs.min() + 3) & (site_end < cs.max() - 3))[0]:
            nums = seen.get(int(rb + site_pos[k] + 1))
            assert nums is not None and (1 if take[k] else 0) in nums, (i, k, seen)
            n_sites += 1
    assert n_sites > 300


def _one_site_scores(read_codes, pos, **push_kw):
    """log_score triangle, gt_cov and max_log_score of a one-SNP graph after one read (layout: harness.canonical_scores)"""
    ref = synth.make_reference(400, seed=5)
    p = 200
    alt = "ACGT"[(int(ref[p]) + 1) % 4]
    recs = [(p, "ACGT"[int(ref[p])], [alt], None)]
    o = Oracle(synth.bases_to_str(ref), recs)
    g = o.genotyper(1, 1)
    reads = []
    for kind, start, n_err in read_codes:
        bases = ref[start:start + 150].copy()
        if kind == "alt":
            bases[p - start] = "ACGT".index(alt)
        for k in range(n_err):  # substitutions far from the site
            bases[5 + 7 * k] = (bases[5 + 7 * k] + 1) % 4
        reads.append(CODE[bases])
    g.push(reads, pos=np.array([s for _, s, _ in read_codes]) + pos, **push_kw)
    w = g.scores()
    # hap header 3 + mapq_squared 2 + 2 alleles x (2 + 2 + 6) = 25 words, then the sample: 4 hap_u32, 2 gt_cov, 3 log_score
    assert w[1] == 2
    return list(w[31:34]), list(w[29:31]), int(w[25])


def test_explain_to_score_on_one_read():
    """Haplotype::explain_to_score (haplotype.cpp:462-585) as SURVEY.md 8(a17) states it: eps = max(12 - mismatches - ...,
    8) - 4; a genotype gets eps when both its alleles are explained, eps - 1 when one is, 0 when none"""
    assert _one_site_scores([("ref", 100, 0)], 0) == ([8, 7, 0], [1, 0], 8)
    assert _one_site_scores([("alt", 100, 0)], 0) == ([0, 7, 8], [0, 1], 8)
    assert _one_site_scores([("alt", 100, 1)], 0) == ([0, 6, 7], [0, 1], 7)
    assert _one_site_scores([("alt", 100, 3)], 0) == ([0, 4, 5], [0, 1], 5)
    assert _one_site_scores([("alt", 100, 5)], 0) == ([0, 3, 4], [0, 1], 4)  # eps never drops below 4
    both, cov, _ = _one_site_scores([("ref", 100, 0), ("alt", 110, 0), ("alt", 90, 0)], 0)
    assert both == [8, 21, 16] and cov == [1, 2]


def test_explain_to_score_penalties():
    """... - 2 for MAPQ below 25 (alignment.cpp:365-480 sets IS_MAPQ_BAD), floor at 8 before the final - 4"""
    assert _one_site_scores([("alt", 100, 0)], 0, mapq=np.array([20]))[0] == [0, 5, 6]
    assert _one_site_scores([("alt", 100, 1)], 0, mapq=np.array([20]))[0] == [0, 4, 5]
    assert _one_site_scores([("alt", 100, 3)], 0, mapq=np.array([20]))[0] == [0, 3, 4]


def test_reverse_complemented_reads_align_in_the_second_orientation():
    """align_read (alignment.cpp:331-363): an unpaired read and the mate of a concordant pair are aligned as given only;
    a paired read that is not part of a concordant pair is also tried reverse-complemented.  A reverse-complemented
    error-free read must then come back in the second orientation exactly as the original does in the first"""
    n_ref, rb, L = 20000, 100000, 150
    rng = np.random.default_rng(12)
    ref = synth.make_reference(n_ref, seed=21)
    recs = synth.make_snp_records(ref, 100, seed=2, region_begin=rb)
    o = Oracle(synth.bases_to_str(ref), recs, region_begin=rb)
    n = 100
    start = rng.integers(1, n_ref - L, size=n)
    fwd_reads = [CODE[ref[s:s + L]] for s in start]
    rc_reads = [CODE[3 - ref[s:s + L][::-1]] for s in start]
    zeros = np.zeros(n, np.int32)
    a = o.align(fwd_reads)
    assert all(x[0]["paths"] and not x[1]["paths"] for x in a)
    unpaired = o.align(rc_reads)
    assert all(not x[0]["paths"] and not x[1]["paths"] for x in unpaired)  # forward only, and forward does not match
    discordant = np.full(n, 1 | 64, np.uint16)  # paired, not a proper pair, insert size far beyond 1200
    b = o.align(rc_reads, flags=discordant, tid=zeros, mtid=zeros, isize=np.full(n, 50000))
    for i in range(n):
        assert b[i][1] == a[i][0] and not b[i][0]["paths"], i
    proper = np.full(n, 1 | 2 | 32 | 64, np.uint16)  # paired, proper pair, mate reverse, first in pair
    c = o.align(rc_reads, flags=proper, tid=zeros, mtid=zeros, isize=np.full(n, 400))
    assert all(not x[0]["paths"] and not x[1]["paths"] for x in c)
