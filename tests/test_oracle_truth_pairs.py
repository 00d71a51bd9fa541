"""Read PAIRS held to simulated truth -- part of the kill suite of tests/oracle_mutants/ (the mechanical audit found update_paths /
get_better_paths, src/typer/alignment.cpp:482-620, untouched by the ground-truth suite: its only pairs were checked for relations
between VCF fields, which hold whoever is dropped).

FR pairs (mate 1 forward, mate 2 reverse: sequences as a BAM file stores them, both on the reference's strand) are drawn error-free
from two known haplotypes at 30x; half of the fragments have mate 1 on the right.  Every mate that holds a site with ten bases to
spare must be counted for its allele: the AD column lies between that count and the count of mates that touch the site; the
genotype is the haplotypes'.  A pair that is dropped, matched with the wrong orientation or selected from the wrong side shows as
a depth below the bound.  Then one pair by hand: mapping quality 24 on one mate and 25 on the other -- the first is "bad" (< 25,
alignment.cpp:497) and its likelihood term is 2 smaller (explain_to_score, haplotype.cpp:482)."""
import numpy as np

from graphtyper_amd import synth
from oracle_lib import Oracle
from test_vcf_text import _parse

READ_LEN = 151
PAIRED, PROPER, REVERSED, MATE_REVERSED, FIRST, SECOND = 1, 2, 16, 32, 64, 128


def test_pairs_are_counted_for_the_alleles_they_were_drawn_with():
    n_ref, rb = 16000, 640000
    rng = np.random.default_rng(4)
    ref = synth.make_reference(n_ref, seed=91)
    recs = synth.make_snp_records(ref, 100, seed=3, region_begin=rb)
    pos = np.array([p - rb for p, _, _, _ in recs])
    alt = np.array(["ACGT".index(a[0]) for _, _, a, _ in recs], np.uint8)
    kind = rng.integers(0, 3, size=len(recs))  # 0: no haplotype, 1: one, 2: both
    haps = [ref.copy(), ref.copy()]
    haps[0][pos[kind >= 1]] = alt[kind >= 1]
    haps[1][pos[kind == 2]] = alt[kind == 2]
    n_pairs = 30 * n_ref // (2 * READ_LEN)
    rows = []  # (position, bases, flag, isize, name, haplotype)
    for i in range(n_pairs):
        h = int(rng.integers(0, 2))
        frag = int(rng.integers(READ_LEN + 40, 700))
        s = int(rng.integers(1, n_ref - frag - 1))
        left, right = haps[h][s:s + READ_LEN], haps[h][s + frag - READ_LEN:s + frag]
        first_left = rng.random() < 0.5
        rows.append((s, left, PAIRED | PROPER | MATE_REVERSED | (FIRST if first_left else SECOND), frag, i, h))
        rows.append((s + frag - READ_LEN, right, PAIRED | PROPER | REVERSED | (SECOND if first_left else FIRST), -frag, i, h))
    rows.sort(key=lambda r: r[0])
    og = Oracle(synth.bases_to_str(ref), recs, region_begin=rb).genotyper(1, 1)
    n = len(rows)
    og.push([synth._CODE_OF_BASE[r[1]] for r in rows], flags=np.array([r[2] for r in rows], np.uint16), tid=np.zeros(n, np.int32), mtid=np.zeros(n, np.int32),
            pos=np.array([r[0] + rb for r in rows], np.int64), isize=np.array([r[3] for r in rows], np.int64), mapq=np.full(n, 60, np.uint8),
            score_diff=np.zeros(n, np.uint8), name=np.array([r[4] for r in rows], np.uint64), sample=np.zeros(n, np.int32), rg=np.zeros(n, np.int32))
    og.finish()
    _, records = _parse(og.vcf_records("chrT", ["S"]))
    assert len(records) == len(recs)
    lo = np.zeros((len(recs), 2), int)
    hi = np.zeros((len(recs), 2), int)
    for s, _, _, _, _, h in rows:
        for k in np.nonzero((pos >= s) & (pos < s + READ_LEN))[0]:
            a = int(haps[h][pos[k]] == alt[k])
            hi[k, a] += 1
            lo[k, a] += int(pos[k] - s >= 10 and s + READ_LEN - 1 - pos[k] >= 10)
    checked = 0
    for k, r in enumerate(records):
        if not 800 <= pos[k] <= n_ref - 800:
            continue  # (thin coverage at the ends: a pair needs room)
        ad = [int(x) for x in r["samples"][0][1].split(",")]
        for a in (0, 1):
            assert lo[k, a] <= ad[a] <= hi[k, a], (k, recs[k], ad, lo[k].tolist(), hi[k].tolist())
        assert r["samples"][0][0] == ("0/0", "0/1", "1/1")[kind[k]], (k, r["samples"][0], kind[k])
        checked += 1
    assert checked > 100 and lo.sum() > 0.85 * hi.sum()


def test_one_pair_with_mapping_qualities_24_and_25():
    """each mate over one site with the alternative allele, nothing else in the stream: the coverage of both sites is 1 for the
    alternative allele with the pair's flag (alt_proper_pair_depth 1), and the likelihood cells get epsilon 12 - 4 = 8 (alt/alt) and 7
    (ref/alt) from the mate with mapping quality 25, 6 and 5 from the mate with 24 (IS_MAPQ_BAD: 12 - 2 = 10 -> max(10, 8) - 4 = 6)"""
    ref = synth.make_reference(2400, seed=33)
    rb = 90000
    sites = [600, 1100]
    recs = [(rb + p, "ACGT"[ref[p]], ["ACGT"[(ref[p] + 1) % 4]], None) for p in sites]
    hap = ref.copy()
    for p in sites:
        hap[p] = (ref[p] + 1) % 4
    for mapqs in ((24, 25), (25, 24)):
        og = Oracle(synth.bases_to_str(ref), recs, region_begin=rb).genotyper(1, 1)
        s1, s2 = 600 - 75, 1100 - 75
        frag = s2 + READ_LEN - s1
        reads = [synth._CODE_OF_BASE[hap[s1:s1 + READ_LEN]], synth._CODE_OF_BASE[hap[s2:s2 + READ_LEN]]]
        og.push(reads, flags=np.array([PAIRED | PROPER | MATE_REVERSED | FIRST, PAIRED | PROPER | REVERSED | SECOND], np.uint16), tid=np.zeros(2, np.int32),
                mtid=np.zeros(2, np.int32), pos=np.array([s1 + rb, s2 + rb], np.int64), isize=np.array([frag, -frag], np.int64),
                mapq=np.array(mapqs, np.uint8), score_diff=np.zeros(2, np.uint8), name=np.array([7, 7], np.uint64), sample=np.zeros(2, np.int32),
                rg=np.zeros(2, np.int32))
        og.finish()
        s = og.scores().tolist()
        # per haplotype 25 words (id, num, clipped_reads, mapq_squared, 2 x 10 per-allele words), per sample 4 + 2 + 3, then the
        # connections of each allele: the first mate's (site 0, allele 1) is linked once to (site 1, allele 1) of the second
        # (vcf_writer.cpp:186-227: one count per pair of keys, at the earlier site)
        assert len(s) == 25 + 9 + 5 + 25 + 9 + 2
        h0, h1 = s[:39], s[39:]
        assert h0[34:] == [0, 1, 1, 0, 1] and h1[34:] == [0, 0]
        for k, h in enumerate((h0, h1)):
            sample = h[25:25 + 9]  # max_log_score, ambiguous_depth, ambiguous_depth_alt, alt_proper_pair_depth, gt_coverage[2], log_score[3]
            eps = 8 if mapqs[k] == 25 else 6
            assert sample == [eps, 0, 0, 1, 0, 1, 0, eps - 1, eps], (mapqs, k, sample)
            assert h[3:5] == [mapqs[k] ** 2, 0]
            strand = h[5 + 10 + 6:5 + 20]  # r1 forward, r1 reverse, r2 forward, r2 reverse of the alternative allele
            assert strand == ([1, 0, 0, 0] if k == 0 else [0, 0, 0, 1])


def test_one_pair_whose_mates_both_hold_the_site():
    """a fragment of 200 bases: both mates read the alternative allele of the one site.  Each is counted (coverage 2, 2 x 8 for
    alt/alt); a site is not linked to itself (vcf_writer.cpp:186-227 links keys of LATER sites only)"""
    ref = synth.make_reference(2400, seed=33)
    rb, site = 90000, 600
    recs = [(rb + site, "ACGT"[ref[site]], ["ACGT"[(ref[site] + 1) % 4]], None)]
    hap = ref.copy()
    hap[site] = (ref[site] + 1) % 4
    og = Oracle(synth.bases_to_str(ref), recs, region_begin=rb).genotyper(1, 1)
    s1, s2 = site - 120, site - 120 + 200 - READ_LEN
    reads = [synth._CODE_OF_BASE[hap[s1:s1 + READ_LEN]], synth._CODE_OF_BASE[hap[s2:s2 + READ_LEN]]]
    og.push(reads, flags=np.array([PAIRED | PROPER | MATE_REVERSED | FIRST, PAIRED | PROPER | REVERSED | SECOND], np.uint16), tid=np.zeros(2, np.int32),
            mtid=np.zeros(2, np.int32), pos=np.array([s1 + rb, s2 + rb], np.int64), isize=np.array([200, -200], np.int64),
            mapq=np.array([60, 60], np.uint8), score_diff=np.zeros(2, np.uint8), name=np.array([7, 7], np.uint64), sample=np.zeros(2, np.int32),
            rg=np.zeros(2, np.int32))
    og.finish()
    s = og.scores().tolist()
    assert len(s) == 25 + 9 + 2
    assert s[25:] == [16, 0, 0, 2, 0, 2, 0, 14, 16, 0, 0]
    assert s[3:5] == [2 * 3600, 0] and s[5 + 10 + 6:5 + 20] == [1, 0, 0, 1]  # mate 1 forward, mate 2 reverse



def test_a_proper_pair_with_the_reference_allele_is_no_alternative_proper_pair():
    """coverage_to_gts (haplotype.cpp:315-361): alt_proper_pair_depth counts the reads of proper pairs that hold an ALTERNATIVE allele"""
    ref = synth.make_reference(2400, seed=33)
    rb = 90000
    sites = [600, 1100]
    recs = [(rb + p, "ACGT"[ref[p]], ["ACGT"[(ref[p] + 1) % 4]], None) for p in sites]
    og = Oracle(synth.bases_to_str(ref), recs, region_begin=rb).genotyper(1, 1)
    s1, s2 = 600 - 75, 1100 - 75
    frag = s2 + READ_LEN - s1
    reads = [synth._CODE_OF_BASE[ref[s1:s1 + READ_LEN]], synth._CODE_OF_BASE[ref[s2:s2 + READ_LEN]]]
    og.push(reads, flags=np.array([PAIRED | PROPER | MATE_REVERSED | FIRST, PAIRED | PROPER | REVERSED | SECOND], np.uint16), tid=np.zeros(2, np.int32),
            mtid=np.zeros(2, np.int32), pos=np.array([s1 + rb, s2 + rb], np.int64), isize=np.array([frag, -frag], np.int64),
            mapq=np.array([60, 60], np.uint8), score_diff=np.zeros(2, np.uint8), name=np.array([7, 7], np.uint64), sample=np.zeros(2, np.int32),
            rg=np.zeros(2, np.int32))
    og.finish()
    s = og.scores().tolist()
    first = 25 + 9 + 5
    assert len(s) == first + 25 + 9 + 2
    for h in (s[:first], s[first:]):
        assert h[25:34] == [8, 0, 0, 0, 1, 0, 8, 7, 0]
