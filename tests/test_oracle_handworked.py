"""The unpinned half of the oracle (seed chaining, walks, filters, orientation selection, scoring) on cases small enough to be
worked BY HAND from the reference's text -- the expected values below were derived from src/typer/alignment.cpp,
genotype_paths.cpp, vcf_writer.cpp and src/graph/haplotype.cpp, not from running anything.  They are not golden vectors of the
reference (it has none for this path); they are what gives the ground-truth suite teeth: tests/oracle_mutants/ holds one-token
misreadings of those files, and every one of them has to fail a test here or in test_oracle_truth.py."""
import numpy as np

from graphtyper_amd import synth
from oracle_lib import Oracle

CODE = np.array([1, 2, 4, 8], np.uint8)


def parse_scores(w, n_samples=1):
    """oracle/gto_capi.cpp: gto_scores_dump -> [dict per haplotype]"""
    w = [int(x) for x in w]
    at, out = 0, []
    while at < len(w):
        h = dict(order=w[at], num=w[at + 1], clipped=w[at + 2], mapq2=w[at + 3] | (w[at + 4] << 32), alleles=[], samples=[])
        at += 5
        for _ in range(h["num"]):
            h["alleles"].append(dict(clipped_bp=w[at] | (w[at + 1] << 32), mapq2=w[at + 2] | (w[at + 3] << 32), score_diff=w[at + 4], mismatches=w[at + 5],
                                     r1f=w[at + 6], r1r=w[at + 7], r2f=w[at + 8], r2r=w[at + 9]))
            at += 10
        tri = h["num"] * (h["num"] + 1) // 2
        for _ in range(n_samples):
            s = dict(max=w[at], amb=w[at + 1], amb_alt=w[at + 2], pp=w[at + 3], cov=w[at + 4:at + 4 + h["num"]])
            at += 4 + h["num"]
            s["log"] = w[at:at + tri]
            at += tri
            s["conn"] = []
            for _a in range(h["num"]):
                n = w[at]
                at += 1
                d = {}
                for _k in range(n):
                    hap2 = w[at]
                    at += 1
                    # (the other haplotype's allele count is not in the stream: connection rows are read by the caller who knows it)
                    d[hap2] = None
                    raise NotImplementedError("connections present: use parse_scores_with(nums)")
                s["conn"].append(d)
            h["samples"].append(s)
        out.append(h)
    return out


def parse_scores_with(w, nums, n_samples=1):
    """... when connections are present: nums = alleles per haplotype, in order"""
    w = [int(x) for x in w]
    at, out = 0, []
    for h_i, num in enumerate(nums):
        assert w[at + 1] == num
        at += 5 + 10 * num
        tri = num * (num + 1) // 2
        samples = []
        for _ in range(n_samples):
            s = dict(max=w[at], amb=w[at + 1], amb_alt=w[at + 2], pp=w[at + 3], cov=w[at + 4:at + 4 + num])
            at += 4 + num
            s["log"] = w[at:at + tri]
            at += tri
            s["conn"] = []
            for _a in range(num):
                n = w[at]
                at += 1
                d = {}
                for _k in range(n):
                    hap2 = w[at]
                    d[hap2] = w[at + 1:at + 1 + nums[hap2]]
                    at += 1 + nums[hap2]
                s["conn"].append(d)
            samples.append(s)
        out.append(samples)
    assert at == len(w)
    return out


def one_snp(n_ref=600, site=300, seed=5):
    ref = synth.make_reference(n_ref, seed=seed)
    alt = (int(ref[site]) + 1) % 4
    return ref, [(site, "ACGT"[int(ref[site])], ["ACGT"[alt]], None)], alt


def substitute(bases, at):
    b = bases.copy()
    for p in at:
        b[p] = (b[p] + 2) % 4
    return b


# ---------------------------------------------------------------------------------------------------------------
# walks at the read's end: budget min(2 + length / 11, best so far), best so far starting at 7 (genotype_paths.cpp:483-553)
# ---------------------------------------------------------------------------------------------------------------
def test_tail_walk_budget():
    """150 bases = four k-mers (bases 0..124) + a tail; the walk's sub-read is seq[124:] (26 characters, the overlap base
    included): budget 2 + 26 / 11 = 4.  Four substitutions in the tail: extended, 4 mismatches.  Five: not extended -- the path
    stays 125 bases long."""
    ref, recs, _ = one_snp(seed=31)
    o = Oracle(synth.bases_to_str(ref), recs)
    clean = ref[100:250]
    four = substitute(clean, [128, 133, 139, 146])
    five = substitute(clean, [128, 133, 139, 143, 146])
    got = o.align([CODE[clean], CODE[four], CODE[five]])
    p = [g[0]["paths"] for g in got]
    assert [len(x) for x in p] == [1, 1, 1]
    assert (p[0][0]["rs"], p[0][0]["re"], p[0][0]["mm"]) == (0, 149, 0)
    assert (p[1][0]["rs"], p[1][0]["re"], p[1][0]["mm"]) == (0, 149, 4)
    assert (p[2][0]["rs"], p[2][0]["re"], p[2][0]["mm"]) == (0, 124, 0) and got[2][0]["longest"] == 125


def duplicated_reference(seed=41, n_ref=3000, a=500, b=1700, size=300, differ_at=None):
    """a reference whose stretch [a, a + size) is repeated at b (optionally with one base changed in the copy)"""
    ref = synth.make_reference(n_ref, seed=seed)
    ref[b:b + size] = ref[a:a + size]
    if differ_at is not None:
        ref[b + differ_at] = (ref[b + differ_at] + 1) % 4
    return ref


def test_every_walk_that_ties_contributes():
    """a read from inside an exact two-copy repeat: two chains of 125 bases, both tails match without mismatches -- both walks tie
    the best budget (0) and BOTH paths are extended (genotype_paths.cpp:533-552); walks are done (2 paths <= 256)"""
    ref = duplicated_reference()
    recs = [(100, "ACGT"[int(ref[100])], ["ACGT"[(int(ref[100]) + 1) % 4]], None)]
    o = Oracle(synth.bases_to_str(ref), recs)
    got = o.align([CODE[ref[560:710]]])
    paths = got[0][0]["paths"]
    assert sorted((p["start"], p["end"], p["mm"]) for p in paths) == [(561, 710, 0), (1761, 1910, 0)]
    assert got[0][0]["longest"] == 150


def test_only_the_fewest_mismatches_stay():
    """the copy differs in one base under the second k-mer: there the read's k-mer is the copy's Hamming-1 neighbour -- a chain
    with one mismatch; remove_paths_with_too_many_mismatches (genotype_paths.cpp:360-380) keeps the paths with the fewest"""
    ref = duplicated_reference(differ_at=110)
    recs = [(100, "ACGT"[int(ref[100])], ["ACGT"[(int(ref[100]) + 1) % 4]], None)]
    o = Oracle(synth.bases_to_str(ref), recs)
    got = o.align([CODE[ref[560:710]]])  # (the changed base is read base 50: inside k-mer 1 only)
    paths = got[0][0]["paths"]
    assert [(p["start"], p["end"], p["mm"]) for p in paths] == [(561, 710, 0)]


# ---------------------------------------------------------------------------------------------------------------
# scoring: which reads count, and with which epsilon
# ---------------------------------------------------------------------------------------------------------------
def score_one(ref, recs, reads, pos, n_samples=1, **kw):
    o = Oracle(synth.bases_to_str(ref), recs)
    g = o.genotyper(n_samples, 1)
    g.push([CODE[r] for r in reads], pos=np.asarray(pos), **kw)
    return g.scores()


def alt_read(ref, alt, site, start, length=150):
    r = ref[start:start + length].copy()
    r[site - start] = alt
    return r


def test_single_reads_need_more_than_94_matched_bases():
    """compare_pair_of_genotype_paths (single, genotype_paths.cpp:943-974): the chosen orientation must have MORE than 94 matched
    bases.  A 94-base read aligns over all its 94 bases (three k-mers) and is not scored; a 95-base read is."""
    ref, recs, alt = one_snp()
    for length, want in ((94, [0, 0]), (95, [0, 1])):
        w = parse_scores(score_one(ref, recs, [alt_read(ref, alt, 300, 260, length)], [260]))
        assert w[0]["samples"][0]["cov"] == want, (length, w[0]["samples"][0])


def test_epsilon_penalties_by_hand():
    """explain_to_score (haplotype.cpp:462-585): eps = max(12 - mismatches - 3 [paths at several loci] - 2 [MAPQ < 25] - 3 [not
    aligned over the whole read] - 1 [fewer than 3 bases of the read beyond the site], 8) - 4; log_score: eps for a genotype of
    explained alleles, eps - 1 with one of them (order: 0/0, 0/1, 1/1)"""
    ref, recs, alt = one_snp()
    log = lambda reads, pos, **kw: parse_scores(score_one(ref, recs, reads, pos, **kw))[0]["samples"][0]["log"]
    assert log([alt_read(ref, alt, 300, 200)], [200]) == [0, 7, 8]
    # the site on read base 3: start + 3 <= order holds (vcf_writer.cpp is_overlapping) -> no penalty; on base 2: - 1
    assert log([alt_read(ref, alt, 300, 297)], [297]) == [0, 7, 8]
    assert log([alt_read(ref, alt, 300, 298)], [298]) == [0, 6, 7]
    # MAPQ 24 is bad (< 25), 25 is not; 22 is (a reading of "< 21" would miss it)
    assert log([alt_read(ref, alt, 300, 200)], [200], mapq=np.array([24])) == [0, 5, 6]
    assert log([alt_read(ref, alt, 300, 200)], [200], mapq=np.array([22])) == [0, 5, 6]
    assert log([alt_read(ref, alt, 300, 200)], [200], mapq=np.array([25])) == [0, 7, 8]
    # a read whose tail does not extend (five substitutions behind base 125): aligned over 125 of 150 bases -> - 3
    short = substitute(alt_read(ref, alt, 300, 250), [128, 133, 139, 143, 146])
    assert log([short], [250]) == [0, 4, 5]


def test_reads_with_paths_at_several_loci():
    """a read from an exact two-copy repeat, one copy with a SNP site: two paths at different loci -- not unique
    (genotype_paths.cpp:219-231) -> eps - 3; both are reference paths, so both stay (remove_non_ref_paths_when_read_matches_ref)"""
    ref = duplicated_reference()
    site = 600
    recs = [(site, "ACGT"[int(ref[site])], ["ACGT"[(int(ref[site]) + 1) % 4]], None)]
    w = parse_scores(score_one(ref, recs, [ref[560:710].copy()], [560]))
    assert w[0]["samples"][0]["log"] == [5, 4, 0] and w[0]["samples"][0]["cov"] == [1, 0]


def test_mismatch_ratio_gates():
    """are_genotype_paths_good (vcf_writer.cpp:28-86): more than 5 % mismatches -> not scored; a read that is not aligned over its
    whole length: more than 2.5 %"""
    ref, recs, alt = one_snp()
    cov = lambda reads, pos: parse_scores(score_one(ref, recs, reads, pos))[0]["samples"][0]["cov"]
    base = alt_read(ref, alt, 300, 250)  # site at read base 50
    # one substitution under each k-mer (bases 5, 40, 70, 100) + three in the tail: 7 of 150 = 4.7 % -> scored
    assert cov([substitute(base, [5, 40, 70, 100, 130, 137, 144])], [250]) == [0, 1]
    # ... + a fourth in the tail: 8 of 150 = 5.3 % -> not scored
    assert cov([substitute(base, [5, 40, 70, 100, 130, 135, 140, 145])], [250]) == [0, 0]
    # tail not extended (five substitutions) and three substitutions in the chain: 3 of 125 = 2.4 % -> scored; four: 3.2 % -> not
    tail = [128, 133, 139, 143, 146]
    assert cov([substitute(base, [5, 40, 70] + tail)], [250]) == [0, 1]
    assert cov([substitute(base, [5, 40, 70, 100] + tail)], [250]) == [0, 0]


def test_ambiguous_read_is_ambiguous_towards_the_reference():
    """add_coverage (haplotype.cpp:180-227): a read that explains the reference allele AND an alternative one (an N on the site)
    counts as ambiguous depth, not as ambiguous ALT depth, and covers no allele uniquely"""
    ref, recs, alt = one_snp()
    r = CODE[ref[200:350]].copy()
    r[100] = 15
    o = Oracle(synth.bases_to_str(ref), recs)
    g = o.genotyper(1, 1)
    g.push([r], pos=np.array([200]))
    s = parse_scores(g.scores())[0]["samples"][0]
    assert s["cov"] == [0, 0] and s["amb"] == 1 and s["amb_alt"] == 0 and s["log"] == [8, 8, 8]


def test_connection_weight():
    """vcf_writer.cpp:120-139: the alleles a read explains at two sites are connected; a pair of allele sets with weight
    |set1| * |set2| >= 3 counts 6 / weight times per allele pair.  A three-allele site read as N (three alleles explained) and a
    SNP 20 bases on carrying the alternative allele: weight 3 -> every (allele of site 1, allele 1 of site 2) pair counts twice."""
    ref = synth.make_reference(600, seed=9)
    s1, s2 = 300, 320
    a1 = ["ACGT"[(int(ref[s1]) + k) % 4] for k in (1, 2)]
    a2 = "ACGT"[(int(ref[s2]) + 1) % 4]
    recs = [(s1, "ACGT"[int(ref[s1])], a1, None), (s2, "ACGT"[int(ref[s2])], [a2], None)]
    r = CODE[ref[200:350]].copy()
    r[s1 - 200] = 15
    r[s2 - 200] = CODE["ACGT".index(a2)]
    o = Oracle(synth.bases_to_str(ref), recs)
    g = o.genotyper(1, 1)
    g.push([r], pos=np.array([200]))
    haps = parse_scores_with(g.scores(), [3, 2])
    conn = haps[0][0]["conn"]  # site 1, sample 0: per allele {other haplotype: counts per allele}
    assert [c.get(1) for c in conn] == [[0, 2], [0, 2], [0, 2]]


def test_equal_orientations_keep_the_first():
    """compare_pair_of_genotype_paths (single): with both orientations aligned over the same length the SECOND wins only with FEWER
    mismatches.  A read that is its own reverse complement aligns identically both ways; with Options::force_align_both_orientations
    both orientations are tried -- the first one is kept, so the read counts on the forward strand (read_strand: r2_forward for an unpaired read)"""
    rng = np.random.default_rng(3)
    x = rng.integers(0, 4, size=75).astype(np.uint8)
    s = np.concatenate([x, (3 - x)[::-1]])  # 150 bases, reverse-palindromic
    ref = synth.make_reference(900, seed=8)
    ref[400:550] = s
    site = 420
    alt = (int(ref[site]) + 1) % 4
    recs = [(site, "ACGT"[int(ref[site])], ["ACGT"[alt]], None)]
    o = Oracle(synth.bases_to_str(ref), recs, force_both=True)  # (Options::force_align_both_orientations: an unpaired read is tried both ways)
    both = o.align([CODE[s]])
    assert both[0][0]["paths"] and both[0][0] == both[0][1]
    g = o.genotyper(1, 1)
    g.push([CODE[s]], pos=np.array([400]))
    a = parse_scores(g.scores())[0]["alleles"][0]
    # (haplotype.cpp:262-281: a read without IS_FIRST_IN_PAIR counts as "read 2"; forward because the first orientation was kept)
    assert (a["r1f"], a["r1r"], a["r2f"], a["r2r"]) == (0, 0, 1, 0)


def test_ten_reads_over_the_reference_test_contigs():
    """tests/golden/handworked_paths.json: ten reads over index_test.fa chr1-chr3, their GenotypePaths worked by hand"""
    import json
    import os
    from fixtures import contig
    from oracle_lib import encode
    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "handworked_paths.json")))["cases"]
    assert len(cases) == 10
    for c in cases:
        ref, recs = contig(c["contig"])
        got = Oracle(ref, recs).align([encode(c["read"])])[0][0]
        want = [dict(start=p["start"], end=p["end"], rs=p["rs"], re=p["re"], mm=p["mm"], vars=[(o, tuple(a)) for o, a in p["vars"]]) for p in c["paths"]]
        have = [dict(start=p["start"], end=p["end"], rs=p["rs"], re=p["re"], mm=p["mm"], vars=sorted((o, tuple(a)) for o, a in p["vars"])) for p in got["paths"]]
        assert have == want, (c["contig"], c["why"], have, want)
