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
